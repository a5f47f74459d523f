// sched.inc.h -- the stream scheduler behind the reference's per-gate API (included by capi.hip).
//
// The reference launches one kernel per gate on the caller's stream and brackets it with
// three small memcpys (src/cufhe_gates_gpu.cu:148-158); throughput comes from hundreds of
// concurrent streams (test/test_util.h:36-62).  Here a gate call only RECORDS the gate; the
// recorded gates of a device are launched together as one batch (one blind-rotate launch,
// one key-switch launch) when the host asks for a result (Synchronize / StreamQuery), when a
// new gate depends on a recorded one, or when the batch is full.  What callers can observe
// is unchanged:
//   - gates issued on one Stream take effect in issue order (a gate that reads or
//     overwrites a ciphertext written/read by a recorded gate forces the batch out first);
//   - `out.tlwehost` is valid once Synchronize() returned or StreamQuery(st) returned true;
//   - non-g gates take their inputs from `tlwehost` as of the call (the reference's H2D copy
//     in stream order), unless the ciphertext is the still-unfetched result of an earlier
//     gate, in which case the device copy is the current value -- as it is in the
//     reference, where that gate's D2H precedes this gate's H2D on the stream;
//   - g-gates touch device memory only;
//   - out may alias an input (test/test_api_gpu.cu:141).
// Host <-> device traffic is batched too: inputs are gathered into one pinned block, moved
// with one H2D copy and scattered to the ciphertexts' device buffers by a kernel; outputs
// travel the other way and are copied into `tlwehost` when completion is observed.

struct cufhe_amd_ctxt {
    int level = 0;
    uint32_t* host = nullptr;
    bool registered = false;
    std::vector<uint32_t*> dev;            // one buffer per GPU (tlwedevices)
    std::vector<uint64_t> write_batch;     // batch that last wrote dev[d] and has not been fetched to host
    std::vector<uint64_t> read_batch;      // newest batch that reads dev[d]
    std::vector<uint64_t> upload_batch;    // batch that already uploads host -> dev[d]
};

namespace {

struct CopyRec { uint32_t* dev; size_t slot; int level; cufhe_amd_ctxt* ctxt; };

struct HostBuf { void* p = nullptr; size_t bytes = 0; };

struct Batch {
    uint64_t id = 0;
    std::vector<GateRef> gates[2];
    std::vector<CopyRec> uploads, downloads;
    std::vector<cufhe_amd_ctxt*> device_only_outs;   // outputs of g-gates (no download record)
    std::vector<uint32_t> in_words;        // gathered host inputs (copied to pinned memory at flush)
    size_t out_words = 0;
    HostBuf pin_in, pin_out;
    uint32_t *dev_in = nullptr, *dev_out = nullptr;
    hipEvent_t done = nullptr;
    bool finalized = false;
    size_t gate_count() const { return gates[0].size() + gates[1].size(); }
    bool empty() const { return gate_count() == 0 && uploads.empty() && downloads.empty(); }
};

struct Sched {
    hipStream_t st = nullptr;
    uint64_t next_id = 1;
    Batch* cur = nullptr;
    std::deque<Batch*> inflight;
    std::map<void*, uint64_t> stream_last;   // user stream handle -> newest batch holding its gates
    std::vector<HostBuf> pinned_cache;
    std::vector<std::pair<uint32_t*, size_t>> dev_cache;
};

// A batch is launched as soon as it holds one full round of the blind-rotate grid (256 CUs x 8
// rotations): the GPU then works on it while the host records the next one.
constexpr size_t kSchedMaxGates = 2048;
std::vector<Sched> g_sched;
std::mutex g_sched_mu;     // the reference API is single-issuer; this only guards against misuse

int sched_get(int device, Sched** out)
{
    if (int rc = use_device(device)) return rc;
    if ((int)g_sched.size() < g_gpu_num) g_sched.resize(g_gpu_num);
    Sched& s = g_sched[device];
    if (!s.st) HIP_TRY(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
    if (!s.cur) { s.cur = new Batch(); s.cur->id = s.next_id++; }
    *out = &s;
    return 0;
}

int pinned_get(Sched& s, size_t bytes, HostBuf* out)
{
    for (size_t i = 0; i < s.pinned_cache.size(); i++)
        if (s.pinned_cache[i].bytes >= bytes) {
            *out = s.pinned_cache[i];
            s.pinned_cache.erase(s.pinned_cache.begin() + i);
            return 0;
        }
    out->bytes = bytes + bytes / 2 + 4096;
    HIP_TRY(hipHostMalloc(&out->p, out->bytes, hipHostMallocDefault));
    return 0;
}
int dev_get(Sched& s, size_t bytes, uint32_t** out, size_t* cap)
{
    for (size_t i = 0; i < s.dev_cache.size(); i++)
        if (s.dev_cache[i].second >= bytes) {
            *out = s.dev_cache[i].first; *cap = s.dev_cache[i].second;
            s.dev_cache.erase(s.dev_cache.begin() + i);
            return 0;
        }
    *cap = bytes + bytes / 2 + 4096;
    HIP_TRY(hipMalloc((void**)out, *cap));
    return 0;
}

int copy_kernel_launch(DeviceState& ds, hipStream_t st, const std::vector<CopyRec>& recs, uint32_t* staging, bool to_ctxt)
{
    // staging <-> ciphertext device buffers, one lincomb (COPY) launch per level
    for (int level = 0; level < 2; level++) {
        std::vector<LinDesc> d;
        for (const CopyRec& r : recs)
            if (r.level == level) {
                uint32_t* slot = staging + r.slot;
                if (to_ctxt) d.push_back({slot, slot, r.dev, 1, 0, 0u, 0u});
                else d.push_back({r.dev, r.dev, slot, 1, 0, 0u, 0u});
            }
        if (d.empty()) continue;
        // The stream's workspace is reused from offset 0 by every launch sequence: the
        // descriptor copy of the next sequence is ordered behind the kernels of this one.
        Scratch sc;
        if (int rc = open_scratch(ds, st, d.size() * sizeof(LinDesc) + 4096, &sc)) return rc;
        LinDesc* dd;
        if (int rc = upload_descs(ds, sc, d, &dd)) return rc;
        if (int rc = launch_lincomb(st, dd, d.size(), level ? kLvl1Words : kLvl0Words)) return rc;
    }
    return 0;
}

int sched_flush(int device)
{
    Sched* sp;
    if (int rc = sched_get(device, &sp)) return rc;
    Sched& s = *sp;
    Batch* b = s.cur;
    if (b->empty()) return 0;
    DeviceState& ds = g_dev[device];
    if (b->gate_count() && !ds.keys_ready && !ds.keys2_ready) return fail(-3, "Initialize(ek) has not been called for this device");

    size_t cap;
    if (!b->uploads.empty()) {
        const size_t bytes = b->in_words.size() * sizeof(uint32_t);
        if (int rc = pinned_get(s, bytes, &b->pin_in)) return rc;
        memcpy(b->pin_in.p, b->in_words.data(), bytes);
        if (int rc = dev_get(s, bytes, &b->dev_in, &cap)) return rc;
        s.dev_cache.push_back({b->dev_in, cap});           // returned to the cache right away: stream-ordered reuse
        HIP_TRY(hipMemcpyAsync(b->dev_in, b->pin_in.p, bytes, hipMemcpyHostToDevice, s.st));
        if (int rc = copy_kernel_launch(ds, s.st, b->uploads, b->dev_in, true)) return rc;
    }
    for (int level = 0; level < 2; level++) {
        const std::vector<GateRef>& gl = b->gates[level];
        if (gl.empty()) continue;
        if (int rc = run_gates(device, s.st, level, gl.size(), [&](size_t g) { return gl[g]; })) return rc;
    }
    if (!b->downloads.empty()) {
        const size_t bytes = b->out_words * sizeof(uint32_t);
        if (int rc = dev_get(s, bytes, &b->dev_out, &cap)) return rc;
        s.dev_cache.push_back({b->dev_out, cap});
        if (int rc = copy_kernel_launch(ds, s.st, b->downloads, b->dev_out, false)) return rc;
        if (int rc = pinned_get(s, bytes, &b->pin_out)) return rc;
        HIP_TRY(hipMemcpyAsync(b->pin_out.p, b->dev_out, bytes, hipMemcpyDeviceToHost, s.st));
    }
    HIP_TRY(hipEventCreateWithFlags(&b->done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->done, s.st));
    const size_t in_cap = b->in_words.capacity();
    std::vector<uint32_t>().swap(b->in_words);
    s.inflight.push_back(b);
    s.cur = new Batch();
    s.cur->id = s.next_id++;
    s.cur->in_words.reserve(in_cap);          // batches of one program tend to repeat in size
    return 0;
}

// the batch's event has completed: hand the results to the host ciphertexts
void sched_finalize(int device, Sched& s, Batch* b)
{
    if (b->finalized) return;
    for (const CopyRec& r : b->downloads) {
        const int words = r.level ? kLvl1Words : kLvl0Words;
        memcpy(r.ctxt->host, (const uint32_t*)b->pin_out.p + r.slot, words * sizeof(uint32_t));
        if (r.ctxt->write_batch[device] == b->id) r.ctxt->write_batch[device] = 0;   // host is current again
    }
    for (cufhe_amd_ctxt* c : b->device_only_outs)
        if (c->write_batch[device] == b->id) c->write_batch[device] = 0;
    if (b->pin_in.p) s.pinned_cache.push_back(b->pin_in);
    if (b->pin_out.p) s.pinned_cache.push_back(b->pin_out);
    (void)hipEventDestroy(b->done);
    b->finalized = true;
}

int sched_drain(int device, bool wait)
{
    Sched* sp;
    if (int rc = sched_get(device, &sp)) return rc;
    Sched& s = *sp;
    while (!s.inflight.empty()) {
        Batch* b = s.inflight.front();
        if (wait) HIP_TRY(hipEventSynchronize(b->done));
        else if (hipEventQuery(b->done) != hipSuccess) break;
        sched_finalize(device, s, b);
        s.inflight.pop_front();
        delete b;
    }
    return 0;
}

bool sched_active() { return !g_sched.empty(); }

int sched_synchronize_all()
{
    for (int d = 0; d < (int)g_sched.size() && d < g_gpu_num; d++) {
        if (!g_sched[d].st) continue;
        if (int rc = sched_flush(d)) return rc;
        if (int rc = sched_drain(d, true)) return rc;
    }
    return 0;
}

void sched_destroy_all()
{
    for (int d = 0; d < (int)g_sched.size(); d++) {
        Sched& s = g_sched[d];
        if (!s.st) continue;
        (void)hipSetDevice(d + g_device_base);
        (void)hipStreamSynchronize(s.st);
        for (Batch* b : s.inflight) { sched_finalize(d, s, b); delete b; }
        s.inflight.clear();
        delete s.cur;
        s.cur = nullptr;
        for (auto& h : s.pinned_cache) (void)hipHostFree(h.p);
        for (auto& p : s.dev_cache) (void)hipFree(p.first);
        s.pinned_cache.clear(); s.dev_cache.clear();
        (void)hipStreamDestroy(s.st);
        s.st = nullptr;
    }
    g_sched.clear();
}

}  // namespace

extern "C" {

int cufhe_amd_ctxt_create(int level, uint32_t* host_words, cufhe_amd_ctxt** out)
{
    if (level != 0 && level != 1) return fail(-1, "level must be 0 or 1");
    if (!host_words || !out) return fail(-1, "null pointer");
    cufhe_amd_ctxt* c = new cufhe_amd_ctxt();
    c->level = level;
    c->host = host_words;
    const size_t bytes = (level ? kLvl1Words : kLvl0Words) * sizeof(uint32_t);
    if (hipHostRegister(host_words, bytes, hipHostRegisterDefault) == hipSuccess) c->registered = true;
    else (void)hipGetLastError();                      // pinning is an optimisation, not a requirement
    c->dev.assign(g_gpu_num, nullptr);
    c->write_batch.assign(g_gpu_num, 0);
    c->read_batch.assign(g_gpu_num, 0);
    c->upload_batch.assign(g_gpu_num, 0);
    for (int d = 0; d < g_gpu_num; d++) {
        if (int rc = use_device(d)) { delete c; return rc; }
        hipError_t e = hipMalloc((void**)&c->dev[d], bytes);
        if (e != hipSuccess) { delete c; return fail(-2, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    }
    *out = c;
    return 0;
}

int cufhe_amd_ctxt_destroy(cufhe_amd_ctxt* c)
{
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(g_sched_mu);
    bool busy = false;
    for (size_t d = 0; d < c->dev.size(); d++) busy = busy || c->write_batch[d] || c->read_batch[d] || c->upload_batch[d];
    if (busy && sched_active()) (void)sched_synchronize_all();
    for (size_t d = 0; d < c->dev.size(); d++)
        if (c->dev[d]) { (void)hipSetDevice((int)d + g_device_base); (void)hipFree(c->dev[d]); }
    if (c->registered) (void)hipHostUnregister(c->host);
    delete c;
    return 0;
}

uint32_t* cufhe_amd_ctxt_device_ptr(cufhe_amd_ctxt* c, int device)
{
    if (!c || device < 0 || device >= (int)c->dev.size()) return nullptr;
    return c->dev[device];
}

int cufhe_amd_enqueue_gate(int device, void* stream, int op, int copying, cufhe_amd_ctxt* out,
                           cufhe_amd_ctxt* in0, cufhe_amd_ctxt* in1, cufhe_amd_ctxt* in2)
{
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (op < 0 || op >= CUFHE_AMD_NUM_OPS) return fail(-1, "unknown gate op");
    if (!out || !in0) return fail(-1, "null ciphertext");
    const bool three = op == CUFHE_AMD_MUX || op == CUFHE_AMD_NMUX;
    const bool one = op == CUFHE_AMD_NOT || op == CUFHE_AMD_COPY;
    if (!one && !in1) return fail(-1, "gate needs a second operand");
    if (three && !in2) return fail(-1, "mux needs a third operand");
    cufhe_amd_ctxt* ins[3] = {in0, one ? nullptr : in1, three ? in2 : nullptr};
    for (cufhe_amd_ctxt* c : ins)
        if (c && c->level != out->level) return fail(-1, "operands of one gate must have the same level");
    Sched* sp;
    if (int rc = sched_get(device, &sp)) return rc;
    if ((int)out->dev.size() <= device) return fail(-1, "ciphertext was created before SetGPUNum raised the GPU count");

    // dependences on gates recorded in the current (unlaunched) batch force it out first
    bool hazard = out->write_batch[device] == sp->cur->id || out->read_batch[device] == sp->cur->id;
    for (cufhe_amd_ctxt* c : ins)
        if (c && c->write_batch[device] == sp->cur->id) hazard = true;
    // ... except a gate overwriting its own input only: checked before its reads are recorded
    if (hazard)
        if (int rc = sched_flush(device)) return rc;
    Sched& s = *sp;
    Batch* b = s.cur;
    const int level = out->level;
    const int words = level ? kLvl1Words : kLvl0Words;

    for (cufhe_amd_ctxt* c : ins) {
        if (!c) continue;
        if (copying && c->write_batch[device] == 0 && c->upload_batch[device] != b->id) {
            const size_t slot = b->in_words.size();
            b->in_words.insert(b->in_words.end(), c->host, c->host + words);
            b->uploads.push_back({c->dev[device], slot, level, c});
            c->upload_batch[device] = b->id;
        }
        c->read_batch[device] = b->id;
    }
    b->gates[level].push_back(GateRef{op, out->dev[device], in0->dev[device],
                                      ins[1] ? ins[1]->dev[device] : nullptr,
                                      ins[2] ? ins[2]->dev[device] : nullptr});
    out->write_batch[device] = b->id;
    if (copying) {
        b->downloads.push_back({out->dev[device], b->out_words, level, out});
        b->out_words += words;
    } else {
        b->device_only_outs.push_back(out);
    }
    s.stream_last[stream] = b->id;
    if (s.cur == b && b->gate_count() >= kSchedMaxGates)
        if (int rc = sched_flush(device)) return rc;
    return 0;
}

/* CtxtCopyH2D / CtxtCopyD2H (include/cufhe_gpu.cuh:193-207) in scheduler order.
 * to_device != 0: tlwehost (as of now) -> device buffer; else device buffer -> tlwehost,
 * visible after Synchronize / StreamQuery like a gate result. */
int cufhe_amd_enqueue_copy(int device, void* stream, cufhe_amd_ctxt* c, int to_device)
{
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (!c) return fail(-1, "null ciphertext");
    Sched* sp;
    if (int rc = sched_get(device, &sp)) return rc;
    if ((int)c->dev.size() <= device) return fail(-1, "ciphertext was created before SetGPUNum raised the GPU count");
    const int words = c->level ? kLvl1Words : kLvl0Words;
    if (to_device) {
        // a write of the device buffer: must not overtake recorded readers/writers
        if (c->write_batch[device] == sp->cur->id || c->read_batch[device] == sp->cur->id)
            if (int rc = sched_flush(device)) return rc;
        Batch* b = sp->cur;
        const size_t slot = b->in_words.size();
        b->in_words.insert(b->in_words.end(), c->host, c->host + words);
        b->uploads.push_back({c->dev[device], slot, c->level, c});
        c->upload_batch[device] = b->id;
        c->write_batch[device] = 0;                  // host and device agree once this lands
        c->read_batch[device] = b->id;               // later gate writes in this batch come after uploads anyway
    } else {
        Batch* b = sp->cur;
        b->downloads.push_back({c->dev[device], b->out_words, c->level, c});
        b->out_words += words;
        if (c->write_batch[device] == 0) c->write_batch[device] = b->id;   // host is stale until fetched
        c->read_batch[device] = b->id;
    }
    sp->stream_last[stream] = sp->cur->id;
    return 0;
}

int cufhe_amd_flush(int device)
{
    std::lock_guard<std::mutex> lk(g_sched_mu);
    return sched_flush(device);
}

int cufhe_amd_sched_stream_query(int device, void* stream)
{
    std::lock_guard<std::mutex> lk(g_sched_mu);
    Sched* sp;
    if (int rc = sched_get(device, &sp)) return rc;
    Sched& s = *sp;
    auto it = s.stream_last.find(stream);
    if (it == s.stream_last.end()) return 1;
    const uint64_t id = it->second;
    if (id == s.cur->id)
        if (int rc = sched_flush(device)) return rc;
    if (int rc = sched_drain(device, false)) return rc;
    for (Batch* b : s.inflight)
        if (b->id <= id) return 0;          // that batch (or an older one) is still running
    s.stream_last.erase(it);
    return 1;
}

}  // extern "C"
