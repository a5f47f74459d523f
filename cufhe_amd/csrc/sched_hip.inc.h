// sched_hip.inc.h -- HIP device layer of the stream scheduler (sched_core.h) and the C ABI of the
// reference's per-gate API (included by capi.hip).
//
// One DeviceSched per GPU; each owns a few internal non-blocking HIP streams (independent flushes
// overlap, dependent ones are ordered by events) and a worker thread that turns flushed dependence
// levels into launches.  Ciphertext traffic is batched per flush: inputs are gathered into one pinned
// block, moved with one H2D copy and scattered to the ciphertexts' device buffers by a kernel before
// the level that reads them; results travel the other way and land in `tlwehost` when completion is
// observed (Synchronize / StreamQuery), as in the reference (src/cufhe_gates_gpu.cu:148-158).

namespace {

long g_sched_streams = 4;        // internal HIP streams per device
long g_sched_threads = 1;        // 1: a launch worker thread per device, 0: launches on the issuing thread
long g_sched_level_gates = -1;   // a dependence level this full is launched at once: -1 = two rounds of the batch kernel's grid (8 rotations per CU each), one when the device is idle
long g_sched_total_gates = 32768;
long g_sched_idle_gates = -1;     // gates of a level at which an IDLE device is handed it (-1: one grid round)
long g_sched_copy_threads = 4;    // host threads (the calling one included) that share a large gather out of / delivery into the tlwehosts; before the first flush
long g_sched_two_lane = 1;        // flushes of several dependence levels are scheduled gate by gate on two lanes when the cost model says so (sched_core.h: compile_two_lane)
long g_sched_rename = 1;          // outputs take fresh device buffers instead of waiting for the old one's users, values return to the ciphertext's own buffer before the host may look (sched_core.h); 0: never
long g_sched_zero_copy = 1;       // 1: ciphertext staging is read / written by the scatter / gather kernels in pinned host memory (no copy-engine step)
long g_sched_affinity = 1;        // 1: a device's launch worker runs on the CPUs local to that GPU (NUMA node of its PCI function)

// "0-15,128-143" -> CPU set; returns the number of CPUs parsed
int parse_cpulist(const std::string& list, cpu_set_t* set)
{
    CPU_ZERO(set);
    int n = 0;
    size_t i = 0;
    while (i < list.size()) {
        char* end = nullptr;
        const long a = strtol(list.c_str() + i, &end, 10);
        if (end == list.c_str() + i) break;
        long b = a;
        i = (size_t)(end - list.c_str());
        if (i < list.size() && list[i] == '-') {
            b = strtol(list.c_str() + i + 1, &end, 10);
            i = (size_t)(end - list.c_str());
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0) { CPU_SET((int)c, set); n++; }
        if (i < list.size() && list[i] == ',') i++;
    }
    return n;
}

class HipBackend : public sched::Backend {
   public:
    explicit HipBackend(int device) : device_(device) {}
    void bind_thread() override { (void)hipSetDevice(phys_device(device_)); }
    // The launch worker of a device: one per GPU in a process that drives a whole node (SetGPUNum(8), the reference's
    // multi-GPU shape, test/test_gate_gpu_multi.cc:36-93).  Eight workers gathering ciphertexts into pinned memory and
    // submitting launches should each sit next to their GPU, not wherever the OS put them: pin to the GPU's local CPUs,
    // restricted to what the process is allowed to use (a container's cpuset); leave the thread alone if that is empty.
    int bind_worker_thread() override
    {
        bind_thread();
        if (!g_sched_affinity) return 0;
        cpu_set_t local, allowed, both;
        if (parse_cpulist(device_local_cpulist(phys_device(device_)), &local) == 0) return 0;
        if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return 0;
        CPU_AND(&both, &local, &allowed);
        const int n = CPU_COUNT(&both);
        if (n == 0 || n == CPU_COUNT(&allowed)) return 0;        // nothing to narrow
        if (pthread_setaffinity_np(pthread_self(), sizeof both, &both) != 0) return 0;
        return n;
    }
    int num_streams() override { return (int)(g_sched_streams < 1 ? 1 : g_sched_streams); }
    // one full-throughput round of the blind-rotate grid: a workgroup of 8 rotations per CU (capi.hip: launch_blind_rotate)
    size_t round_gates() override
    {
        int cus = 0;
        if (g_cus_override > 0) cus = (int)g_cus_override;
        else if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, phys_device(device_)) != hipSuccess) { (void)hipGetLastError(); cus = 0; }
        return (size_t)kBrWavesPerBlock * (size_t)(cus > 0 ? cus : 256);
    }
    int words(int level) override
    {
        if (g_param_set >= 0) return ps_ctxt_words((int)g_param_set, level);       // the active parameter set's sizes (2: a TRLWE, 3: a TRGSW in the NTT domain)
        return level == 3 ? (int)(2 * kBkStepDoubles) : slot_words(level);
    }
    // device slots are carved for the largest ciphertext of any compiled set, so "param_set" may change while ciphertexts live
    int slot_words(int level) override
    {
        return level == 0 ? kLvl0Words : level == 1 ? kLvl1Words : level == 2 ? 2 * kN : kMaxTrgswNttWords;     // 3: TRGSW, NTT domain (doubles, two words each)
    }
    int alloc_device(size_t bytes, void** p) override { return chk(hipMalloc(p, bytes), "hipMalloc"); }
    int free_device(void* p) override { return chk(hipFree(p), "hipFree"); }
    // The gather kernel writes results straight into this memory and the host reads them once the group's event has completed: that
    // needs host-coherent (fine-grained) pinned memory mapped into the device, asked for explicitly rather than left to what
    // hipHostMallocDefault happens to mean
    int alloc_pinned(size_t bytes, void** p) override { return chk(hipHostMalloc(p, bytes, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc"); }
    int free_pinned(void* p) override { return chk(hipHostFree(p), "hipHostFree"); }
    int h2d(int s, void* dst, const void* src, size_t bytes) override
    {
        hipStream_t st;
        if (int rc = stream(s, &st)) return rc;
        return chk(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st), "hipMemcpyAsync H2D");
    }
    int d2h(int s, void* dst, const void* src, size_t bytes) override
    {
        hipStream_t st;
        if (int rc = stream(s, &st)) return rc;
        return chk(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st), "hipMemcpyAsync D2H");
    }
    int copy_ctxts(int s, const sched::CopyRec* recs, size_t n, uint32_t* staging, bool to_ctxt) override
    {
        hipStream_t st;
        if (int rc = stream(s, &st)) return rc;
        DeviceState& ds = g_dev[device_];
        for (int level = 0; level < sched::kLevels; level++) {     // one lincomb (COPY) launch per ciphertext kind
            std::vector<LinDesc> d;
            for (size_t i = 0; i < n; i++)
                if (recs[i].level == level) {
                    uint32_t* slot = staging + recs[i].slot;
                    if (to_ctxt) d.push_back({slot, slot, recs[i].dev, 1, 0, 0u, 0u});
                    else d.push_back({recs[i].dev, recs[i].dev, slot, 1, 0, 0u, 0u});
                }
            if (d.empty()) continue;
            // the stream's workspace is reused from offset 0 by every launch sequence: the descriptor
            // copy of the next sequence is ordered behind the kernels of this one
            Scratch sc;
            if (int rc = open_scratch(ds, st, d.size() * sizeof(LinDesc) + 4096, &sc)) return keep(rc);
            LinDesc* dd;
            if (int rc = upload_descs(ds, sc, d, &dd)) return keep(rc);
            if (int rc = launch_lincomb(st, dd, d.size(), words(level))) return keep(rc);
        }
        return 0;
    }
    int run_gates(int s, int level, const sched::GateRef* g, size_t n) override
    {
        hipStream_t st;
        if (int rc = stream(s, &st)) return rc;
        if (level == 2) return keep(run_trlwe_ops(device_, (void*)st, g, n));
        return keep(::run_gates(device_, (void*)st, level, n, [&](size_t i) { return g[i]; }));
    }
    // Two lanes: measured on MI355X with both lanes running (tools/two_lane_probe.py, profiles/r06_two_lane_probe.txt) -- a step of the
    // paired low-latency kernel on half of the CUs, key switch included, 4.80 ms beside the bulk lane (4.59 alone); a chunk of the batch
    // kernel on the other half 18.6 ms (17.9 alone).  Half of the CUs each: an in-order stream then never has more workgroups in flight
    // than the other lane leaves free, so neither lane ever queues behind the other (full-width chunks beside the chain: 132 ms
    // against 79).  Only for the hand-scheduled BASELINE path: the parameter-set and N = 2048 paths pick their own shapes.
    bool lane_model(LaneModel* m) override
    {
        if (!g_sched_two_lane || g_param_set >= 0 || g_lvl0_ring != 1024) return false;
        const size_t cus = round_gates() / kBrWavesPerBlock;
        if (cus < 16) return false;
        m->chain_gates = 2 * (cus / 2);
        m->bulk_gates = (size_t)kBrWavesPerBlock * (cus / 2);
        m->chain_ms = 4.80;
        m->bulk_ms = 18.6;
        return true;
    }
    // one dependence level of n rotations by the rules of launch_blind_rotate, key switch and launch gaps included (MI355X, ms)
    double launch_ms(size_t n) override
    {
        if (n == 0) return 0.0;
        const size_t c = std::max<size_t>(1, round_gates() / kBrWavesPerBlock), round = (size_t)kBrWavesPerBlock * c;
        auto small = [&](size_t t) {
            if (t <= c) return 3.1;
            if (t > 6 * c) return 18.2;
            const size_t rem = t % (2 * c), paired = (rem == 0 || rem > c) ? t : t - rem;
            return (double)((paired + 2 * c - 1) / (2 * c)) * 5.0 + (paired < t ? 2.9 : 0.0);
        };
        const size_t full = n / round, tail = n % round;
        return (double)full * 18.25 + (tail ? small(tail) : 0.0);      // tools/tail_times.py, gpurun_out/r06_tail_times_ks.txt
    }
    int gate_weight(int op) override { return op == CUFHE_AMD_MUX || op == CUFHE_AMD_NMUX ? 2 : op == CUFHE_AMD_NOT || op == CUFHE_AMD_COPY ? 0 : 1; }
    int run_gates_lane(int s, int level, const sched::GateRef* g, size_t n, int lane) override
    {
        // chain lane: the paired low-latency kernel (the single one for at most a rotation per CU of its half); bulk lane: the batch kernel
        const size_t cus = round_gates() / kBrWavesPerBlock;
        g_br_shape = lane == 1 ? 1 : n <= cus / 2 ? 3 : 2;
        const int rc = run_gates(s, level, g, n);
        g_br_shape = 0;
        return rc;
    }
    int event_create(void** ev) override
    {
        hipEvent_t e;
        if (int rc = chk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return rc;
        *ev = (void*)e;
        return 0;
    }
    int event_destroy(void* ev) override { return chk(hipEventDestroy((hipEvent_t)ev), "hipEventDestroy"); }
    int event_record(int s, void* ev) override
    {
        hipStream_t st;
        if (int rc = stream(s, &st)) return rc;
        return chk(hipEventRecord((hipEvent_t)ev, st), "hipEventRecord");
    }
    int event_query(void* ev) override
    {
        const hipError_t e = hipEventQuery((hipEvent_t)ev);
        if (e == hipSuccess) return fault_or(1);
        if (e == hipErrorNotReady) return 0;
        return chk(e, "hipEventQuery");
    }
    int event_sync(void* ev) override
    {
        if (int rc = chk(hipEventSynchronize((hipEvent_t)ev), "hipEventSynchronize")) return rc;
        return fault_or(0);
    }
    int stream_wait(int s, void* ev) override
    {
        hipStream_t st;
        if (int rc = stream(s, &st)) return rc;
        return chk(hipStreamWaitEvent(st, (hipEvent_t)ev, 0), "hipStreamWaitEvent");
    }
    std::string error_text() override
    {
        std::lock_guard<std::mutex> lk(mu_);
        return err_;
    }
    // the caller's own stream (Stream::st()) behind one of our completion events / an internal stream behind the caller's stream
    int caller_stream_wait(void* caller_stream, void* ev) override
    {
        return chk(hipStreamWaitEvent((hipStream_t)caller_stream, (hipEvent_t)ev, 0), "hipStreamWaitEvent (caller's stream)");
    }
    int wait_for_caller_stream(int s, void* caller_stream) override
    {
        hipStream_t st;
        if (int rc = stream(s, &st)) return rc;
        // an event may be re-recorded while an earlier wait on it is pending: that wait keeps the state it captured
        hipEvent_t e;
        {
            std::lock_guard<std::mutex> lk(st_mu_);
            auto it = caller_ev_.find(caller_stream);
            if (it == caller_ev_.end()) {
                if (int rc = chk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return rc;
                caller_ev_[caller_stream] = e;
            } else e = it->second;
        }
        if (int rc = chk(hipEventRecord(e, (hipStream_t)caller_stream), "hipEventRecord (caller's stream)")) return rc;
        return chk(hipStreamWaitEvent(st, e, 0), "hipStreamWaitEvent");
    }
    void forget_caller_stream(void* caller_stream)
    {
        std::lock_guard<std::mutex> lk(st_mu_);
        auto it = caller_ev_.find(caller_stream);
        if (it == caller_ev_.end()) return;
        (void)hipEventDestroy(it->second);
        caller_ev_.erase(it);
    }
    // hipHostMalloc memory is mapped into the device's address space: the scatter / gather kernels use it in place
    void* device_alias(void* pinned) override
    {
        if (!g_sched_zero_copy) return nullptr;
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, pinned, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return d;
    }
    // timing marks of the scheduler's trace: only while the device is profiling (cufhe_amd_profile_enable)
    void* mark(int s) override
    {
        if (!g_dev[device_].profiling) return nullptr;
        hipStream_t st;
        if (stream(s, &st)) return nullptr;
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        if (hipEventRecord(e, st) != hipSuccess) { (void)hipEventDestroy(e); return nullptr; }
        return (void*)e;
    }
    float elapsed_ms(void* a, void* b) override
    {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) { (void)hipGetLastError(); return 0.0f; }
        return ms;
    }
    // the device must be idle: drop the internal streams (and their workspaces)
    void destroy_streams()
    {
        bind_thread();
        for (hipStream_t st : st_) {
            DeviceState& ds = g_dev[device_];
            {
                std::lock_guard<std::mutex> lk(ds.staging_mu);
                auto it = ds.workspaces.find(st);
                if (it != ds.workspaces.end()) {
                    (void)hipFree(it->second.base);
                    ds.workspaces.erase(it);
                }
            }
            (void)hipStreamDestroy(st);
        }
        st_.clear();
        for (auto& kv : caller_ev_) (void)hipEventDestroy(kv.second);
        caller_ev_.clear();
    }

   private:
    // completion observed: a kernel may have reported through the device's fault word (capi.hip: device_fault)
    int fault_or(int ok)
    {
        if (int rc = device_fault(device_)) return keep(rc);
        return ok;
    }
    int chk(hipError_t e, const char* what)
    {
        if (e == hipSuccess) return 0;
        std::lock_guard<std::mutex> lk(mu_);
        err_ = std::string(what) + ": " + hipGetErrorString(e);
        return -2;
    }
    int keep(int rc)       // the launch helpers report through the calling thread's g_err
    {
        if (rc) {
            std::lock_guard<std::mutex> lk(mu_);
            err_ = g_err;
        }
        return rc;
    }
    int stream(int s, hipStream_t* out)
    {
        std::lock_guard<std::mutex> lk(st_mu_);
        while ((int)st_.size() <= s) {
            hipStream_t st;
            if (int rc = chk(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate")) return rc;
            st_.push_back(st);
        }
        *out = st_[s];
        return 0;
    }
    int device_;
    std::unordered_map<void*, hipEvent_t> caller_ev_;     // one event per caller stream whose raw handle is out (guarded by st_mu_)
    std::vector<hipStream_t> st_;
    std::mutex mu_, st_mu_;
    std::string err_;
};

// The scheduler of the current SetGPUNum generation.  It is created on first use and lives until the
// GPU count (or device_base) changes; a scheduler that still owns live ciphertexts at that point is
// parked instead of destroyed, so that those ciphertexts can still be released.  Nothing is torn down
// from a static destructor: at process exit the HIP runtime may already be gone.
sched::Scheduler* g_scheduler = nullptr;
std::vector<HipBackend*> g_sched_backends;
std::vector<sched::Scheduler*> g_parked;
std::unordered_map<cufhe_amd_ctxt*, sched::Scheduler*> g_ctxt_owner;
std::mutex g_sched_mu;     // the reference API is single-issuer; this only guards against misuse

// the options that shape the scheduler, applied to the live one (flush rules in grid rounds of the device: nothing is a chip constant)
void sched_apply_settings()
{
    if (!g_scheduler) return;
    for (int d = 0; d < g_scheduler->gpu_num(); d++) {
        sched::DeviceSched& ds = g_scheduler->dev(d);
        ds.set_round_gates(ds.backend()->round_gates());
        if (g_sched_level_gates > 0) ds.set_level_flush_gates((size_t)g_sched_level_gates);
        if (g_sched_idle_gates > 0) ds.idle_flush_gates = std::min(ds.level_flush_gates, (size_t)g_sched_idle_gates);
        ds.total_flush_gates = (size_t)g_sched_total_gates;
        ds.rename_outputs = g_sched_rename != 0;
        ds.two_lane = (int)g_sched_two_lane;
        ds.copy_threads = (int)g_sched_copy_threads;
        ds.copy_op = CUFHE_AMD_COPY;
    }
}

sched::Scheduler* scheduler()
{
    if (!g_scheduler) {
        g_sched_backends.clear();
        g_scheduler = new sched::Scheduler(g_gpu_num, g_sched_threads != 0, [](int d) {
            HipBackend* b = new HipBackend(d);
            g_sched_backends.push_back(b);
            return b;
        });
        sched_apply_settings();
    }
    return g_scheduler;
}

bool sched_active() { return g_scheduler != nullptr; }

int sched_synchronize_all()
{
    if (!g_scheduler) return 0;
    if (int rc = g_scheduler->synchronize_all()) {
        for (int d = 0; d < g_scheduler->gpu_num(); d++)
            if (!g_scheduler->dev(d).error_text().empty()) return fail(rc, g_scheduler->dev(d).error_text());
        return fail(rc, "scheduler error");
    }
    return 0;
}

// CleanUp: everything recorded completes, cached staging buffers and the internal streams go; the
// scheduler itself (ciphertext slabs, workers) stays for the ciphertexts that outlive CleanUp.
void sched_quiesce()
{
    if (!g_scheduler) return;
    (void)g_scheduler->synchronize_all();
    for (int d = 0; d < g_scheduler->gpu_num(); d++) {
        g_scheduler->dev(d).backend()->bind_thread();
        g_scheduler->dev(d).release_buffers();
        if (g_scheduler->live_ctxts() == 0) g_scheduler->dev(d).release_slabs();
    }
    for (HipBackend* b : g_sched_backends) b->destroy_streams();
}

// SetGPUNum / device_base change: the next use builds a scheduler for the new device set
void sched_retire_generation()
{
    if (!g_scheduler) return;
    sched_quiesce();
    if (g_scheduler->live_ctxts() == 0) delete g_scheduler;
    else g_parked.push_back(g_scheduler);
    g_scheduler = nullptr;
    g_sched_backends.clear();
}

int sched_error(sched::DeviceSched& ds, int rc) { return fail(rc, ds.error_text().empty() ? "scheduler error" : ds.error_text()); }

}  // namespace

extern "C" {

int cufhe_amd_ctxt_create(int level, uint32_t* host_words, cufhe_amd_ctxt** out)
{
    if (level < 0 || level >= sched::kLevels) return fail(-1, "level must be 0, 1, 2 (TRLWE) or 3 (TRGSW in the NTT domain)");
    if (!host_words || !out) return fail(-1, "null pointer");
    std::lock_guard<std::mutex> lk(g_sched_mu);
    sched::Scheduler* S = scheduler();
    std::string err;
    if (int rc = S->ctxt_create(level, host_words, out, &err)) return fail(rc, err);
    g_ctxt_owner[*out] = S;
    return 0;
}

int cufhe_amd_ctxt_destroy(cufhe_amd_ctxt* c)
{
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(g_sched_mu);
    auto it = g_ctxt_owner.find(c);
    if (it == g_ctxt_owner.end()) return fail(-1, "unknown ciphertext handle");
    sched::Scheduler* S = it->second;
    g_ctxt_owner.erase(it);
    S->ctxt_destroy(c);       // no flush, no wait: the buffers are recycled when the last gate naming them retires
    return 0;
}

int cufhe_amd_ctxt_words(int level)
{
    if (level < 0 || level > 3) return fail(-1, "level must be 0, 1, 2 (a TRLWE) or 3 (a TRGSW in the NTT domain)");
    if (g_param_set >= 0) return ps_ctxt_words((int)g_param_set, level);
    return level == 3 ? (int)(2 * kBkStepDoubles) : level == 2 ? 2 * kN : level ? kLvl1Words : kLvl0Words;
}

uint32_t* cufhe_amd_ctxt_device_ptr(cufhe_amd_ctxt* c, int device)
{
    if (!c || device < 0 || device >= (int)c->d.size()) return nullptr;
    return c->d[device].home;       // the ciphertext's own buffer: holds its value whenever completion has been observed (sched_core.h)
}

static int sched_check_ctxt(sched::Scheduler* S, cufhe_amd_ctxt* c)
{
    if (c->owner != (void*)S) return fail(-1, "ciphertext was created before SetGPUNum changed the GPU set");
    if (c->destroyed) return fail(-1, "gate on a destroyed ciphertext");
    // the caller's host buffer has the size of the parameter set that was active when the ciphertext was created ("param_set"):
    // copies to and from it move the active set's size
    if (c->words != S->dev(0).backend()->words(c->level))
        return fail(-1, "ciphertext was created under another parameter set (\"param_set\"): its host buffer has a different size");
    return 0;
}

int cufhe_amd_enqueue_gate(int device, void* stream, int op, int copying, cufhe_amd_ctxt* out,
                           cufhe_amd_ctxt* in0, cufhe_amd_ctxt* in1, cufhe_amd_ctxt* in2)
{
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (int rc = check_device(device)) return rc;
    if (op < 0 || op >= CUFHE_AMD_NUM_OPS) return fail(-1, "unknown gate op");
    if (!out || !in0) return fail(-1, "null ciphertext");
    const bool three = op == CUFHE_AMD_MUX || op == CUFHE_AMD_NMUX;
    const bool one = op == CUFHE_AMD_NOT || op == CUFHE_AMD_COPY;
    if (!one && !in1) return fail(-1, "gate needs a second operand");
    if (three && !in2) return fail(-1, "mux needs a third operand");
    cufhe_amd_ctxt* ins[3] = {in0, one ? nullptr : in1, three ? in2 : nullptr};
    sched::Scheduler* S = scheduler();
    if (int rc = sched_check_ctxt(S, out)) return rc;
    if (out->level > 1) return fail(-1, "gates take lvl0 or lvl1 ciphertexts");
    for (cufhe_amd_ctxt* c : ins) {
        if (!c) continue;
        if (int rc = sched_check_ctxt(S, c)) return rc;
        if (c->level != out->level) return fail(-1, "operands of one gate must have the same level");
    }
    const DeviceState& ds = g_dev[device];
    if (!ds.keys_ready && !ds.keys2_ready && g_param_set < 0 && !one) return fail(-3, "Initialize(ek) has not been called for this device");
    if (int rc = S->dev(device).record_gate(stream, op, copying != 0, out, ins)) return sched_error(S->dev(device), rc);
    return 0;
}

/* gGateBootstrappingTLWE2TRLWElvl01NTT / gRefresh / gSampleExtractAndKeySwitch and their copying forms
 * (src/cufhe_gates_gpu.cu:86-146), recorded like gates: `out` / `in` are ciphertext handles of the levels the
 * operation maps between (level 2 = TRLWE). */
int cufhe_amd_enqueue_trlwe_op(int device, void* stream, int op, int copying, cufhe_amd_ctxt* out, cufhe_amd_ctxt* in)
{
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (int rc = check_device(device)) return rc;
    if (!out || !in) return fail(-1, "null ciphertext");
    int lin, lout;
    switch (op) {
        case CUFHE_AMD_TL_BOOTSTRAP: lin = 0; lout = 2; break;
        case CUFHE_AMD_TL_REFRESH: lin = 2; lout = 2; break;
        case CUFHE_AMD_TL_SEIKS: lin = 2; lout = 0; break;
        default: return fail(-1, "unknown TRLWE-level op");
    }
    sched::Scheduler* S = scheduler();
    if (int rc = sched_check_ctxt(S, out)) return rc;
    if (int rc = sched_check_ctxt(S, in)) return rc;
    if (in->level != lin || out->level != lout) return fail(-1, "operand levels do not fit the TRLWE-level operation");
    if (!g_dev[device].keys_ready && g_param_set < 0) return fail(-3, "Initialize(ek) has not been called for this device");
    cufhe_amd_ctxt* ins[3] = {in, nullptr, nullptr};
    if (int rc = S->dev(device).record_gate(stream, op, copying != 0, out, ins, 2)) return sched_error(S->dev(device), rc);
    return 0;
}

/* TRGSW2NTT (src/bootstrap_gpu.cu:75-94) on a TRGSW holder (ciphertext handle of level 3): the reference leaves the NTT-domain words
 * in trgswhost AND in trgswdevices[st.device_id()].  The holder's host words are rewritten here, synchronously, so the scheduler is
 * told first (recorded uploads of the old words read them before they change; snapshots of "the current trgswhost" are dropped), and
 * the upload of the new words to the stream's device is recorded like a CtxtCopyH2D: a following gCMUXNTT finds them in the device
 * buffer, a following CMUXNTT does not take an older upload for current. */
int cufhe_amd_trgsw_to_ntt(int device, void* stream, const uint32_t* trgsw_host, cufhe_amd_ctxt* out)
{
    sched::Scheduler* S;
    {
        std::lock_guard<std::mutex> lk(g_sched_mu);
        if (int rc = check_device(device)) return rc;
        if (!trgsw_host || !out) return fail(-1, "null pointer");
        S = scheduler();
        if (int rc = sched_check_ctxt(S, out)) return rc;
        if (out->level != 3) return fail(-1, "TRGSW2NTT needs a TRGSW holder (ciphertext handle of level 3)");
        if (int rc = S->before_direct_host_write(out)) return fail(rc, "scheduler: flushing a device ahead of TRGSW2NTT failed");
    }
    // the conversion takes the library's own lock (ensure_ntt): not under the scheduler's (SetGPUNum nests them the other way round)
    if (int rc = cufhe_amd_trgsw_to_ntt_host(device, stream, trgsw_host, reinterpret_cast<double*>(out->host))) return rc;
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (int rc = S->dev(device).record_copy(stream, out, true)) return sched_error(S->dev(device), rc);
    return 0;
}

/* CMUXNTT (src/cufhe_gates_gpu.cu:68-85, __CMUXNTT__ src/bootstrap_gpu.cu:197-285) recorded like a gate: res = c0 + cs [x] (c1 - c0).
 * copying != 0 is the reference's form (cs, c1, c0 taken from their host members in stream order, res delivered to
 * res.trlwehost); 0 uses device buffers only.  Needs Initialize() (the NTT tables), no evaluation key. */
int cufhe_amd_enqueue_cmux(int device, void* stream, int copying, cufhe_amd_ctxt* res, cufhe_amd_ctxt* cs, cufhe_amd_ctxt* c1,
                           cufhe_amd_ctxt* c0)
{
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = check_device(device)) return rc;
        HIP_TRY(hipSetDevice(phys_device(device)));
        if (int rc = ensure_ntt(device)) return rc;
    }
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (!res || !cs || !c1 || !c0) return fail(-1, "null operand");
    sched::Scheduler* S = scheduler();
    for (cufhe_amd_ctxt* c : {res, cs, c1, c0})
        if (int rc = sched_check_ctxt(S, c)) return rc;
    if (res->level != 2 || c1->level != 2 || c0->level != 2 || cs->level != 3) return fail(-1, "CMUXNTT takes TRLWEs and a TRGSW in the NTT domain");
    cufhe_amd_ctxt* ins[3] = {c1, c0, cs};
    if (int rc = S->dev(device).record_gate(stream, CUFHE_AMD_TL_CMUX, copying != 0, res, ins, 2)) return sched_error(S->dev(device), rc);
    return 0;
}

/* CtxtCopyH2D / CtxtCopyD2H (include/cufhe_gpu.cuh:193-207) in issue order.
 * to_device != 0: tlwehost -> device buffer; else device buffer -> tlwehost, visible after
 * Synchronize / StreamQuery like a gate result. */
int cufhe_amd_enqueue_copy(int device, void* stream, cufhe_amd_ctxt* c, int to_device)
{
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (int rc = check_device(device)) return rc;
    if (!c) return fail(-1, "null ciphertext");
    sched::Scheduler* S = scheduler();
    if (int rc = sched_check_ctxt(S, c)) return rc;
    if (int rc = S->dev(device).record_copy(stream, c, to_device != 0)) return sched_error(S->dev(device), rc);
    return 0;
}

int cufhe_amd_flush(int device)
{
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (int rc = check_device(device)) return rc;
    if (!g_scheduler) return 0;
    if (int rc = g_scheduler->dev(device).flush()) return sched_error(g_scheduler->dev(device), rc);
    return 0;
}

int cufhe_amd_sched_stream_query(int device, void* stream)
{
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (int rc = check_device(device)) return rc;
    if (!g_scheduler) return 1;
    const int q = g_scheduler->dev(device).stream_query(stream);
    if (q < 0) return sched_error(g_scheduler->dev(device), q);
    return q;
}

/* Stream::st() (include/cufhe_gpu.cuh:183): the raw handle is about to be handed to the caller */
int cufhe_amd_stream_fence(int device, void* stream)
{
    if (int rc = use_device(device)) return rc;
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (!stream) return 0;                       // the default stream is never handed out by class Stream
    sched::Scheduler* S = scheduler();
    if (int rc = S->dev(device).stream_fence(stream)) return sched_error(S->dev(device), rc);
    return 0;
}

int cufhe_amd_sched_get_stats(int device, cufhe_amd_sched_stats* out, int reset)
{
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (int rc = check_device(device)) return rc;
    if (!out) return fail(-1, "null");
    memset(out, 0, sizeof(*out));
    if (!g_scheduler) return 0;
    sched::Stats& s = g_scheduler->dev(device).stats();
    out->gates = s.gates; out->groups = s.groups; out->levels = s.levels; out->launch_sequences = s.launch_sequences;
    out->uploads = s.uploads; out->uploads_shared = s.uploads_shared; out->downloads = s.downloads;
    out->forced_syncs = s.forced_syncs; out->max_level_gates = s.max_level_gates; out->cross_stream_waits = s.cross_stream_waits;
    out->record_ns = s.record_ns; out->retire_ns = s.retire_ns; out->launch_ns = s.launch_ns.load();
    out->renames = s.renames;
    out->home_copies = s.home_copies;
    out->two_lane_groups = s.two_lane_groups.load();
    out->two_lane_launches = s.two_lane_launches.load();
    out->worker_cpus = s.worker_cpus.load();
    if (reset) {
        const uint64_t cpus = s.worker_cpus.load();      // a property of the worker thread, not a counter
        s = sched::Stats();
        s.worker_cpus.store(cpus);
    }
    return 0;
}

/* timeline of the most recent flushes (oldest first): what the PCIe-inclusive rate of the per-gate API is made of */
int cufhe_amd_sched_get_trace(int device, cufhe_amd_group_trace* out, int max, int clear)
{
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (int rc = check_device(device)) return rc;
    if (!out || max < 0) return fail(-1, "null / negative");
    if (!g_scheduler) return 0;
    static_assert(sizeof(cufhe_amd_group_trace) == sizeof(sched::GroupTrace), "the C struct mirrors sched::GroupTrace");
    return (int)g_scheduler->dev(device).get_trace(reinterpret_cast<sched::GroupTrace*>(out), (size_t)max, clear != 0);
}

}  // extern "C"
