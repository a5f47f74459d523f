// sched_harness.cpp -- the stream scheduler (cufhe_amd/csrc/sched_core.h) against a stubbed device layer.
//
// The stub is an asynchronous machine with the ordering rules of HIP streams and nothing more:
// work submitted to one internal stream runs in order, streams are ordered only by events, copies
// read their source when they EXECUTE, and execution happens at arbitrary later moments (a seeded
// random interleaving, advanced only when the scheduler queries or waits for an event).  "Gates" are
// a non-commutative word mixer, executed in random order within a launch -- so a scheduler that puts
// two dependent gates into one launch, lets a copy overtake a gate, or delivers a stale result fails
// the comparison with a plain in-order interpreter of the reference API's semantics
// (include/cufhe_gpu.cuh:193-313, src/cufhe_gates_gpu.cu:148-167).
//
// Usage: sched_harness <seeds> <gpus> <threaded 0|1>      prints one line per check, "ALL PASS" at the end.
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <random>

#include "../../cufhe_amd/csrc/sched_core.h"

using namespace cufhe_amd::sched;

static const int kWords[3] = {37, 53, 71};  // toy ciphertext sizes (odd on purpose): lvl0, lvl1, TRLWE
enum { OP_NOT = 12, OP_COPY = 13, OP_MUX = 10, OP_NMUX = 11, TL_BOOT = 100, TL_REFRESH = 101, TL_SEIKS = 102,
       OP_HOME_COPY = 99 };     // DeviceSched::copy_op here: the scheduler's own copy of a renamed value back to the ciphertext's buffer (the user-visible ops, Copy included, are word mixers)
static int tl_in_level(int op) { return op == TL_BOOT ? 0 : 2; }
static int tl_out_level(int op) { return op == TL_SEIKS ? 0 : 2; }

static inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t mix(int op, uint32_t a, uint32_t b, uint32_t c, uint32_t w)
{
    uint32_t v = a * 0x9E3779B1u + rotl(b, 5) * 0x85EBCA77u + rotl(c, 11) * 0xC2B2AE3Du + (uint32_t)op * 0x27D4EB2Fu + w;
    return v ^ (a >> 15) ^ (v << 7);
}
static void toy_gate(int op, int level, uint32_t* out, const uint32_t* a, const uint32_t* b, const uint32_t* c)
{
    if (op == OP_HOME_COPY) { memmove(out, a, kWords[level] * 4); return; }
    if (op >= TL_BOOT) {      // TRLWE-level operations map between ciphertext kinds
        const int wi = kWords[tl_in_level(op)], wo = kWords[tl_out_level(op)];
        std::vector<uint32_t> r(wo);
        for (int w = 0; w < wo; w++) r[w] = mix(op, a[w % wi], a[(w * 7 + 3) % wi], 0u, (uint32_t)w);
        memcpy(out, r.data(), r.size() * 4);
        return;
    }
    std::vector<uint32_t> r(kWords[level]);
    for (int w = 0; w < kWords[level]; w++) r[w] = mix(op, a[w], b ? b[w] : 0u, c ? c[w] : 0u, (uint32_t)w);
    memcpy(out, r.data(), r.size() * 4);
}

static bool g_two_lane = true;     // argv[6]: flushes of several levels are scheduled gate by gate on two lanes when eligible (DeviceSched::compile_two_lane)
static bool g_zero_copy = false;   // argv[5]: the stub's pinned memory is "visible to the device" (Backend::device_alias): no staging copies

struct FakeEvent { uint64_t submitted = 0, completed = 0; };

class FakeBackend : public Backend {
   public:
    FakeBackend(int nstreams, uint64_t seed) : q_(nstreams), rng_(seed) {}
    void bind_thread() override {}
    int num_streams() override { return (int)q_.size(); }
    int words(int level) override { return kWords[level]; }
    int alloc_device(size_t bytes, void** p) override { *p = calloc(1, bytes); return 0; }
    int free_device(void* p) override { free(p); return 0; }
    int alloc_pinned(size_t bytes, void** p) override { *p = malloc(bytes); memset(*p, 0xAB, bytes); return 0; }
    int free_pinned(void* p) override { free(p); return 0; }
    void* device_alias(void* pinned) override { return g_zero_copy ? pinned : nullptr; }
    int h2d(int s, void* dst, const void* src, size_t bytes) override { push(s, [=] { memcpy(dst, src, bytes); }); return 0; }
    int d2h(int s, void* dst, const void* src, size_t bytes) override { push(s, [=] { memcpy(dst, src, bytes); }); return 0; }
    int copy_ctxts(int s, const CopyRec* recs, size_t n, uint32_t* staging, bool to_ctxt) override
    {
        std::vector<CopyRec> r(recs, recs + n);
        push(s, [=] {
            for (const CopyRec& c : r) {
                if (to_ctxt) memcpy(c.dev, staging + c.slot, kWords[c.level] * 4);
                else memcpy(staging + c.slot, c.dev, kWords[c.level] * 4);
            }
        });
        return 0;
    }
    int run_gates(int s, int level, const GateRef* g, size_t n) override
    {
        std::vector<GateRef> v(g, g + n);
        {
            std::lock_guard<std::mutex> lk(mu_);      // the launch worker and the issuing thread (event_query) share rng_
            launches++;
            std::shuffle(v.begin(), v.end(), rng_);
        }
        push(s, [=] {
            for (const GateRef& r : v) toy_gate(r.op, level, r.out, r.in0, r.in1, r.in2);
        });
        return 0;
    }
    int event_create(void** ev) override { *ev = new FakeEvent(); return 0; }
    int event_destroy(void* ev) override
    {
        // as in HIP, a wait that was enqueued on the event stays valid when the event object goes: the scheduler only destroys
        // completed events, so such a wait is satisfied
        std::lock_guard<std::mutex> lk(mu_);
        FakeEvent* e = (FakeEvent*)ev;
        auto release = [&](std::deque<Op>& q) {
            for (Op& op : q)
                if (op.wait == e) {
                    if (e->completed < op.target) { fprintf(stderr, "stub device: an event with a pending, unsatisfied wait was destroyed\n"); abort(); }
                    op.wait = nullptr;
                }
        };
        for (auto& q : q_) release(q);
        for (auto& kv : cq_) release(kv.second);
        delete e;
        return 0;
    }
    int event_record(int s, void* ev) override
    {
        std::lock_guard<std::mutex> lk(mu_);
        FakeEvent* e = (FakeEvent*)ev;
        e->submitted++;
        q_[s].push_back({[e] { e->completed++; }, nullptr, 0});
        return 0;
    }
    int stream_wait(int s, void* ev) override
    {
        std::lock_guard<std::mutex> lk(mu_);
        FakeEvent* e = (FakeEvent*)ev;
        q_[s].push_back({[] {}, e, e->submitted});
        return 0;
    }
    int event_query(void* ev) override
    {
        std::lock_guard<std::mutex> lk(mu_);
        const int n = (int)(rng_() % 4);
        for (int i = 0; i < n; i++) pump_one();
        FakeEvent* e = (FakeEvent*)ev;
        return e->completed == e->submitted ? 1 : 0;
    }
    int event_sync(void* ev) override
    {
        std::lock_guard<std::mutex> lk(mu_);
        FakeEvent* e = (FakeEvent*)ev;
        while (e->completed != e->submitted)
            if (!pump_one()) { fprintf(stderr, "stub device: event can never complete (missing dependence?)\n"); abort(); }
        return 0;
    }
    std::string error_text() override { return "stub"; }
    // two lanes: a toy model under which the per-gate plan wins whenever a flush is eligible -- the harness is after its ORDER, not its timing
    bool lane_model(LaneModel* m) override
    {
        if (!g_two_lane) return false;
        m->chain_gates = 3; m->bulk_gates = 8; m->chain_ms = 1.0; m->bulk_ms = 3.0;
        return true;
    }
    double launch_ms(size_t n) override { return n ? 1000.0 : 0.0; }
    int gate_weight(int op) override { return op == OP_MUX || op == OP_NMUX ? 2 : op == OP_HOME_COPY ? 0 : 1; }
    int run_gates_lane(int s, int level, const GateRef* g, size_t n, int lane) override
    {
        { std::lock_guard<std::mutex> lk(mu_); lane_launches[lane == 0 ? 0 : 1]++; }
        return run_gates(s, level, g, n);
    }
    uint64_t lane_launches[2] = {0, 0};
    // the caller's own streams (raw handles of Stream::st()): queues like the internal ones, filled by the test itself
    int caller_stream_wait(void* cs, void* ev) override
    {
        std::lock_guard<std::mutex> lk(mu_);
        FakeEvent* e = (FakeEvent*)ev;
        caller_waits++;
        cq_[cs].push_back({[] {}, e, e->submitted});
        return 0;
    }
    int wait_for_caller_stream(int s, void* cs) override
    {
        std::lock_guard<std::mutex> lk(mu_);
        FakeEvent* e = new FakeEvent();
        owned_.push_back(e);
        e->submitted++;
        cq_[cs].push_back({[e] { e->completed++; }, nullptr, 0});
        q_[s].push_back({[] {}, e, e->submitted});
        return 0;
    }
    void push_caller(void* cs, std::function<void()> fn)
    {
        std::lock_guard<std::mutex> lk(mu_);
        cq_[cs].push_back({std::move(fn), nullptr, 0});
    }
    ~FakeBackend() override { for (FakeEvent* e : owned_) delete e; }
    uint64_t caller_waits = 0;
    void drain()
    {
        std::lock_guard<std::mutex> lk(mu_);
        while (pump_one()) {}
    }
    uint64_t launches = 0;

   private:
    struct Op { std::function<void()> fn; FakeEvent* wait; uint64_t target; };
    void push(int s, std::function<void()> fn)
    {
        std::lock_guard<std::mutex> lk(mu_);
        q_[s].push_back({std::move(fn), nullptr, 0});
    }
    bool pump_one()
    {
        std::vector<std::deque<Op>*> ready;
        auto is_ready = [](std::deque<Op>& q) { return !q.empty() && (!q.front().wait || q.front().wait->completed >= q.front().target); };
        for (auto& q : q_)
            if (is_ready(q)) ready.push_back(&q);
        for (auto& kv : cq_)
            if (is_ready(kv.second)) ready.push_back(&kv.second);
        if (ready.empty()) return false;
        std::deque<Op>* q = ready[rng_() % ready.size()];
        Op op = std::move(q->front());
        q->pop_front();
        op.fn();
        return true;
    }
    std::mutex mu_;
    std::vector<std::deque<Op>> q_;
    std::map<void*, std::deque<Op>> cq_;
    std::vector<FakeEvent*> owned_;
    std::mt19937_64 rng_;
};

// ---- in-order interpreter of the reference semantics ----
struct ModelCtxt {
    int level;
    std::vector<uint32_t> host;                 // what tlwehost will eventually hold
    std::vector<std::vector<uint32_t>> dev;     // per device
    std::vector<bool> dev_defined;
};

static bool g_rename = false;      // argv[4]: outputs take fresh device buffers instead of waiting (DeviceSched::rename_outputs)

struct Test {
    int G;
    Scheduler* S;
    std::vector<FakeBackend*> be;
    std::mt19937_64 rng;
    struct C {
        cufhe_amd_ctxt* h; std::vector<uint32_t> host; ModelCtxt m; void* last_host_writer_stream = nullptr; bool alive = true;
        std::vector<uint32_t*> home;                   // the device pointers as of creation: what the reference publishes as tlwedevices
        std::vector<void*> last_dev_writer_stream;     // per device: the caller stream of the newest gate that wrote the device value
        std::vector<void*> last_dev_upload_stream;     // per device: the caller stream of an upload that is the NEWEST write of the device value (nullptr: a gate's write is)
        std::vector<void*> only_stream;                // per device: the one caller stream all recorded accesses were issued on (nullptr: none yet, kMany: several)
        // a ciphertext the program only ever touches on ONE stream of one device: what a caller may read and write through that stream's raw
        // handle with a defined outcome (across streams the reference orders nothing)
        void* priv_stream = nullptr;
        int priv_dev = -1;
    };
    static void* many() { return (void*)(uintptr_t)-1; }
    static void touch(C* c, int dev, void* st) { void*& o = c->only_stream[dev]; o = (o == nullptr || o == st) ? st : many(); }
    // reads the caller enqueued on a raw handle: what the buffer must hold when the read executes
    struct CallerRead { std::vector<uint32_t> got, want; bool done = false; };
    std::vector<std::shared_ptr<CallerRead>> caller_reads;
    std::vector<C*> ct;
    int failures = 0;

    Test(int gpus, bool threaded, uint64_t seed, int nstreams) : G(gpus), rng(seed)
    {
        S = new Scheduler(gpus, threaded, [&](int d) {
            FakeBackend* b = new FakeBackend(nstreams, seed * 131 + d);
            be.push_back(b);
            return b;
        });
        for (int d = 0; d < gpus; d++) {
            S->dev(d).rename_outputs = g_rename; S->dev(d).two_lane = g_two_lane ? 1 : 0; S->dev(d).copy_op = OP_HOME_COPY;
            S->dev(d).parallel_copy_min = seed % 2 ? 3 : 1024;       // every other program: the gathers and deliveries split over the copy threads
        }
    }
    ~Test()
    {
        S->synchronize_all();
        for (C* c : ct) { if (c->alive) S->ctxt_destroy(c->h); delete c; }
        delete S;
    }
    C* make(int level)
    {
        C* c = new C();
        c->host.resize(kWords[level]);
        for (auto& w : c->host) w = (uint32_t)rng();
        std::string err;
        if (S->ctxt_create(level, c->host.data(), &c->h, &err)) { fprintf(stderr, "ctxt_create: %s\n", err.c_str()); abort(); }
        c->m.level = level;
        c->m.host = c->host;
        c->m.dev.assign(G, std::vector<uint32_t>(kWords[level], 0));
        c->m.dev_defined.assign(G, false);
        for (int d = 0; d < G; d++) c->home.push_back(c->h->d[d].dev);
        c->last_dev_writer_stream.assign(G, nullptr);
        c->only_stream.assign(G, nullptr);
        c->last_dev_upload_stream.assign(G, nullptr);
        ct.push_back(c);
        return c;
    }
    bool trace = getenv("SCHED_HARNESS_TRACE") != nullptr;
    int idx(C* c) { for (size_t i = 0; i < ct.size(); i++) if (ct[i] == c) return (int)i; return -1; }
    void gate(int dev, void* st, int op, bool copying, C* out, C* a, C* b, C* c3)
    {
        if (trace) printf("T gate dev %d st %p op %d %s out %d in %d %d %d\n", dev, st, op, copying ? "copying" : "g", idx(out), idx(a), b ? idx(b) : -1, c3 ? idx(c3) : -1);
        C* ins[3] = {a, b, c3};
        cufhe_amd_ctxt* hs[3] = {a->h, b ? b->h : nullptr, c3 ? c3->h : nullptr};
        if (int rc = S->dev(dev).record_gate(st, op, copying, out->h, hs, op >= TL_BOOT ? 2 : -1)) { fprintf(stderr, "record_gate rc %d: %s\n", rc, S->dev(dev).error_text().c_str()); abort(); }
        for (C* i : ins)
            if (i) touch(i, dev, st);
        touch(out, dev, st);
        // model, in issue order
        if (copying)
            for (C* i : ins)
                if (i) { i->m.dev[dev] = i->m.host; i->m.dev_defined[dev] = true; i->last_dev_upload_stream[dev] = st; }
        std::vector<uint32_t> r(kWords[out->m.level]);
        toy_gate(op, out->m.level, r.data(), a->m.dev[dev].data(), b ? b->m.dev[dev].data() : nullptr, c3 ? c3->m.dev[dev].data() : nullptr);
        out->m.dev[dev] = r;
        out->m.dev_defined[dev] = true;
        out->last_dev_writer_stream[dev] = st;
        out->last_dev_upload_stream[dev] = nullptr;
        if (copying) { out->m.host = r; out->last_host_writer_stream = st; }
    }
    void copy(int dev, void* st, C* c, bool to_device)
    {
        if (trace) printf("T copy dev %d st %p %s ctxt %d\n", dev, st, to_device ? "H2D" : "D2H", idx(c));
        if (int rc = S->dev(dev).record_copy(st, c->h, to_device)) { fprintf(stderr, "record_copy rc %d\n", rc); abort(); }
        touch(c, dev, st);
        if (to_device) { c->m.dev[dev] = c->m.host; c->m.dev_defined[dev] = true; c->last_dev_upload_stream[dev] = st; }
        else { c->m.host = c->m.dev[dev]; c->last_host_writer_stream = st; }
    }
    // Stream::st(): the raw handle is handed out (stream_fence), then the caller puts its own work on it.
    // A read of a ciphertext's OWN device buffer (the published tlwedevices pointer) must see what the gates issued on `st` before
    // the call left there; a write into it must be seen by the g-gates issued on `st` afterwards (src/cufhe_gates_gpu.cu:148-167: in
    // the reference all of this is one stream's order).
    void caller_read(int dev, void* st, C* c)
    {
        if (trace) printf("T caller_read dev %d st %p ctxt %d\n", dev, st, idx(c));
        if (int rc = S->dev(dev).stream_fence(st)) { fprintf(stderr, "stream_fence rc %d\n", rc); abort(); }
        auto r = std::make_shared<CallerRead>();
        r->want = c->m.dev[dev];
        const uint32_t* home = c->home[dev];
        const int words = kWords[c->m.level];
        be[dev]->push_caller(st, [r, home, words] { r->got.assign(home, home + words); r->done = true; });
        caller_reads.push_back(r);
    }
    void caller_write(int dev, void* st, C* c)
    {
        if (trace) printf("T caller_write dev %d st %p ctxt %d\n", dev, st, idx(c));
        if (int rc = S->dev(dev).stream_fence(st)) { fprintf(stderr, "stream_fence rc %d\n", rc); abort(); }
        std::vector<uint32_t> w(kWords[c->m.level]);
        for (auto& x : w) x = (uint32_t)rng();
        uint32_t* home = c->home[dev];
        be[dev]->push_caller(st, [w, home] { memcpy(home, w.data(), w.size() * 4); });
        c->m.dev[dev] = w;
        c->m.dev_defined[dev] = true;
        c->last_dev_writer_stream[dev] = st;
        c->last_dev_upload_stream[dev] = nullptr;
        touch(c, dev, st);
    }
    void check_caller_reads()
    {
        for (auto& r : caller_reads) {
            if (!r->done) { failures++; printf("FAIL a read on the caller's stream never ran\n"); continue; }
            if (r->got != r->want) { failures++; printf("FAIL a read on the caller's stream (raw handle of Stream::st()) saw stale words\n"); }
        }
        caller_reads.clear();
    }
    void check_host(C* c, const char* when)
    {
        if (c->host != c->m.host) {
            failures++;
            printf("FAIL host mismatch (%s) ctxt %p [%d] level %d\n", when, (void*)c, idx(c), c->m.level);
        }
    }
    // StreamQuery(st) returned true: what gates on st wrote must be in the ciphertexts' OWN device buffers (the pointers
    // published at creation), renaming or not -- and so must a value that an upload on st refreshed after a gate on ANOTHER stream
    // had renamed the ciphertext away from its buffer (the upload lands in the renamed buffer).
    void check_home_after_query(int dev, void* st)
    {
        for (auto* b : be) b->drain();
        for (C* c : ct) {
            if (!c->alive || !c->m.dev_defined[dev]) continue;
            if (c->last_dev_upload_stream[dev] ? c->last_dev_upload_stream[dev] != st : c->last_dev_writer_stream[dev] != st) continue;
            if (c->h->d[dev].dev != c->home[dev]) { failures++; printf("FAIL ctxt %p [%d] still renamed after StreamQuery (dev %d stream %p)\n", (void*)c, idx(c), dev, st); }
        }
    }
    void sync_and_check(bool check_dev)
    {
        if (trace) printf("T synchronize\n");
        if (int rc = S->synchronize_all()) { fprintf(stderr, "synchronize rc %d\n", rc); abort(); }
        for (auto* b : be) b->drain();
        check_caller_reads();
        for (C* c : ct) {
            if (!c->alive) continue;
            check_host(c, "after Synchronize");
            if (check_dev)
                for (int d = 0; d < G; d++)
                    if (c->m.dev_defined[d] && (c->h->d[d].dev != c->home[d] || memcmp(c->home[d], c->m.dev[d].data(), kWords[c->m.level] * 4))) {
                        failures++;
                        printf("FAIL device mismatch ctxt %p [%d] dev %d%s\n", (void*)c, idx(c), d, c->h->d[d].dev != c->home[d] ? " (value not in the ciphertext's own buffer)" : "");
                    }
        }
    }
};

static int random_program(uint64_t seed, int gpus, bool threaded)
{
    std::mt19937_64 prng(seed);
    Test t(gpus, threaded, seed, 1 + (int)(prng() % 4));
    for (int d = 0; d < gpus; d++) {
        t.S->dev(d).set_round_gates(8 + prng() % 32);        // (the idle rule: one round)
        t.S->dev(d).set_level_flush_gates(4 + prng() % 40);  // exercise partial flushes
        t.S->dev(d).total_flush_gates = 30 + prng() % 200;
    }
    const int kStreams = 6;
    const bool use_raw_handles = seed % 3 == 0;     // (a handle that is out switches the shared-upload shortcut off: not in every program)
    auto stream_handle = [&](int dev, int s) { return (void*)(uintptr_t)(0x1000 + dev * 64 + s); };
    for (int i = 0; i < 30; i++) t.make(i % 5 == 4 ? 2 : i % 3 == 0 ? 1 : 0);
    if (use_raw_handles)
        for (int i = 0; i < 12; i++) {
            Test::C* c = t.make(i % 3 == 0 ? 1 : 0);
            c->priv_dev = (int)(prng() % gpus);
            c->priv_stream = stream_handle(c->priv_dev, (int)(prng() % 2));     // two streams per device carry the private ciphertexts
        }
    const int steps = 300 + (int)(prng() % 500);
    int cur_dev = 0;
    void* cur_st = nullptr;
    auto usable = [&](Test::C* c) { return c->alive && (!c->priv_stream || (c->priv_stream == cur_st && c->priv_dev == cur_dev)); };
    auto pick = [&](int level) {
        for (;;) {
            Test::C* c = t.ct[prng() % t.ct.size()];
            if (usable(c) && c->m.level == level) return c;
        }
    };
    auto pick_defined = [&](int level, int dev) -> Test::C* {
        for (int tries = 0; tries < 64; tries++) {
            Test::C* c = t.ct[prng() % t.ct.size()];
            if (usable(c) && c->m.level == level && c->m.dev_defined[dev]) return c;
        }
        return nullptr;
    };
    for (int step = 0; step < steps; step++) {
        const int dev = (int)(prng() % gpus);
        void* st = stream_handle(dev, (int)(prng() % kStreams));
        cur_dev = dev;
        cur_st = st;
        const int level = prng() % 4 == 0 ? 1 : 0;
        const unsigned r = (unsigned)(prng() % 100);
        if (use_raw_handles && prng() % 16 == 0) {  // the caller takes the raw handle of the stream and reads / writes a device buffer on it
            Test::C* c = nullptr;
            for (int tries = 0; tries < 64 && !c; tries++) {
                Test::C* x = t.ct[prng() % t.ct.size()];
                if (x->alive && x->priv_stream == st && x->priv_dev == dev && x->m.dev_defined[dev]) c = x;
            }
            if (!c) continue;
            if (prng() % 2) t.caller_read(dev, st, c);
            else t.caller_write(dev, st, c);
            continue;
        }
        if (r < 8) {                                // TRLWE-level operation (copying or device-resident)
            const int op = TL_BOOT + (int)(prng() % 3);
            const bool copying = prng() % 2 == 0;
            Test::C* in = copying ? pick(tl_in_level(op)) : pick_defined(tl_in_level(op), dev);
            if (!in) continue;
            Test::C* out = (tl_in_level(op) == tl_out_level(op) && prng() % 4 == 0) ? in : pick(tl_out_level(op));
            t.gate(dev, st, op, copying, out, in, nullptr, nullptr);
        } else if (r < 40) {                        // copying gate
            const int kind = (int)(prng() % 10);
            Test::C* out = pick(level);
            Test::C* a = prng() % 5 == 0 ? out : pick(level);      // in-place now and then
            if (kind == 0) t.gate(dev, st, prng() % 2 ? OP_NOT : OP_COPY, true, out, a, nullptr, nullptr);
            else if (kind <= 2) t.gate(dev, st, prng() % 2 ? OP_MUX : OP_NMUX, true, out, a, pick(level), pick(level));
            else t.gate(dev, st, (int)(prng() % 10), true, out, a, prng() % 7 == 0 ? a : pick(level), nullptr);
        } else if (r < 65) {                        // g-gate on device-resident values
            Test::C *a = pick_defined(level, dev), *b = pick_defined(level, dev), *c3 = pick_defined(level, dev);
            if (!a || !b || !c3) continue;
            Test::C* out = prng() % 4 == 0 ? a : pick(level);
            const int kind = (int)(prng() % 10);
            if (kind == 0) t.gate(dev, st, OP_NOT, false, out, a, nullptr, nullptr);
            else if (kind <= 2) t.gate(dev, st, OP_MUX, false, out, a, b, c3);
            else t.gate(dev, st, (int)(prng() % 10), false, out, a, b, nullptr);
        } else if (r < 72) {
            t.copy(dev, st, pick(level), true);
        } else if (r < 80) {
            Test::C* c = pick_defined(level, dev);
            if (c) t.copy(dev, st, c, false);
        } else if (r < 92) {                        // StreamQuery poll
            const int q = t.S->dev(dev).stream_query(st);
            if (q < 0) { fprintf(stderr, "stream_query rc %d\n", q); abort(); }
            if (q == 1) {
                for (Test::C* c : t.ct)
                    if (c->alive && c->last_host_writer_stream == st) t.check_host(c, "after StreamQuery");
                t.check_home_after_query(dev, st);
            }
        } else if (r < 95) {                        // Synchronize, then the host may edit ciphertexts
            t.sync_and_check(true);
            for (int k = 0; k < 3; k++) {
                Test::C* c = t.ct[prng() % t.ct.size()];
                if (!c->alive) continue;
                for (auto& w : c->host) w = (uint32_t)prng();
                c->m.host = c->host;
            }
        } else if (r < 96) {                        // the caller rewrites a tlwehost itself, without Synchronize (TRGSW2NTT on a holder)
            Test::C* c = t.ct[prng() % t.ct.size()];
            if (!c->alive) continue;
            if (int rc = t.S->before_direct_host_write(c->h)) { fprintf(stderr, "before_direct_host_write rc %d\n", rc); abort(); }
            for (auto* b : t.be) b->drain();
            t.check_host(c, "before a direct host write");      // a result that was on its way has landed
            for (auto& w : c->host) w = (uint32_t)prng();
            c->m.host = c->host;
        } else if (r < 97) {                        // StreamSynchronize(st): everything issued on st is complete and delivered
            if (int rc = t.S->dev(dev).stream_synchronize(st)) { fprintf(stderr, "stream_synchronize rc %d\n", rc); abort(); }
            for (Test::C* c : t.ct)
                if (c->alive && c->last_host_writer_stream == st) t.check_host(c, "after StreamSynchronize");
            t.check_home_after_query(dev, st);
        } else if (r < 98) {                        // a ciphertext goes out of scope while work on it is recorded
            Test::C* c = t.ct[prng() % t.ct.size()];
            if (!c->alive || c->priv_stream) continue;
            t.S->ctxt_destroy(c->h);
            c->alive = false;
            for (auto& w : c->host) w = 0xDEADBEEFu;      // the caller's memory is gone: nobody may read it any more
            t.make(c->m.level);
        } else {
            t.S->dev(dev).flush();
        }
    }
    t.sync_and_check(true);
    return t.failures;
}

// ---- random single-kind netlists: what the per-gate (two-lane) order of a flush is made for ----
// One device, level-0 ciphertexts only, inputs uploaded by the first gates that read them, then a few hundred gates whose operands are
// drawn from everything computed so far: deep chains beside wide independent work, temporaries re-used (renamed), in-place gates,
// copying and device-resident gates mixed, Mux gates (two rotations), explicit D2H copies, several streams; Synchronize once at the end
// (or in the middle, so that a second flush depends on the first).  Every value is compared with the in-order interpreter.
static uint64_t g_two_lane_groups = 0, g_two_lane_launches = 0;
static int dag_program(uint64_t seed, bool threaded)
{
    std::mt19937_64 prng(seed);
    Test t(1, threaded, seed, 2 + (int)(prng() % 3));
    t.S->dev(0).set_round_gates(100000);            // nothing is launched before the Synchronize: the whole netlist is one flush
    t.S->dev(0).total_flush_gates = 1u << 30;
    const int inputs = 6 + (int)(prng() % 10), temps = 4 + (int)(prng() % 12), gates = 60 + (int)(prng() % 400);
    std::vector<Test::C*> pool;
    for (int i = 0; i < inputs + temps; i++) pool.push_back(t.make(0));
    std::vector<bool> defined(pool.size(), false);
    auto st = [&](int i) { return (void*)(uintptr_t)(0x500 + i % 5); };
    // the inputs reach the device with the first copying gates
    for (int i = 0; i < inputs; i += 2) {
        Test::C* out = pool[inputs + (i / 2) % temps];
        t.gate(0, st(i), (int)(prng() % 10), true, out, pool[i], pool[(i + 1) % inputs], nullptr);
        defined[i] = defined[(i + 1) % inputs] = true;
        defined[inputs + (i / 2) % temps] = true;
    }
    const int mid_sync = prng() % 3 == 0 ? (int)(prng() % gates) : -1;
    for (int k = 0; k < gates; k++) {
        auto pick_def = [&]() {
            for (;;) {
                const size_t i = prng() % pool.size();
                if (defined[i]) return i;
            }
        };
        const size_t a = pick_def(), b = pick_def(), c = pick_def();
        const bool chain = prng() % 3 != 0;              // most gates extend what was just computed: long chains
        static size_t last = 0;
        const size_t in0 = chain && defined[last] ? last : a;
        size_t o = inputs + prng() % temps;
        if (prng() % 6 == 0) o = in0;                    // in place
        const unsigned r = (unsigned)(prng() % 20);
        if (r == 0) t.gate(0, st(k), OP_NOT, false, pool[o], pool[in0], nullptr, nullptr);
        else if (r <= 2) t.gate(0, st(k), prng() % 2 ? OP_MUX : OP_NMUX, false, pool[o], pool[in0], pool[b], pool[c]);
        else if (r == 3) t.copy(0, st(k), pool[in0], false);
        else if (r == 4 && pool[in0]->m.dev[0] == pool[in0]->m.host) t.gate(0, st(k), (int)(prng() % 10), true, pool[o], pool[in0], pool[b], nullptr);   // a copying gate whose re-upload is the device value
        else t.gate(0, st(k), (int)(prng() % 10), false, pool[o], pool[in0], pool[b], nullptr);
        if (r != 3) { defined[o] = true; last = o; }
        if (k == mid_sync) t.sync_and_check(true);
    }
    for (size_t i = 0; i < pool.size(); i++)
        if (defined[i] && prng() % 2) t.copy(0, st((int)i), pool[i], false);      // fetch some of the results
    t.sync_and_check(true);
    g_two_lane_groups += t.S->dev(0).stats().two_lane_groups.load();
    g_two_lane_launches += t.S->dev(0).stats().two_lane_launches.load();
    return t.failures;
}

// ---- shaped programs with launch-count expectations ----
struct Shape { const char* name; uint64_t launch_sequences, levels, gates, groups, uploads, uploads_shared; int failures; uint64_t two_lane_groups = 0, two_lane_launches = 0; };

static Shape chained(bool threaded)
{
    // test/test_api_gpu.cu:140-159 as tests/cpp/test_gate_api.cpp Chained(): 64 chains x 5 in-place gates on 8 streams
    Test t(1, threaded, 7, 4);
    const int K = 64;
    std::vector<Test::C*> a, b, c;
    for (int i = 0; i < K; i++) { a.push_back(t.make(0)); b.push_back(t.make(0)); c.push_back(t.make(0)); }
    for (int i = 0; i < K; i++) {
        void* st = (void*)(uintptr_t)(0x100 + i % 8);
        t.gate(0, st, 0, true, a[i], a[i], b[i], nullptr);
        t.gate(0, st, 4, true, a[i], a[i], b[i], nullptr);
        t.gate(0, st, 5, true, a[i], a[i], c[i], nullptr);
        t.gate(0, st, OP_NOT, true, a[i], a[i], nullptr, nullptr);
        t.gate(0, st, OP_MUX, true, a[i], a[i], b[i], c[i]);
    }
    t.sync_and_check(true);
    const Stats& s = t.S->dev(0).stats();
    return {"chained", s.launch_sequences, s.levels, s.gates, s.groups, s.uploads, s.uploads_shared, t.failures};
}

static Shape ripple(bool threaded)
{
    // tests/cpp/test_gate_api.cpp RippleAdders(): 16 x 8-bit ripple-carry adders, one stream each, issued bit by bit
    Test t(1, threaded, 8, 4);
    const int A = 16, B = 8;
    std::vector<Test::C*> x, y, sum, carry, t1, t2;
    for (int i = 0; i < A * B; i++) { x.push_back(t.make(0)); y.push_back(t.make(0)); sum.push_back(t.make(0)); }
    for (int i = 0; i < A; i++) { carry.push_back(t.make(0)); t1.push_back(t.make(0)); t2.push_back(t.make(0)); }
    for (int k = 0; k < B; k++)
        for (int i = 0; i < A; i++) {
            void* st = (void*)(uintptr_t)(0x200 + i);
            Test::C *X = x[i * B + k], *Y = y[i * B + k], *Sm = sum[i * B + k], *Cy = carry[i];
            t.gate(0, st, 5, true, t1[i], X, Y, nullptr);
            t.gate(0, st, 5, true, Sm, t1[i], Cy, nullptr);
            t.gate(0, st, 3, true, t2[i], t1[i], Cy, nullptr);
            t.gate(0, st, 3, true, t1[i], X, Y, nullptr);
            t.gate(0, st, 4, true, Cy, t1[i], t2[i], nullptr);
        }
    t.sync_and_check(true);
    const Stats& s = t.S->dev(0).stats();
    return {"ripple_adders", s.launch_sequences, s.levels, s.gates, s.groups, s.uploads, s.uploads_shared, t.failures, s.two_lane_groups.load(), s.two_lane_launches.load()};
}

static Shape intensive(bool threaded)
{
    // test/test_intensive.cc:21-128: poll StreamQuery and refill, three inputs shared by every stream
    Test t(1, threaded, 9, 4);
    const int kStreams = 200, kRounds = 4;
    Test::C *in0 = t.make(0), *in1 = t.make(0), *inc = t.make(0);
    std::vector<Test::C*> out;
    std::vector<int> round(kStreams, 0);
    for (int i = 0; i < kStreams; i++) out.push_back(t.make(0));
    auto st = [](int i) { return (void*)(uintptr_t)(0x300 + i); };
    for (int i = 0; i < kStreams; i++) t.gate(0, st(i), 0, true, out[i], in0, in1, nullptr);
    int done = 0;
    long polls = 0;
    while (done < kStreams && polls < 100000000) {
        for (int i = 0; i < kStreams; i++) {
            polls++;
            if (round[i] >= kRounds) continue;
            if (t.S->dev(0).stream_query(st(i)) != 1) continue;
            t.check_host(out[i], "intensive");
            if (++round[i] == kRounds) { done++; continue; }
            if (round[i] % 2) t.gate(0, st(i), OP_MUX, true, out[i], inc, in1, in0);
            else t.gate(0, st(i), 0, true, out[i], in0, in1, nullptr);
        }
    }
    if (done < kStreams) { t.failures++; printf("FAIL intensive never completed\n"); }
    t.sync_and_check(true);
    const Stats& s = t.S->dev(0).stats();
    return {"intensive", s.launch_sequences, s.levels, s.gates, s.groups, s.uploads, s.uploads_shared, t.failures};
}

static Shape multi_gpu(bool threaded)
{
    // test/test_gate_gpu_multi.cc:36-93: default-constructed streams round-robin the devices
    // (include/cufhe_gpu.cuh:154-159); every gate must land on its stream's device only
    const int G = 3, K = 96;
    Test t(G, threaded, 10, 2);
    std::vector<Test::C*> a, b, o;
    for (int i = 0; i < K; i++) { a.push_back(t.make(0)); b.push_back(t.make(0)); o.push_back(t.make(0)); }
    for (int i = 0; i < K; i++) t.gate(i % G, (void*)(uintptr_t)(0x400 + i), 0, true, o[i], a[i], b[i], nullptr);
    // a second round consuming results produced on ANOTHER device: the host copy has to land first
    for (int i = 0; i < K; i++) t.gate((i + 1) % G, (void*)(uintptr_t)(0x400 + i), 5, true, a[i], o[i], b[i], nullptr);
    t.sync_and_check(true);
    Shape sh{"multi_gpu", 0, 0, 0, 0, 0, 0, t.failures};
    for (int d = 0; d < G; d++) {
        const Stats& s = t.S->dev(d).stats();
        if (s.gates != 2 * K / G) { sh.failures++; printf("FAIL device %d recorded %llu gates, expected %d\n", d, (unsigned long long)s.gates, 2 * K / G); }
        sh.launch_sequences += s.launch_sequences; sh.levels += s.levels; sh.gates += s.gates; sh.groups += s.groups;
        sh.uploads += s.uploads; sh.uploads_shared += s.uploads_shared;
    }
    return sh;
}

// ---- host cost of the issuing thread at the real ciphertext size (no device work at all) ----
class NullBackend : public Backend {
   public:
    void bind_thread() override {}
    int num_streams() override { return 4; }
    int words(int level) override { return level ? 1025 : 631; }
    int alloc_device(size_t bytes, void** p) override { *p = malloc(bytes); return 0; }
    int free_device(void* p) override { free(p); return 0; }
    int alloc_pinned(size_t bytes, void** p) override { *p = malloc(bytes); return 0; }
    int free_pinned(void* p) override { free(p); return 0; }
    int h2d(int, void*, const void*, size_t) override { return 0; }
    int d2h(int, void*, const void*, size_t) override { return 0; }
    int copy_ctxts(int, const CopyRec*, size_t, uint32_t*, bool) override { return 0; }
    int run_gates(int, int, const GateRef*, size_t) override { return 0; }
    int event_create(void** ev) override { *ev = (void*)1; return 0; }
    int event_destroy(void*) override { return 0; }
    int event_record(int, void*) override { return 0; }
    int event_query(void*) override { return 1; }
    int event_sync(void*) override { return 0; }
    int stream_wait(int, void*) override { return 0; }
    std::string error_text() override { return ""; }
};

// The benchmark's dependent netlist (tools/bench_api.cpp: 256 x 16-bit ripple-carry adders, issued adder by adder) against a device
// that does nothing but carries the HIP backend's cost model (sched_hip.inc.h: lane_model, launch_ms for 256 CUs): with
// CUFHE_AMD_SCHED_DEBUG=1 the scheduler prints what the level order and the two-lane plan are estimated to cost -- the plan can be
// tuned here, on the CPU.
class ModelBackend : public NullBackend {
   public:
    bool lane_model(LaneModel* m) override { m->chain_gates = 256; m->bulk_gates = 1024; m->chain_ms = 4.80; m->bulk_ms = 18.6; return true; }
    double launch_ms(size_t n) override
    {
        if (n == 0) return 0.0;
        const size_t c = 256, round = 8 * c;
        auto small = [&](size_t t) {
            if (t <= c) return 3.1;
            if (t > 6 * c) return 18.2;
            const size_t rem = t % (2 * c), paired = (rem == 0 || rem > c) ? t : t - rem;
            return (double)((paired + 2 * c - 1) / (2 * c)) * 5.0 + (paired < t ? 2.9 : 0.0);
        };
        const size_t full = n / round, tail = n % round;
        return (double)full * 18.25 + (tail ? small(tail) : 0.0);
    }
    int gate_weight(int op) override { return op == 10 || op == 11 ? 2 : op == 12 || op == 13 ? 0 : 1; }
    uint64_t lanes[3] = {0, 0, 0};
    int run_gates_lane(int, int, const GateRef*, size_t, int lane) override { lanes[lane == 0 ? 0 : 1]++; return 0; }
    int run_gates(int, int, const GateRef*, size_t) override { lanes[2]++; return 0; }
};

static void plan_netlist(int adders, int bits)
{
    ModelBackend* mb = nullptr;
    Scheduler S(1, true, [&](int) { mb = new ModelBackend(); return mb; });
    S.dev(0).set_round_gates(2048);
    std::vector<std::vector<uint32_t>> host;
    std::vector<cufhe_amd_ctxt*> c;
    std::string err;
    auto make = [&]() { host.emplace_back(631, 7u); c.push_back(nullptr); S.ctxt_create(0, host.back().data(), &c.back(), &err); return c.back(); };
    host.reserve((size_t)adders * (3 * bits + 3) + 8);
    std::vector<cufhe_amd_ctxt*> x, y, sum, carry, t1, t2;
    for (int i = 0; i < adders * bits; i++) { x.push_back(make()); y.push_back(make()); sum.push_back(make()); }
    for (int i = 0; i < adders; i++) { carry.push_back(make()); t1.push_back(make()); t2.push_back(make()); }
    auto gate = [&](void* st, int op, cufhe_amd_ctxt* o, cufhe_amd_ctxt* a, cufhe_amd_ctxt* b) {
        cufhe_amd_ctxt* ins[3] = {a, b, nullptr};
        S.dev(0).record_gate(st, op, true, o, ins);
    };
    for (int i = 0; i < adders; i++) {
        void* st = (void*)(uintptr_t)(0x1000 + i % 256);
        for (int k = 0; k < bits; k++) {
            cufhe_amd_ctxt *X = x[i * bits + k], *Y = y[i * bits + k], *Sm = sum[i * bits + k], *C = carry[i];
            gate(st, 5, t1[i], X, Y);
            gate(st, 5, Sm, t1[i], C);
            gate(st, 3, t2[i], t1[i], C);
            gate(st, 3, t1[i], X, Y);
            gate(st, 4, C, t1[i], t2[i]);
        }
    }
    S.synchronize_all();
    const Stats& st = S.dev(0).stats();
    printf("PLAN {\"adders\": %d, \"bits\": %d, \"gates\": %llu, \"levels\": %llu, \"two_lane_groups\": %llu, \"chain_steps\": %llu, \"bulk_chunks\": %llu, \"other_launches\": %llu}\n",
           adders, bits, (unsigned long long)st.gates, (unsigned long long)st.levels, (unsigned long long)st.two_lane_groups.load(),
           (unsigned long long)mb->lanes[0], (unsigned long long)mb->lanes[1], (unsigned long long)mb->lanes[2]);
    for (auto* k : c) S.ctxt_destroy(k);
}

static void host_cost(int gpus, int gates_per_gpu)
{
    // what tools/bench_api.cpp does, for `gpus` devices from one issuing thread: Nand(out, a, b, st) on
    // host-resident lvl0 ciphertexts over 256 streams per device, then Synchronize()
    Scheduler S(gpus, true, [](int) { return new NullBackend(); });
    const int total = gpus * gates_per_gpu, W = 631;
    std::vector<std::vector<uint32_t>> host(3 * (size_t)total, std::vector<uint32_t>(W, 7u));
    std::vector<cufhe_amd_ctxt*> c(3 * (size_t)total);
    std::string err;
    for (size_t i = 0; i < c.size(); i++) S.ctxt_create(0, host[i].data(), &c[i], &err);
    double best = 1e30, best_rec = 0;
    for (int rep = 0; rep < 5; rep++) {
        for (auto& h : host) h[0]++;          // fresh inputs every round
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < total; i++) {
            cufhe_amd_ctxt* ins[3] = {c[3 * (size_t)i + 1], c[3 * (size_t)i + 2], nullptr};
            S.dev(i % gpus).record_gate((void*)(uintptr_t)(0x1000 + (i / gpus) % 256), 0, true, c[3 * (size_t)i], ins);
        }
        auto t1 = std::chrono::steady_clock::now();
        S.synchronize_all();
        auto t2 = std::chrono::steady_clock::now();
        const double tot = std::chrono::duration<double, std::micro>(t2 - t0).count();
        if (tot < best) { best = tot; best_rec = std::chrono::duration<double, std::micro>(t1 - t0).count(); }
    }
    // busy time of the issuing thread (recording + delivering results) and of the busiest launch worker, from the
    // scheduler's own clocks over all five rounds: wall time above also contains waiting for the workers
    uint64_t rec = 0, ret = 0, worker = 0, gates = 0;
    for (int d = 0; d < gpus; d++) {
        const Stats& st = S.dev(d).stats();
        rec += st.record_ns; ret += st.retire_ns; gates += st.gates;
        worker = std::max<uint64_t>(worker, st.launch_ns.load());
    }
    printf("HOSTCOST {\"gpus\": %d, \"gates\": %d, \"wall_us_per_gate\": %.4f, \"enqueue_wall_us_per_gate\": %.4f, "
           "\"issuing_thread_us_per_gate\": %.4f, \"issuing_thread_gates_per_s\": %.0f, \"worker_us_per_gate_per_device\": %.4f}\n",
           gpus, total, best / total, best_rec / total, (rec + ret) * 1e-3 / gates, gates / ((rec + ret) * 1e-9),
           worker * 1e-3 / (gates / gpus));
    for (auto* x : c) S.ctxt_destroy(x);
}

int main(int argc, char** argv)
{
    if (argc > 1 && !strcmp(argv[1], "plan")) {
        plan_netlist(argc > 2 ? atoi(argv[2]) : 256, argc > 3 ? atoi(argv[3]) : 16);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "hostcost")) {
        host_cost(1, 4096);
        host_cost(8, 4096);
        return 0;
    }
    const int seeds = argc > 1 ? atoi(argv[1]) : 50;
    const int gpus = argc > 2 ? atoi(argv[2]) : 2;
    const bool threaded = argc > 3 ? atoi(argv[3]) != 0 : true;
    g_rename = argc > 4 ? atoi(argv[4]) != 0 : false;
    g_zero_copy = argc > 5 ? atoi(argv[5]) != 0 : false;
    g_two_lane = argc > 6 ? atoi(argv[6]) != 0 : true;
    int failures = 0;
    if (const char* one = getenv("SCHED_HARNESS_SEED")) {      // one random program, for debugging: SCHED_HARNESS_SEED=1038 sched_harness 0 3 1
        const int sd = atoi(one);
        if (sd >= 5000) {
            const int f = dag_program(sd, threaded);
            printf("netlist seed %d: %d mismatches\n", sd, f);
            return f ? 1 : 0;
        }
        const int f = random_program(sd, 1 + ((sd - 1000) % gpus), threaded);
        printf("seed %d: %d mismatches\n", sd, f);
        return f ? 1 : 0;
    }
    for (int s = 1; s <= seeds; s++) {
        const int f = random_program(1000 + s, 1 + (s % gpus), threaded);
        if (f) printf("FAIL random program seed %d: %d mismatches\n", 1000 + s, f);
        failures += f;
    }
    printf("random programs: %d seeds, %d failures\n", seeds, failures);
    int dag_failures = 0;
    for (int s = 1; s <= seeds; s++) {
        const int f = dag_program(5000 + s, threaded);
        if (f) printf("FAIL random netlist seed %d: %d mismatches\n", 5000 + s, f);
        dag_failures += f;
    }
    printf("random netlists: %d seeds, %d failures, %llu flushes scheduled gate by gate on two lanes (%llu launches)\n", seeds, dag_failures,
           (unsigned long long)g_two_lane_groups, (unsigned long long)g_two_lane_launches);
    failures += dag_failures;
    for (Shape sh : {chained(threaded), ripple(threaded), intensive(threaded), multi_gpu(threaded)}) {
        printf("SHAPE {\"name\": \"%s\", \"launch_sequences\": %llu, \"levels\": %llu, \"gates\": %llu, \"groups\": %llu, "
               "\"uploads\": %llu, \"uploads_shared\": %llu, \"failures\": %d, \"two_lane_groups\": %llu, \"two_lane_launches\": %llu}\n",
               sh.name, (unsigned long long)sh.launch_sequences, (unsigned long long)sh.levels, (unsigned long long)sh.gates,
               (unsigned long long)sh.groups, (unsigned long long)sh.uploads, (unsigned long long)sh.uploads_shared, sh.failures,
               (unsigned long long)sh.two_lane_groups, (unsigned long long)sh.two_lane_launches);
        failures += sh.failures;
    }
    printf("%s\n", failures ? "FAILED" : "ALL PASS");
    return failures ? 1 : 0;
}
