// sched_core.h -- the level-wise stream scheduler behind the reference's per-gate API.
//
// The reference launches one kernel per gate on the caller's stream, bracketed by three small
// memcpys (src/cufhe_gates_gpu.cu:148-158); throughput comes from hundreds of concurrent streams
// (test/test_util.h:36-62) and dependent gates are chained by stream order
// (test/test_api_gpu.cu:140-159, include/cufhe_gpu.cuh:282-313).  Here a gate call only RECORDS
// the gate, together with its data dependences; the recorded program of a device is kept as
// DEPENDENCE LEVELS, and level k of all streams is launched together: one blind-rotate launch and
// one key-switch launch per level and ciphertext kind, however the caller interleaved its streams.
//
// What a caller of the reference API can observe is unchanged:
//   * the result of a program is the result of executing its calls in issue order (the reference
//     promises this per stream and leaves cross-stream conflicts undefined; the single issuing
//     thread of its API makes issue order a legal execution of every race-free program);
//   * `out.tlwehost` holds a gate's result once Synchronize() returned or StreamQuery(st) returned
//     true for the stream the gate was issued on;
//   * non-g gates take their inputs from `tlwehost` in stream order: the memory, or, when an earlier
//     gate's result is still on its way to that `tlwehost`, that result -- the reference's D2H of
//     the earlier gate precedes this gate's H2D on the stream.  As in the reference (whose H2D is an
//     asynchronous copy from the pinned `tlwehost`, src/cufhe_gates_gpu.cu:148-158) the memory is
//     read some time after the call, by the device's launch thread: the caller must not modify an
//     input before it has observed completion of the gates reading it.  A ciphertext may be
//     DESTROYED right after the call (its words are saved first);
//   * g-gates touch device buffers only; CtxtCopyH2D / CtxtCopyD2H move data in issue order;
//   * `out` may alias an input (test/test_api_gpu.cu:141); inputs may be shared by any number of
//     gates on any streams (test/test_intensive.cc:103-107).
//
// Nothing in this file calls HIP: the device is reached through `Backend`, so the scheduler is
// exercised on the CPU against a stubbed device layer (tests/host/sched_harness.cpp), including
// SetGPUNum(G > 1) routing.  capi.hip supplies the HIP backend.
//
// Threads: one issuing thread (the reference's contract) records; each device has a worker thread
// that turns flushed levels into launches, so recording for device B overlaps launching on device
// A and one process can feed every GPU of a node.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace cufhe_amd {
namespace sched {

struct GateRef { int op; uint32_t* out; const uint32_t* in0; const uint32_t* in1; const uint32_t* in2; };
// Ciphertext kinds ("levels"): 0 = lvl0 TLWE, 1 = lvl1 TLWE, 2 = lvl1 TRLWE (the cuFHETRLWElvl1 of
// include/cufhe_gpu.cuh:124-134).  Gates are launched per kind of their OUTPUT, except the TRLWE-level
// operations (bootstrap to TRLWE, Refresh, SampleExtractAndKeySwitch), which mix kinds and form kind 2.
constexpr int kKinds = 3;
// Buffer kinds a ciphertext handle can have: the three above plus 3 = a TRGSW in the NTT domain (struct
// cuFHETRGSWNTTlvl1, include/cufhe_gpu.cuh:136-146), which is only ever an INPUT (of CMUXNTT, a kind-2 operation).
constexpr int kLevels = 4;
struct CopyRec { uint32_t* dev; size_t slot; int level; };      // staging word offset <-> a ciphertext's device buffer

// The device layer of ONE device.  Every method returns 0 or a negative status (text through
// error_text()).  `s` is the index of one of the backend's internal streams; work submitted to one
// internal stream executes in submission order, different streams are ordered only by events.
class Backend {
   public:
    virtual ~Backend() {}
    virtual void bind_thread() = 0;                                   // make this device current in the calling thread
    // the same for the device's launch worker, which may also be placed on the CPUs close to the device;
    // returns the number of CPUs the thread was pinned to (0: left to the OS)
    virtual int bind_worker_thread() { bind_thread(); return 0; }
    virtual int num_streams() = 0;
    // gates of one full-throughput launch of this device (a grid round: 8 rotations per compute unit on the HIP backend); the
    // scheduler's flush rules are multiples of it
    virtual size_t round_gates() { return 2048; }
    virtual int words(int level) = 0;                                 // words of a level-`level` ciphertext (what copies move)
    virtual int slot_words(int level) { return words(level); }        // words a device slot must hold (>= words(level) at any time)
    virtual int alloc_device(size_t bytes, void** p) = 0;
    virtual int free_device(void* p) = 0;
    virtual int alloc_pinned(size_t bytes, void** p) = 0;
    virtual int free_pinned(void* p) = 0;
    virtual int h2d(int s, void* dst, const void* src, size_t bytes) = 0;
    virtual int d2h(int s, void* dst, const void* src, size_t bytes) = 0;
    // staging[rec.slot ..] -> rec.dev (to_ctxt) or the reverse, `words(rec.level)` words each
    virtual int copy_ctxts(int s, const CopyRec* recs, size_t n, uint32_t* staging, bool to_ctxt) = 0;
    // n independent gates on ciphertexts of one level: no gate of the call reads what another writes
    virtual int run_gates(int s, int level, const GateRef* g, size_t n) = 0;
    virtual int event_create(void** ev) = 0;
    virtual int event_destroy(void* ev) = 0;
    virtual int event_record(int s, void* ev) = 0;
    virtual int event_query(void* ev) = 0;                            // 1 complete, 0 not yet
    virtual int event_sync(void* ev) = 0;
    virtual int stream_wait(int s, void* ev) = 0;
    virtual std::string error_text() = 0;
    // optional device-side timing marks (a timing event recorded on stream s, or nullptr when the backend is not
    // profiling) and the milliseconds between two of them once both have completed; marks are freed with event_destroy
    // the address under which the device reads / writes a block from alloc_pinned directly, or nullptr if it cannot: the scheduler
    // then skips the staging copies (h2d / d2h) and lets copy_ctxts work on the pinned block itself
    virtual void* device_alias(void* /*pinned*/) { return nullptr; }
    // The CALLER's own stream (the raw handle the reference hands out as Stream::st(), include/cufhe_gpu.cuh:183).  In the reference a
    // gate IS enqueued on that stream (src/cufhe_gates_gpu.cu:148-167), so whatever the caller puts on it afterwards runs behind the
    // gate and whatever it put there before runs ahead of it; here gates run on internal streams, and these two calls restore that
    // order when the caller asks for the handle (DeviceSched::stream_fence): the caller's stream waits for one of our events /
    // internal stream `s` waits for everything the caller's stream holds now.
    // Two lanes (DeviceSched::compile_two_lane).  A flush whose levels are narrow costs a launch sequence per level however few
    // gates it holds; when the device can run a grid on half of its compute units beside another one, the gates of a flush can instead
    // be scheduled one by one: the CHAIN lane runs steps of up to chain_gates gates on a low-latency shape, the BULK lane chunks of up
    // to bulk_gates on the throughput shape, on two internal streams.  lane_model returns false when the backend cannot (the default).
    // launch_ms(n): what one level of n gates costs the level-by-level order.  gate_weight(op): blind rotations of a gate (0: linear).
    struct LaneModel { size_t chain_gates = 0, bulk_gates = 0; double chain_ms = 0, bulk_ms = 0; };
    virtual bool lane_model(LaneModel* /*m*/) { return false; }
    virtual double launch_ms(size_t /*n*/) { return 0.0; }
    virtual int gate_weight(int /*op*/) { return 1; }
    // the same as run_gates with the launch shape of a lane: 0 = chain, 1 = bulk (anything else: by the backend's own rules)
    virtual int run_gates_lane(int s, int level, const GateRef* g, size_t n, int /*lane*/) { return run_gates(s, level, g, n); }
    virtual int caller_stream_wait(void* /*caller_stream*/, void* /*ev*/) { return 0; }
    virtual int wait_for_caller_stream(int /*s*/, void* /*caller_stream*/) { return 0; }
    virtual void* mark(int /*s*/) { return nullptr; }
    virtual float elapsed_ms(void* /*a*/, void* /*b*/) { return 0.0f; }
};

// What happened to one flushed group, for the timeline of a run (cufhe_amd_sched_get_trace): host times are
// steady_clock nanoseconds (std::chrono::steady_clock::now().time_since_epoch()), device spans milliseconds from timing
// events (0 unless the backend was profiling).
struct GroupTrace {
    uint64_t id = 0;
    uint32_t levels = 0, gates = 0, stream = 0, pad = 0;
    uint64_t in_bytes = 0, out_bytes = 0;
    int64_t t_queued = 0;          // issuing thread: the group was handed to the launch worker
    int64_t t_launch_begin = 0;    // worker: picked up
    int64_t t_gather_end = 0;      // worker: inputs copied out of tlwehost into the pinned block
    int64_t t_submit_end = 0;      // worker: everything submitted to the stream
    int64_t t_done_seen = 0;       // issuing thread: completion observed (event synchronised / queried true)
    int64_t t_delivered = 0;       // issuing thread: results copied into tlwehost, buffers recycled
    float dev_h2d_ms = 0, dev_body_ms = 0, dev_d2h_ms = 0;     // H2D copy + scatter | gates | gather + D2H copy
};
inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace sched
}  // namespace cufhe_amd

// template<class P> struct Ctxt of the reference (include/cufhe_gpu.cuh:102-121): caller-owned
// `tlwehost` plus one device buffer per GPU, and the bookkeeping that orders recorded accesses.
struct cufhe_amd_ctxt {
    struct PerDev {
        uint32_t* dev = nullptr;    // the buffer that holds (or will hold) the current value
        // Renaming (DeviceSched::rename_outputs): `home` is the buffer the ciphertext was created with -- the pointer the
        // reference publishes as Ctxt::tlwedevices[i] (include/cufhe_gpu.cuh:80-84).  While dev != home the value lives in a
        // renamed buffer; it is copied back before the host can observe completion (restore_homes), so that `home` holds
        // the value whenever the caller may look.  home_deps: the recorded levels that still name `home` (the write and the
        // reads of the value it held when the ciphertext was renamed away from it).
        uint32_t* home = nullptr;
        std::vector<uint32_t> home_deps;
        void* wstream = nullptr;    // caller stream of the newest recorded gate write (completion of that stream must cover the copy back)
        // caller stream(s) that (re-)uploaded the value SINCE that write (CtxtCopyH2D / a non-g gate's input, recorded or recognised as
        // already there): completion observed on any of them, too, must find the value in the home buffer
        void* ustream = nullptr;
        bool umany = false;         // more than one such stream: every completion restores
        void note_upload_stream(void* st) { if (!ustream) ustream = st; else if (ustream != st) umany = true; }
        int renamed_idx = -1;       // position in DeviceSched::renamed_
        uint32_t ready = 0;         // depth from which a gate may read `dev` (0: resident since long)
        uint32_t wdepth = 0;        // depth of the newest recorded write of `dev` (0: none on record)
        bool w_upload = false;      // ... which was an upload (runs before the gates of its level)
        uint32_t widx = 0;          // ... else the position of the writing gate in that level's gate list of kind wkind
        uint8_t wkind = 0;
        uint64_t version = 0;       // bumps at every recorded write of `dev`
        uint32_t last_use = 0;      // newest level naming this buffer in any way (keeps a destroyed ciphertext alive)
        // levels that read `dev` since that write: a few inline, the rest (rare) in a vector
        uint32_t rd[4] = {0, 0, 0, 0};
        uint32_t nrd = 0;
        std::vector<uint32_t> rd_more;
        // the newest upload host -> dev, kept to recognise an unchanged shared input
        void* snap_plan = nullptr;   // the (unretired) level whose staging block holds that upload
        size_t snap_off = 0;
        uint64_t snap_version = 0;
        uint32_t snap_hits = 0;
        uint32_t last_upload = 0;   // depth of the newest recorded upload (its copy out of tlwehost happens at launch)
        bool snap_owned = false;
        std::vector<uint32_t> snap_own;
    };
    int level = 0;
    int words = 0;                  // words of the caller's host buffer: those of a level-`level` ciphertext when it was created
    uint32_t* host = nullptr;       // nullptr once the caller destroyed the ciphertext
    void* owner = nullptr;          // the Scheduler that created it
    std::atomic<int> host_reads{0};             // recorded uploads whose copy out of `host` is still to be done
    std::atomic<uint32_t*> shadow{nullptr};     // the words of a destroyed ciphertext, for those copies
    bool destroyed = false;
    // a result on its way to `host`: produced on device host_dev as version host_version
    int host_dev = -1;
    uint64_t host_version = 0;
    uint64_t host_token = 0;
    void* host_stream = nullptr;    // ... recorded on this caller stream,
    uint32_t host_epoch = 0;        // whose raw handle had been handed out this often by then (DeviceSched::stream_fence)
    std::vector<PerDev> d;
};

namespace cufhe_amd {
namespace sched {

struct Stats {
    uint64_t gates = 0;               // gates recorded
    uint64_t groups = 0;              // flushes handed to the device
    uint64_t levels = 0;              // dependence levels launched
    uint64_t launch_sequences = 0;    // run_gates calls = (level, ciphertext kind) pairs with gates
    uint64_t uploads = 0, uploads_shared = 0, downloads = 0;
    uint64_t forced_syncs = 0;        // a stale `tlwehost` had to be waited for (see resolve_host)
    uint64_t max_level_gates = 0;
    uint64_t cross_stream_waits = 0;
    uint64_t renames = 0;             // outputs that took a fresh device buffer instead of waiting for the old one's users
    uint64_t home_copies = 0;         // renamed values copied back to the ciphertext's own buffer before the host could look
    std::atomic<uint64_t> two_lane_groups{0}, two_lane_launches{0};      // flushes scheduled gate by gate on two lanes, and their launches
    std::atomic<uint64_t> worker_cpus{0};   // CPUs the launch worker is pinned to
    // host time: on the issuing thread (recording, delivering results) and on the launch worker
    uint64_t record_ns = 0, retire_ns = 0;
    std::atomic<uint64_t> launch_ns{0};
    Stats() {}
    Stats(const Stats& o) { *this = o; }
    Stats& operator=(const Stats& o)
    {
        gates = o.gates; groups = o.groups; levels = o.levels; launch_sequences = o.launch_sequences;
        uploads = o.uploads; uploads_shared = o.uploads_shared; downloads = o.downloads; forced_syncs = o.forced_syncs;
        max_level_gates = o.max_level_gates; cross_stream_waits = o.cross_stream_waits; renames = o.renames; home_copies = o.home_copies;
        record_ns = o.record_ns; retire_ns = o.retire_ns;
        launch_ns.store(o.launch_ns.load());
        worker_cpus.store(o.worker_cpus.load());
        two_lane_groups.store(o.two_lane_groups.load());
        two_lane_launches.store(o.two_lane_launches.load());

        return *this;
    }
};
struct ScopedNs {
    uint64_t* acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit ScopedNs(uint64_t* a) : acc(a) {}
    ~ScopedNs() { *acc += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

struct Delivery { cufhe_amd_ctxt* c; size_t slot; uint64_t token; };

// The producers of a gate's operands inside the recorded program: (level, index in that level's gate list of the same kind), level 0 =
// the operand is resident (an upload, or a value produced before the program on record).  What per-gate scheduling of a flush needs
// (DeviceSched::compile_two_lane); the level-by-level launch order does not look at it.
struct GateDep {
    uint32_t depth[3] = {0, 0, 0};
    uint32_t idx[3] = {0, 0, 0};
    uint8_t kind[3] = {0, 0, 0};
    // the scheduler's own copy of a renamed value back to its ciphertext's buffer (restore_homes): besides its operand it must follow
    // every recorded reader of the value the home buffer held -- the level it was placed at does; a per-gate order runs it last
    bool home_copy = false;
};

struct Plan {                         // one dependence level of the recorded program
    uint32_t depth = 0;
    std::vector<GateRef> gates[kKinds];
    std::vector<GateDep> deps[kKinds];             // parallel to gates
    bool level_ordered = false;                    // some gate of this level relies on level order for a buffer hazard (it could not be renamed)
    std::vector<CopyRec> uploads, downloads;
    std::vector<cufhe_amd_ctxt*> upload_ctxts;     // parallel to uploads
    std::vector<Delivery> deliveries;              // parallel to downloads
    size_t in_words = 0;                           // staging words of the uploads (filled by the launch thread)
    size_t out_words = 0;
    std::vector<uint32_t> dep_depths;              // earlier levels this one must follow
    std::vector<void*> streams;                    // caller streams with work in this level
    size_t in_base = 0, out_base = 0;              // offsets in the group's staging blocks
    size_t gate_count() const { return gates[0].size() + gates[1].size() + gates[2].size(); }
    void reset()
    {
        gates[0].clear(); gates[1].clear(); gates[2].clear(); uploads.clear(); downloads.clear(); upload_ctxts.clear(); deliveries.clear();
        deps[0].clear(); deps[1].clear(); deps[2].clear();
        level_ordered = false;
        dep_depths.clear(); streams.clear();
        in_words = out_words = in_base = out_base = 0;
    }
};

struct EventHolder {
    Backend* be;
    void* ev = nullptr;
    explicit EventHolder(Backend* b) : be(b) {}
    ~EventHolder() { if (ev) be->event_destroy(ev); }
};

struct Group {                        // consecutive levels flushed together
    uint64_t id = 0;
    uint32_t first_depth = 0, last_depth = 0;
    std::vector<Plan*> plans;
    int stream = 0;
    std::vector<std::shared_ptr<EventHolder>> deps;
    std::shared_ptr<EventHolder> done;
    size_t in_words = 0, out_words = 0;
    void *pin_in = nullptr, *pin_out = nullptr;
    uint32_t *dev_in = nullptr, *dev_out = nullptr;
    size_t pin_in_cap = 0, pin_out_cap = 0, dev_in_cap = 0, dev_out_cap = 0;
    std::atomic<int> state{0};        // 0 queued, 1 launched, 2 retired
    int error = 0;
    std::string error_text;
    GroupTrace trace;
    void* marks[4] = {nullptr, nullptr, nullptr, nullptr};
    bool zero_copy_in = false, zero_copy_out = false;     // dev_in / dev_out alias the pinned blocks: nothing to recycle
    std::vector<void*> ext_waits;     // caller streams whose own work (enqueued through the raw handle) this group must follow
    std::vector<void*> lane_events;   // events between the two lanes of a per-gate scheduled flush (destroyed when the group retires)
};

// A few helper threads for the two host copies that sit on the latency path of a flush: ciphertexts out of the tlwehosts into the
// pinned staging block (launch worker), results out of the pinned block into the tlwehosts (issuing thread).  run(n, f) calls f(0) ..
// f(n-1), the caller taking its share, and returns when all are done; one run at a time.
class CopyHelpers {
   public:
    explicit CopyHelpers(int threads)
    {
        for (int i = 0; i < threads; i++) th_.emplace_back([this] { loop(); });
    }
    ~CopyHelpers()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    template <class F>
    void run(size_t n, F f)
    {
        if (n == 0) return;
        std::lock_guard<std::mutex> one(run_mu_);
        std::function<void(size_t)> fn = f;
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            next_ = 0;
            total_ = n;
            done_ = 0;
            epoch_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return done_ == total_; });
        fn_ = nullptr;
    }

   private:
    void work()
    {
        for (;;) {
            size_t i;
            const std::function<void(size_t)>* fn;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (!fn_ || next_ >= total_) return;
                i = next_++;
                fn = fn_;
            }
            (*fn)(i);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (++done_ == total_) cv_done_.notify_all();
            }
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
            }
            work();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, cv_done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t next_ = 0, total_ = 0, done_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

class Scheduler;

class DeviceSched {
   public:
    DeviceSched(Scheduler* owner, int device, Backend* be, bool threaded)
        : owner_(owner), device_(device), be_(be), threaded_(threaded)
    {
        nstreams_ = std::max(1, be_->num_streams());
        set_round_gates(be_->round_gates());
        if (threaded_) worker_ = std::thread([this] { worker_loop(); });
    }
    ~DeviceSched()
    {
        stop_worker();
        for (Plan* p : plan_pool_) delete p;
    }

    void stop_worker()
    {
        if (!worker_.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        worker_.join();
    }

    Backend* backend() { return be_; }
    Stats& stats() { return stats_; }
    // the most recent retired groups, oldest first (at most 64 are kept)
    size_t get_trace(GroupTrace* out, size_t max, bool clear)
    {
        size_t n = std::min(max, trace_.size());
        for (size_t i = 0; i < n; i++) out[i] = trace_[trace_.size() - n + i];
        if (clear) trace_.clear();
        return n;
    }
    const std::string& error_text() const { return err_; }
    size_t pending_levels() const { return levels_.size(); }

    // kind: which launch group the gate joins (the output's level for ordinary gates, 2 for TRLWE-level operations)
    int record_gate(void* stream, int op, bool copying, cufhe_amd_ctxt* out, cufhe_amd_ctxt* const (&ins)[3], int kind = -1);
    int record_copy(void* stream, cufhe_amd_ctxt* c, bool to_device);
    int flush(size_t max_levels = (size_t)-1);
    int flush_for_completion()                 // the caller is about to wait for everything: renamed values go home in the same flush
    {
        restore_homes(nullptr, true);
        return flush();
    }
    int stream_query(void* stream);            // 1: everything issued on `stream` is complete and delivered
    int synchronize();
    // Stream::st() of the reference hands the caller the stream its gates were ENQUEUED on (include/cufhe_gpu.cuh:183,
    // src/cufhe_gates_gpu.cu:148-167).  stream_fence gives the raw handle the same meaning at the moment it is handed out: everything
    // recorded on `stream` is launched (values written on it return to their ciphertexts' own buffers first), the caller's stream is made
    // to wait for those launches, and from now on work recorded on `stream` waits for what the caller has put on the raw stream.
    int stream_fence(void* stream);
    // cudaStreamSynchronize(st.st()) of the reference: returns when everything issued on `stream` is complete AND delivered
    int stream_synchronize(void* stream);
    bool is_external(void* stream) const { return ext_streams_.count(stream) != 0; }
    void forget_stream(void* stream)
    {
        streams_.erase(stream);
        cached_stream_ = nullptr;
        cached_ss_ = nullptr;
    }
    // the caller destroys its stream: no queued launch may still name the raw handle
    int retire_external_stream(void* stream)
    {
        if (!ext_streams_.count(stream)) return 0;
        int rc = 0;
        auto it = streams_.find(stream);
        if (it != streams_.end() && it->second.max_depth >= base_depth_) rc = flush();
        wait_worker_idle();
        ext_streams_.erase(stream);
        fence_epoch_.erase(stream);
        return rc;
    }
    // returns once no launch thread of this device is in the middle of reading tlwehost memory
    void copy_fence() { std::lock_guard<std::mutex> lk(copy_mu_); }
    bool level_done(uint32_t depth) { return done(depth); }
    // has the level at `depth` been launched, i.e. have its uploads been copied out of tlwehost?
    bool uploads_copied(uint32_t depth)
    {
        if (depth >= base_depth_) return false;
        Group* g = find_group(depth);
        return !g || g->state.load(std::memory_order_acquire) >= 1;
    }
    // launch everything recorded and wait until the launch thread has copied its inputs
    int flush_and_copy()
    {
        if (int rc = flush()) return rc;
        wait_worker_idle();
        return 0;
    }
    // release every cached buffer; the device must be idle (synchronize() first)
    void release_buffers();
    // ... and the ciphertext slabs: only once no ciphertext of this scheduler is alive
    void release_slabs()
    {
        for (void* s : slabs_) be_->free_device(s);
        slabs_.clear();
        for (auto& f : free_slots_) f.clear();
        retired_.clear();
    }

    // one slot per ciphertext, carved from slabs (the reference pays a cudaMalloc per Ctxt per GPU,
    // include/cufhe_gpu.cuh:76-95)
    int slot_alloc(int level, uint32_t** out);
    void slot_free(int level, uint32_t* p) { free_slots_[level].push_back(p); }

    // A front level this full is launched at once: two rounds of the blind-rotate grid while the device still has work (102.5 k
    // gates/s per launch against 100.3 k for one round; 32 768 gates through the per-gate API 96.1 k -> 99.4 k gates/s), ONE round when
    // the device is idle -- nothing to overlap the recording of the second round with (4096 gates: 43.7 ms against 44.5 - 50).
    // Both in grid rounds of the device (Backend::round_gates; MI355X: 8 rotations on each of its 256 CUs), set by set_round_gates.
    size_t level_flush_gates = 0;
    size_t idle_flush_gates = 0;
    size_t total_flush_gates = 32768;  // bound on the recorded program
    void set_round_gates(size_t round)
    {
        round_gates_ = std::max<size_t>(8, round);
        level_flush_gates = 2 * round_gates_;
        idle_flush_gates = round_gates_;
    }
    void set_level_flush_gates(size_t gates)       // an explicit flush size ("sched_level_gates"); the idle rule never exceeds it
    {
        level_flush_gates = std::max<size_t>(1, gates);
        idle_flush_gates = std::min(level_flush_gates, round_gates_);
    }
    size_t round_gates() const { return round_gates_; }
    // Renaming: an output whose device buffer still has recorded users (an earlier write not yet superseded, readers of
    // the old value) takes a FRESH buffer instead of waiting for them, so that only true data dependences order the
    // program (a temporary re-used down a ripple-carry chain no longer serialises the adders' independent gates).  The
    // old buffer is recycled once every level up to the last one naming it has retired.  The buffer a ciphertext was
    // created with (Ctxt::tlwedevices[i], include/cufhe_gpu.cuh:80-84) stays its HOME: it is never recycled while the
    // ciphertext lives, a later write returns to it when nothing recorded names it any more, and a value still living in a
    // renamed buffer when the caller asks for completion (Synchronize, StreamQuery of the stream that wrote it) is copied
    // home first -- one Copy gate per such ciphertext in the flush that the request triggers -- so that the published
    // pointer holds the value whenever the host is entitled to look, exactly as without renaming.  TLWE ciphertexts only
    // (the copy back is an ordinary Copy gate of the ciphertext's level).
    bool rename_outputs = true;
    // Per-gate scheduling of flushes with several dependence levels on two lanes (compile_two_lane below; "sched_two_lane").  Needs
    // renaming (the recorded program must be single-assignment) and a backend with a lane model.
    int two_lane = 1;                  // 0: never; 1: when the backend's cost model says it beats the level order; 2: whenever the flush is eligible (tests)
    int copy_threads = 4;              // host threads that share a large gather / delivery ("sched_copy_threads"; before the first flush)
    size_t parallel_copy_min = 1024;   // ... from this many ciphertexts of one level on
    int copy_op = 13;                  // the op code of Copy (CUFHE_AMD_COPY) in GateRef::op
    void forget_renamed(cufhe_amd_ctxt* c)        // the ciphertext goes away: nothing to copy home any more
    {
        cufhe_amd_ctxt::PerDev& pd = c->d[device_];
        if (pd.renamed_idx < 0) return;
        cufhe_amd_ctxt* last = renamed_.back();
        renamed_[(size_t)pd.renamed_idx] = last;
        last->d[device_].renamed_idx = pd.renamed_idx;
        renamed_.pop_back();
        pd.renamed_idx = -1;
    }
    // every level up to `depth` has completed and been retired (groups retire out of order and run on different internal
    // streams: the newest level naming a buffer being done does not mean that an older reader in another group is)
    bool all_done_through(uint32_t depth)
    {
        if (depth == 0) return true;
        if (depth >= base_depth_) return false;
        return live_.empty() || live_.front()->first_depth > depth;
    }

   private:
    struct StreamState { uint32_t max_depth = 0; std::vector<uint64_t> open; };
    struct Buf { void* p; size_t cap; };
    struct RetiredBuf { uint32_t* p; int level; uint32_t last_use; };      // renamed-away buffers still named by recorded levels
    std::vector<RetiredBuf> retired_;
    std::vector<cufhe_amd_ctxt*> renamed_;        // live ciphertexts whose value is in a renamed buffer on this device
    uint32_t home_copy_depth_ = 0;                // the lowest level on record that holds a copy home (0: none)
    int restore_homes(void* only_stream, bool all = false);     // record the copies back: all, or those last written / uploaded on one caller stream (nullptr is a stream like any other: the default one)
    static uint32_t max_depth_of(const std::vector<uint32_t>& v)
    {
        uint32_t m = 0;
        for (uint32_t d : v) m = std::max(m, d);
        return m;
    }
    void collect_retired()
    {
        size_t k = 0;
        for (const RetiredBuf& r : retired_) {
            if (all_done_through(r.last_use)) slot_free(r.level, r.p);
            else retired_[k++] = r;
        }
        retired_.resize(k);
    }

    int fail(int rc, const std::string& what)
    {
        err_ = what;
        return rc;
    }
    Plan& plan_at(uint32_t depth)
    {
        while (base_depth_ + levels_.size() <= depth) {
            Plan* p;
            if (plan_pool_.empty()) p = new Plan();
            else {
                p = plan_pool_.back();
                plan_pool_.pop_back();
            }
            p->depth = base_depth_ + (uint32_t)levels_.size();
            levels_.push_back(p);
        }
        return *levels_[depth - base_depth_];
    }
    Group* find_group(uint32_t depth)
    {
        // live_ is sorted by depth; groups retire out of order, so a retired one may sit in the middle
        size_t lo = 0, hi = live_.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (live_[mid]->last_depth < depth) lo = mid + 1;
            else hi = mid;
        }
        if (lo < live_.size() && live_[lo]->first_depth <= depth) return live_[lo];
        return nullptr;
    }
    // has the level at `depth` completed and been retired?
    bool done(uint32_t depth)
    {
        if (depth == 0) return true;
        if (depth >= base_depth_) return false;
        Group* g = find_group(depth);
        return !g || g->state.load(std::memory_order_acquire) == 2;
    }
    static uint32_t max_reader(const cufhe_amd_ctxt::PerDev& pd)
    {
        uint32_t m = 0;
        for (uint32_t i = 0; i < pd.nrd && i < 4; i++) m = std::max(m, pd.rd[i]);
        for (uint32_t r : pd.rd_more) m = std::max(m, r);
        return m;
    }
    static bool has_readers(const cufhe_amd_ctxt::PerDev& pd) { return pd.nrd != 0; }
    static void clear_readers(cufhe_amd_ctxt::PerDev& pd)
    {
        pd.nrd = 0;
        if (!pd.rd_more.empty()) pd.rd_more.clear();
    }
    template <class F>
    static void for_readers(const cufhe_amd_ctxt::PerDev& pd, F f)
    {
        for (uint32_t i = 0; i < pd.nrd && i < 4; i++) f(pd.rd[i]);
        for (uint32_t r : pd.rd_more) f(r);
    }
    void add_dep(Plan& p, uint32_t depth)
    {
        if (depth == 0 || depth >= p.depth || done(depth)) return;
        if (!p.dep_depths.empty() && p.dep_depths.back() == depth) return;
        p.dep_depths.push_back(depth);
    }
    void add_reader(cufhe_amd_ctxt::PerDev& pd, uint32_t depth)
    {
        bool seen = false;
        for_readers(pd, [&](uint32_t r) { seen = seen || r == depth; });
        if (seen) return;
        if (pd.nrd >= 4 && (pd.nrd & 7) == 0) {     // now and then forget levels that have retired
            uint32_t keep[4], k = 0;
            std::vector<uint32_t> more;
            for_readers(pd, [&](uint32_t r) {
                if (done(r)) return;
                if (k < 4) keep[k++] = r;
                else more.push_back(r);
            });
            for (uint32_t i = 0; i < k; i++) pd.rd[i] = keep[i];
            pd.nrd = k + (uint32_t)more.size();
            pd.rd_more.swap(more);
        }
        if (pd.nrd < 4) pd.rd[pd.nrd] = depth;
        else pd.rd_more.push_back(depth);
        pd.nrd++;
    }
    static void use(cufhe_amd_ctxt::PerDev& pd, uint32_t depth) { pd.last_use = std::max(pd.last_use, depth); }
    StreamState& stream_state(void* stream)
    {
        if (stream != cached_stream_ || !cached_ss_) {
            cached_ss_ = &streams_[stream];      // references to map elements survive rehashing
            cached_stream_ = stream;
        }
        return *cached_ss_;
    }
    void note_stream(Plan& p, void* stream, uint32_t depth)
    {
        StreamState& ss = stream_state(stream);
        ss.max_depth = std::max(ss.max_depth, depth);
        if (p.streams.empty() || p.streams.back() != stream) p.streams.push_back(stream);
    }
    // nothing queued, nothing running: every flushed group has completed (whether or not its results were delivered yet)
    bool device_idle()
    {
        for (Group* g : live_) {
            const int st = g->state.load(std::memory_order_acquire);
            if (st == 0) return false;
            if (st == 1 && !g->error && g->done->ev && be_->event_query(g->done->ev) != 1) return false;
        }
        return true;
    }
    int resolve_host(cufhe_amd_ctxt* c, bool* need_upload, void* stream);
    void record_upload(cufhe_amd_ctxt* c, void* stream);
    int after_record();
    bool two_lane_available()       // could a flush of several levels be compiled into a per-gate plan? (the gates of compile_two_lane)
    {
        Backend::LaneModel m;
        return two_lane && rename_outputs && nstreams_ >= 2 && be_->lane_model(&m) && m.chain_gates != 0 && m.bulk_gates != 0;
    }
    int launch(Group* g);                       // worker (or inline): submit the group's work
    // one launch of a per-gate scheduled flush: lane 0 = chain, 1 = bulk; wait_other = the newest launch of the OTHER lane that
    // produces one of its operands (-1: none; launches of one lane run in order on one stream)
    struct LaneLaunch { int lane = 0; int index = 0; int wait_other = -1; bool signal = false; std::vector<GateRef> gates; };
    // tail: the independent remainder, one launch behind both lanes; post: the copies home, behind everything
    struct TwoLanePlan { int kind = 0; std::vector<LaneLaunch> seq; std::vector<GateRef> tail, post; double est_ms = 0, level_ms = 0; };
    bool compile_two_lane(Group* g, TwoLanePlan* out);
    int retire(Group* g);                       // issuing thread: deliver results, recycle
    void worker_loop();
    void wait_worker_idle()
    {
        if (!threaded_) return;
        std::unique_lock<std::mutex> lk(mu_);
        cv_idle_.wait(lk, [&] { return queue_.empty() && !busy_; });
    }
    int get_buf(std::vector<Buf>& cache, size_t bytes, bool pinned, void** p, size_t* cap);

    Scheduler* owner_;
    int device_;
    Backend* be_;
    bool threaded_;
    int nstreams_ = 1;
    size_t round_gates_ = 2048;
    int rr_ = 0;
    uint64_t worker_cpus_ = 0;
    std::deque<GroupTrace> trace_;
    std::string err_;
    Stats stats_;

    // the recorded, not yet launched program: levels_[i] has depth base_depth_ + i
    uint32_t base_depth_ = 1;
    std::deque<Plan*> levels_;
    size_t pending_gates_ = 0;
    std::unordered_map<void*, StreamState> streams_;
    std::unordered_set<void*> ext_streams_;     // caller streams whose raw handle has been handed out (stream_fence)
    std::unordered_map<void*, uint32_t> fence_epoch_;      // ... and how often
    uint32_t fence_epoch(void* stream) const
    {
        if (fence_epoch_.empty()) return 0;
        auto it = fence_epoch_.find(stream);
        return it == fence_epoch_.end() ? 0 : it->second;
    }
    void* cached_stream_ = nullptr;
    StreamState* cached_ss_ = nullptr;
    std::vector<Plan*> plan_pool_;              // retired levels, vectors keep their capacity
    uint64_t next_group_ = 1;
    std::deque<Group*> live_;                   // launched or queued groups, oldest first
    int sticky_error_ = 0;

    std::vector<uint32_t*> free_slots_[kLevels];
    std::vector<void*> slabs_;

    std::mutex copy_mu_;                        // held while the launch thread copies out of tlwehost memory
    // shared with the worker
    std::mutex mu_;
    std::condition_variable cv_, cv_idle_;
    std::deque<Group*> queue_;
    bool busy_ = false, stop_ = false;
    std::vector<Buf> pinned_cache_, dev_cache_;
    std::thread worker_;
    // the copies in and out of the tlwehosts are split over copy_threads (the calling thread included) from parallel_copy_min ciphertexts on
    std::unique_ptr<CopyHelpers> helpers_;
    std::once_flag helpers_once_;
    CopyHelpers* helpers()
    {
        std::call_once(helpers_once_, [&] { if (copy_threads > 1) helpers_.reset(new CopyHelpers(copy_threads - 1)); });
        return helpers_.get();
    }
};

class Scheduler {
   public:
    // `make_backend(device)` supplies the device layer; the scheduler owns the returned objects
    template <class Factory>
    Scheduler(int gpu_num, bool threaded, Factory make_backend)
    {
        for (int d = 0; d < gpu_num; d++) {
            backends_.emplace_back(make_backend(d));
            devs_.emplace_back(new DeviceSched(this, d, backends_.back().get(), threaded));
        }
    }
    // Tears the device objects down through the backend: only for a scheduler whose devices are
    // idle and whose ciphertexts are gone (the HIP library keeps its scheduler for the process
    // lifetime instead, see capi.hip).
    ~Scheduler()
    {
        for (auto& d : devs_) d->stop_worker();
        for (auto& d : devs_) {
            d->release_buffers();
            d->release_slabs();
        }
    }
    int live_ctxts() const { return live_ctxts_; }
    int gpu_num() const { return (int)devs_.size(); }
    DeviceSched& dev(int d) { return *devs_[d]; }
    uint64_t new_token() { return ++token_; }

    int ctxt_create(int level, uint32_t* host_words, cufhe_amd_ctxt** out, std::string* err)
    {
        cufhe_amd_ctxt* c = new cufhe_amd_ctxt();
        c->level = level;
        c->words = devs_[0]->backend()->words(level);
        c->host = host_words;
        c->d.resize(devs_.size());
        for (size_t d = 0; d < devs_.size(); d++)
            if (int rc = devs_[d]->slot_alloc(level, &c->d[d].dev)) {
                *err = devs_[d]->error_text();
                for (size_t e = 0; e < d; e++) devs_[e]->slot_free(level, c->d[e].dev);
                delete c;
                return rc;
            }
        for (size_t d = 0; d < devs_.size(); d++) c->d[d].home = c->d[d].dev;
        live_ctxts_++;
        c->owner = this;
        *out = c;
        return 0;
    }
    // The caller's ciphertext goes away; its device buffers are recycled once the last recorded or
    // in-flight gate naming them has retired.  No flush, no wait: RAII temporaries in a circuit
    // stay cheap.
    void ctxt_destroy(cufhe_amd_ctxt* c)
    {
        if (c->host && c->host_reads.load(std::memory_order_acquire) > 0) {
            // recorded uploads have not read tlwehost yet: save the words for them, then make sure no launch
            // thread is still reading the caller's memory
            const size_t words = (size_t)devs_[0]->backend()->words(c->level);
            uint32_t* copy = new uint32_t[words];
            memcpy(copy, c->host, words * 4);
            c->shadow.store(copy, std::memory_order_release);
            for (auto& d : devs_) d->copy_fence();
        }
        c->destroyed = true;
        c->host = nullptr;
        for (size_t d = 0; d < devs_.size(); d++) devs_[d]->forget_renamed(c);
        if (!collect(c)) zombies_.push_back(c);
    }
    // called when levels retire: release destroyed ciphertexts that nothing recorded names any more
    void collect_zombies()
    {
        if (zombies_.empty()) return;
        size_t k = 0;
        for (cufhe_amd_ctxt* c : zombies_)
            if (!collect(c)) zombies_[k++] = c;
        zombies_.resize(k);
    }
    // A recorded result is about to be routed to c's tlwehost (a copying gate's output, CtxtCopyD2H) from
    // `device`.  On that device launch order keeps earlier uploads of c ahead of the delivery; on the OTHER
    // devices nothing does: their recorded uploads must have read the memory first, and what they hold can
    // no longer be taken for "the current tlwehost".
    int before_host_write(cufhe_amd_ctxt* c, int device)
    {
        for (size_t e = 0; e < devs_.size(); e++) {
            if ((int)e == device) continue;
            cufhe_amd_ctxt::PerDev& pe = c->d[e];
            pe.snap_plan = nullptr;
            pe.snap_owned = false;
            if (pe.last_upload && !devs_[e]->uploads_copied(pe.last_upload)) {
                devs_[device]->stats().forced_syncs++;
                if (int rc = devs_[e]->flush_and_copy()) return rc;
            }
        }
        return 0;
    }
    // The CALLER is about to overwrite c's tlwehost itself, now (TRGSW2NTT fills trgswhost synchronously,
    // src/bootstrap_gpu.cu:75-94) -- not through a recorded delivery.  A result still on its way to that memory must land
    // first, recorded uploads that have not read it yet must read the old words first, and no device may go on taking
    // what it holds for "the current tlwehost".
    int before_direct_host_write(cufhe_amd_ctxt* c)
    {
        if (c->host_dev >= 0)
            if (int rc = devs_[c->host_dev]->synchronize()) return rc;
        for (size_t e = 0; e < devs_.size(); e++) {
            cufhe_amd_ctxt::PerDev& pe = c->d[e];
            pe.snap_plan = nullptr;
            pe.snap_owned = false;
            if (pe.last_upload && !devs_[e]->uploads_copied(pe.last_upload))
                if (int rc = devs_[e]->flush_and_copy()) return rc;
        }
        return 0;
    }
    int synchronize_all()
    {
        // hand every device its recorded work first, then wait: the devices run concurrently
        int rc = 0;
        for (auto& d : devs_)
            if (int r = d->flush_for_completion()) rc = rc ? rc : r;
        for (auto& d : devs_)
            if (int r = d->synchronize()) rc = rc ? rc : r;
        return rc;
    }

   private:
    bool collect(cufhe_amd_ctxt* c);            // release c if no recorded or in-flight level names it
    void ctxt_release(cufhe_amd_ctxt* c)
    {
        for (size_t d = 0; d < c->d.size() && d < devs_.size(); d++) {
            if (c->d[d].dev) devs_[d]->slot_free(c->level, c->d[d].dev);
            if (c->d[d].home && c->d[d].home != c->d[d].dev) devs_[d]->slot_free(c->level, c->d[d].home);
        }
        delete[] c->shadow.load();
        delete c;
        live_ctxts_--;
    }
    std::vector<std::unique_ptr<Backend>> backends_;
    std::vector<std::unique_ptr<DeviceSched>> devs_;
    uint64_t token_ = 0;
    int live_ctxts_ = 0;
    std::vector<cufhe_amd_ctxt*> zombies_;
};

inline bool Scheduler::collect(cufhe_amd_ctxt* c)
{
    for (size_t d = 0; d < devs_.size(); d++) {
        if (!devs_[d]->all_done_through(c->d[d].last_use)) return false;
        for (uint32_t dd : c->d[d].home_deps)
            if (!devs_[d]->all_done_through(dd)) return false;
    }
    ctxt_release(c);
    return true;
}

// ---------------------------------------------------------------------------------------------

inline int DeviceSched::slot_alloc(int level, uint32_t** out)
{
    std::vector<uint32_t*>& fl = free_slots_[level];
    if (fl.empty()) {
        const size_t slot_bytes = ((size_t)be_->slot_words(level) * 4 + 255) & ~(size_t)255;
        const size_t count = std::max<size_t>(16, std::min<size_t>(512, ((size_t)4 << 20) / slot_bytes));    // slabs of <= 4 MiB
        void* slab = nullptr;
        be_->bind_thread();
        if (int rc = be_->alloc_device(slot_bytes * count, &slab)) return fail(rc, be_->error_text());
        slabs_.push_back(slab);
        for (size_t i = count; i-- > 0;) fl.push_back((uint32_t*)((char*)slab + i * slot_bytes));
    }
    *out = fl.back();
    fl.pop_back();
    return 0;
}

inline int DeviceSched::get_buf(std::vector<Buf>& cache, size_t bytes, bool pinned, void** p, size_t* cap)
{
    {
        std::lock_guard<std::mutex> lk(mu_);
        size_t best = cache.size();
        for (size_t i = 0; i < cache.size(); i++)
            if (cache[i].cap >= bytes && (best == cache.size() || cache[i].cap < cache[best].cap)) best = i;
        if (best != cache.size()) {
            *p = cache[best].p;
            *cap = cache[best].cap;
            cache.erase(cache.begin() + best);
            return 0;
        }
    }
    *cap = bytes + bytes / 2 + 4096;
    return pinned ? be_->alloc_pinned(*cap, p) : be_->alloc_device(*cap, p);
}

inline void DeviceSched::release_buffers()
{
    std::lock_guard<std::mutex> lk(mu_);
    for (Buf& b : pinned_cache_) be_->free_pinned(b.p);
    for (Buf& b : dev_cache_) be_->free_device(b.p);
    pinned_cache_.clear();
    dev_cache_.clear();
}

// Where does the value a non-g gate must read for `c` live?  The reference copies `tlwehost` to the
// device in stream order.  If an earlier result is still travelling to that `tlwehost`, the value
// is that result: it is the device buffer itself when nothing overwrote the buffer since (the
// normal chain Nand(c, ..) ; Or(d, c, ..)); otherwise the host copy has to land first.
inline int DeviceSched::resolve_host(cufhe_amd_ctxt* c, bool* need_upload, void* stream)
{
    cufhe_amd_ctxt::PerDev& pd = c->d[device_];
    if (c->host_dev >= 0) {
        // (the device buffer stands for the travelling result only while the caller cannot have written it itself through the raw
        // handle of the stream that produced it: the reference's upload would bring the RESULT back, src/cufhe_gates_gpu.cu:148-158)
        if (c->host_dev == device_ && c->host_version == pd.version && fence_epoch(c->host_stream) == c->host_epoch) {
            pd.note_upload_stream(stream);       // the reference would upload here: completion on this stream finds the value in the home buffer
            *need_upload = false;
            return 0;
        }
        stats_.forced_syncs++;
        const int other = c->host_dev;
        if (int rc = owner_->dev(other).synchronize()) return fail(rc, owner_->dev(other).error_text());
    }
    // `tlwehost` is plain memory now.  An input that the device buffer still holds from an earlier upload
    // is not copied again: shared inputs (test/test_intensive.cc) stay pure reads.  While that upload has
    // not retired the caller cannot have changed the memory (it has not observed completion of the gates
    // reading it); afterwards the words are compared with a copy kept for inputs that were re-used.
    // (not while the caller holds a raw stream handle: it may have written the device buffer itself, and the reference's non-g gate
    // uploads tlwehost whatever the buffer holds)
    if (pd.snap_version == pd.version && ext_streams_.empty()) {
        const bool same = pd.snap_plan != nullptr ||
                          (pd.snap_owned && c->host && !memcmp(pd.snap_own.data(), c->host, (size_t)be_->words(c->level) * 4));
        if (same) {
            pd.snap_hits++;
            stats_.uploads_shared++;
            pd.note_upload_stream(stream);       // as far as this stream's completion goes, the value was uploaded here
            *need_upload = false;
            return 0;
        }
    }
    *need_upload = true;
    return 0;
}

// tlwehost (as of now) -> device buffer, at the earliest level that follows every recorded access
inline void DeviceSched::record_upload(cufhe_amd_ctxt* c, void* stream)
{
    cufhe_amd_ctxt::PerDev& pd = c->d[device_];
    uint32_t U = std::max(base_depth_, pd.ready);
    if (pd.w_upload) U = std::max(U, pd.wdepth + 1);
    if (has_readers(pd)) U = std::max(U, max_reader(pd) + 1);
    Plan& p = plan_at(U);
    add_dep(p, pd.wdepth);
    for_readers(pd, [&](uint32_t r) { add_dep(p, r); });
    const size_t slot = p.in_words;
    p.in_words += (size_t)be_->words(c->level);
    p.uploads.push_back({pd.dev, slot, c->level});
    p.upload_ctxts.push_back(c);
    c->host_reads.fetch_add(1, std::memory_order_relaxed);
    use(pd, U);
    pd.version++;
    pd.wdepth = U;
    pd.w_upload = true;
    pd.ready = U;
    clear_readers(pd);
    pd.snap_plan = &p;
    pd.snap_off = slot;
    pd.snap_version = pd.version;
    pd.snap_hits = 0;
    pd.snap_owned = false;
    pd.last_upload = U;
    pd.note_upload_stream(stream);   // the newest write of the value: completion observed on THIS stream, too, must find it in the home buffer
    note_stream(p, stream, U);       // completion of the stream implies this level has retired: tlwehost may be edited again
    stats_.uploads++;
}

inline int DeviceSched::record_gate(void* stream, int op, bool copying, cufhe_amd_ctxt* out,
                                    cufhe_amd_ctxt* const (&ins)[3], int kind)
{
    if (kind < 0) kind = out->level;
    ScopedNs timer(&stats_.record_ns);
    // inputs that must be refreshed from the host (may wait for another device: do it first)
    bool need_up[3] = {false, false, false};
    if (copying)
        for (int i = 0; i < 3; i++) {
            if (!ins[i]) continue;
            bool seen = false;
            for (int j = 0; j < i; j++) seen = seen || ins[j] == ins[i];
            if (seen) continue;
            if (!ins[i]->host) return fail(-1, "gate on a destroyed ciphertext");
            if (int rc = resolve_host(ins[i], &need_up[i], stream)) return rc;
        }
    if (copying)
        if (int rc = owner_->before_host_write(out, device_)) return fail(rc, "scheduler: flushing another device failed");
    for (int i = 0; i < 3; i++)
        if (need_up[i]) record_upload(ins[i], stream);

    uint32_t Din = base_depth_;       // the true data dependences: every input has been produced
    for (int i = 0; i < 3; i++)
        if (ins[i]) Din = std::max(Din, ins[i]->d[device_].ready);
    cufhe_amd_ctxt::PerDev& po = out->d[device_];
    uint32_t D = std::max(Din, po.ready);                             // write after write
    if (has_readers(po)) D = std::max(D, max_reader(po) + 1);         // write after read
    // operand buffers as of now: an in-place gate reads the buffer its output may be about to leave
    const uint32_t* const in_dev[3] = {ins[0]->d[device_].dev, ins[1] ? ins[1]->d[device_].dev : nullptr,
                                       ins[2] ? ins[2]->d[device_].dev : nullptr};
    // Per-gate scheduling of a flush (two_lane) orders gates by their data dependences ONLY: every other use of the output buffer that is
    // still on record -- a reader of the value it holds, a write this gate does not read, a result of that value still travelling to
    // tlwehost which this gate does not supersede -- must then be taken out of the way by renaming, whether or not the level order
    // would have kept them apart.
    bool users = false;
    if (two_lane && rename_outputs && out->level <= 1) {
        bool is_input = false;
        for (int i = 0; i < 3; i++) is_input = is_input || ins[i] == out;
        for_readers(po, [&](uint32_t r) { users = users || r >= base_depth_; });
        if (!is_input && po.wdepth >= base_depth_) users = true;
        if (is_input && !copying && out->host_dev == device_ && out->host_version == po.version) users = true;
    }
    // the producers of the operands, as the buffers stand now (an in-place gate names the producer of the value it overwrites)
    GateDep gd;
    for (int i = 0; i < 3; i++) {
        if (!ins[i]) continue;
        const cufhe_amd_ctxt::PerDev& pd = ins[i]->d[device_];
        if (pd.wdepth >= base_depth_ && !pd.w_upload) { gd.depth[i] = pd.wdepth; gd.idx[i] = pd.widx; gd.kind[i] = pd.wkind; }
    }
    uint32_t* fresh = nullptr;
    bool back_home = false;
    bool hazard_by_level = false;
    if (rename_outputs && (D > Din || users) && out->level <= 1) {
        if (po.dev != po.home && all_done_through(max_depth_of(po.home_deps))) {
            fresh = po.home;                                          // nothing recorded names the home buffer any more
            back_home = true;
            D = Din;
        } else if (slot_alloc(out->level, &fresh) == 0) D = Din;
        else {
            fresh = nullptr;
            hazard_by_level = true;
        }
    } else if (D > Din || users) {
        hazard_by_level = true;
    }

    Plan& p = plan_at(D);
    if (hazard_by_level) p.level_ordered = true;
    for (int i = 0; i < 3; i++) {
        if (!ins[i]) continue;
        cufhe_amd_ctxt::PerDev& pd = ins[i]->d[device_];
        add_dep(p, pd.wdepth);
        use(pd, D);
    }
    if (!fresh) {
        add_dep(p, po.wdepth);
        for_readers(po, [&](uint32_t r) { add_dep(p, r); });
    }
    for (int i = 0; i < 3; i++)
        if (ins[i]) add_reader(ins[i]->d[device_], D);
    if (fresh) {
        // the old buffer keeps serving the levels that name it (this gate included, if it is in place)
        if (po.dev == po.home) {
            // leaving home: remember who still names it -- the write and the reads of the value it holds, and this gate
            po.home_deps.clear();
            if (po.wdepth) po.home_deps.push_back(po.wdepth);
            for_readers(po, [&](uint32_t r) { po.home_deps.push_back(r); });
            if (po.last_use) po.home_deps.push_back(po.last_use);
            for (int i = 0; i < 3; i++)
                if (ins[i] == out) po.home_deps.push_back(D);
            po.renamed_idx = (int)renamed_.size();
            renamed_.push_back(out);
        } else {
            retired_.push_back({po.dev, out->level, std::max(po.last_use, D)});
            if (back_home) {
                po.home_deps.clear();
                forget_renamed(out);
            }
        }
        po.dev = fresh;
        po.last_use = 0;
        stats_.renames++;
    }
    // the write: in-place gates are safe, every kernel reads its operands before it writes
    po.version++;
    po.wdepth = D;
    po.w_upload = false;
    po.ready = D + 1;
    clear_readers(po);
    po.snap_plan = nullptr;
    po.snap_owned = false;
    po.wstream = stream;
    po.ustream = nullptr;
    po.umany = false;
    use(po, D);
    po.widx = (uint32_t)p.gates[kind].size();
    po.wkind = (uint8_t)kind;
    p.gates[kind].push_back(GateRef{op, po.dev, in_dev[0], in_dev[1], in_dev[2]});
    p.deps[kind].push_back(gd);
    if (copying) {
        if (home_copy_depth_ && D >= home_copy_depth_) p.level_ordered = true;      // a per-gate order fetches results BEFORE the copies home run
        const size_t slot = p.out_words;
        p.out_words += (size_t)be_->words(out->level);
        p.downloads.push_back({po.dev, slot, out->level});
        const uint64_t token = owner_->new_token();
        p.deliveries.push_back({out, slot, token});
        out->host_dev = device_;
        out->host_version = po.version;
        out->host_token = token;
        out->host_stream = stream;
        out->host_epoch = fence_epoch(stream);
        stats_.downloads++;
    }
    note_stream(p, stream, D);
    pending_gates_++;
    stats_.gates++;
    return after_record();
}

inline int DeviceSched::record_copy(void* stream, cufhe_amd_ctxt* c, bool to_device)
{
    cufhe_amd_ctxt::PerDev& pd = c->d[device_];
    if (to_device) {     // CtxtCopyH2D, include/cufhe_gpu.cuh:193-199
        if (!c->host) return fail(-1, "copy of a destroyed ciphertext");
        bool need = false;
        if (int rc = resolve_host(c, &need, stream)) return rc;
        if (need) record_upload(c, stream);
        return 0;
    }
    // CtxtCopyD2H, :201-207: the gather of a level runs after its gates
    if (int rc = owner_->before_host_write(c, device_)) return fail(rc, "scheduler: flushing another device failed");
    const uint32_t D = std::max(base_depth_, pd.wdepth);
    Plan& p = plan_at(D);
    if (home_copy_depth_ && D >= home_copy_depth_) p.level_ordered = true;          // a per-gate order fetches results BEFORE the copies home run
    add_dep(p, pd.wdepth);
    add_reader(pd, D);
    const size_t slot = p.out_words;
    p.out_words += (size_t)be_->words(c->level);
    p.downloads.push_back({pd.dev, slot, c->level});
    const uint64_t token = owner_->new_token();
    p.deliveries.push_back({c, slot, token});
    use(pd, D);
    c->host_dev = device_;
    c->host_version = pd.version;
    c->host_token = token;
    c->host_stream = stream;
    c->host_epoch = fence_epoch(stream);
    stats_.downloads++;
    note_stream(p, stream, D);
    return 0;
}

// The caller is about to observe completion (Synchronize, or StreamQuery of `only_stream`): every value that lives in a
// renamed buffer goes back to the buffer its ciphertext was created with, as one Copy gate per ciphertext at the first level
// that follows the value's producer and the last recorded users of the home buffer.
inline int DeviceSched::restore_homes(void* only_stream, bool all)
{
    for (size_t k = 0; k < renamed_.size();) {
        cufhe_amd_ctxt* c = renamed_[k];
        cufhe_amd_ctxt::PerDev& pd = c->d[device_];
        if (!all && !pd.umany && pd.wstream != only_stream && !(pd.ustream == only_stream && only_stream != nullptr)) { k++; continue; }
        uint32_t D = std::max(base_depth_, pd.ready);
        D = std::max(D, max_depth_of(pd.home_deps) + 1);
        Plan& p = plan_at(D);
        add_dep(p, pd.wdepth);
        for (uint32_t dd : pd.home_deps) add_dep(p, dd);
        GateDep gd;
        gd.home_copy = true;
        if (pd.wdepth >= base_depth_ && !pd.w_upload) { gd.depth[0] = pd.wdepth; gd.idx[0] = pd.widx; gd.kind[0] = pd.wkind; }
        home_copy_depth_ = home_copy_depth_ ? std::min(home_copy_depth_, D) : D;
        const uint32_t copy_idx = (uint32_t)p.gates[c->level].size();
        p.gates[c->level].push_back(GateRef{copy_op, pd.home, pd.dev, nullptr, nullptr});
        p.deps[c->level].push_back(gd);
        retired_.push_back({pd.dev, c->level, std::max(pd.last_use, D)});
        pd.dev = pd.home;
        pd.home_deps.clear();
        pd.wdepth = D;                 // same value, same version: a result on its way to tlwehost still matches
        pd.w_upload = false;
        pd.widx = copy_idx;
        pd.wkind = (uint8_t)c->level;
        pd.ready = D + 1;
        clear_readers(pd);
        pd.last_use = D;
        note_stream(p, pd.wstream, D);
        if (pd.ustream) note_stream(p, pd.ustream, D);
        pending_gates_++;
        stats_.home_copies++;
        forget_renamed(c);             // swaps the last entry into k
    }
    return 0;
}

inline int DeviceSched::after_record()
{
    // A full front level is launched at once when the program behind it is flat or as wide as it is: the launch then
    // overlaps the recording and the copies of the next one.  When the levels behind it are narrow (the carry chains of
    // adders whose independent gates fill the front level) launching it would move the front, and the chains recorded
    // afterwards would sit one level later than their siblings: every chain level would then be a mix of all bit
    // positions and cost a started round more (measured: 310 ms against 251 for 256 sixteen-bit adders).  Such a front
    // level waits for the caller's Synchronize, up to eight rounds -- unless what follows can be scheduled gate by gate
    // (compile_two_lane: levels then do not matter): the idle device takes the first round of the front level at once and the
    // recording of the rest (0.33 us per gate: 6.8 ms for 20 480) and its plan overlap that launch.
    if (!levels_.empty()) {
        const size_t front = levels_.front()->gate_count();
        // (the idle check polls events: only once per workgroup-per-CU's worth of gates)
        if (front >= idle_flush_gates && front < level_flush_gates && (front - idle_flush_gates) % std::max<size_t>(1, round_gates_ / 8) == 0 &&
            (levels_.size() == 1 || two_lane_available()) && device_idle())
            return flush(1);
        if (front >= level_flush_gates &&
            (levels_.size() == 1 || 2 * levels_[1]->gate_count() >= level_flush_gates || front >= 8 * level_flush_gates))
            return flush(1);
    }
    if (pending_gates_ >= total_flush_gates) return flush();
    return 0;
}

inline int DeviceSched::flush(size_t max_levels)
{
    // trailing levels without any work (created by plan_at) are dropped, leading ones launched as no-ops
    while (!levels_.empty()) {
        Plan* b = levels_.back();
        if (b->gate_count() || !b->uploads.empty() || !b->downloads.empty()) break;
        b->reset();
        plan_pool_.push_back(b);
        levels_.pop_back();
    }
    if (levels_.empty()) return 0;
    const size_t k = std::min(max_levels, levels_.size());
    Group* g = new Group();
    g->id = next_group_++;
    g->first_depth = base_depth_;
    g->last_depth = base_depth_ + (uint32_t)k - 1;
    g->stream = rr_++ % nstreams_;
    size_t ngates = 0;
    for (size_t i = 0; i < k; i++) {
        Plan* p = levels_.front();
        levels_.pop_front();
        p->in_base = g->in_words;
        p->out_base = g->out_words;
        g->in_words += p->in_words;
        g->out_words += p->out_words;
        g->plans.push_back(p);
        ngates += p->gate_count();
        stats_.max_level_gates = std::max<uint64_t>(stats_.max_level_gates, p->gate_count());
        if (p->gate_count()) stats_.levels++;
        for (int l = 0; l < kKinds; l++)
            if (!p->gates[l].empty()) stats_.launch_sequences++;
    }
    // dependences on levels that were launched earlier on another internal stream
    for (Plan* p : g->plans)
        for (uint32_t dd : p->dep_depths) {
            if (dd >= g->first_depth) continue;
            Group* dg = find_group(dd);
            if (!dg || dg->state.load(std::memory_order_acquire) == 2 || dg->stream == g->stream) continue;
            bool have = false;
            for (auto& e : g->deps) have = have || e == dg->done;
            if (!have) {
                g->deps.push_back(dg->done);
                stats_.cross_stream_waits++;
            }
        }
    base_depth_ += (uint32_t)k;
    if (home_copy_depth_ && home_copy_depth_ < base_depth_) home_copy_depth_ = levels_.empty() ? 0 : base_depth_;      // (conservative for what is left on record)
    pending_gates_ -= std::min(pending_gates_, ngates);
    for (Plan* p : g->plans)
        for (void* st : p->streams) {
            StreamState& ss = streams_[st];
            if (ss.open.empty() || ss.open.back() != g->id) ss.open.push_back(g->id);
        }
    if (!ext_streams_.empty())
        for (Plan* p : g->plans)
            for (void* st : p->streams)
                if (ext_streams_.count(st) && std::find(g->ext_waits.begin(), g->ext_waits.end(), st) == g->ext_waits.end()) g->ext_waits.push_back(st);
    g->done = std::make_shared<EventHolder>(be_);
    g->trace.id = g->id;
    g->trace.levels = (uint32_t)k;
    g->trace.gates = (uint32_t)ngates;
    g->trace.stream = (uint32_t)g->stream;
    g->trace.in_bytes = g->in_words * 4;
    g->trace.out_bytes = g->out_words * 4;
    g->trace.t_queued = now_ns();
    live_.push_back(g);
    stats_.groups++;
    if (threaded_) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            queue_.push_back(g);
        }
        cv_.notify_one();
        return 0;
    }
    be_->bind_thread();
    launch(g);
    return 0;       // a launch error surfaces at the next Synchronize / StreamQuery
}

inline void DeviceSched::worker_loop()
{
    worker_cpus_ = (uint64_t)std::max(0, be_->bind_worker_thread());
    stats_.worker_cpus.store(worker_cpus_);
    for (;;) {
        Group* g;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
            if (queue_.empty()) return;
            g = queue_.front();
            queue_.pop_front();
            busy_ = true;
        }
        launch(g);
        {
            std::lock_guard<std::mutex> lk(mu_);
            busy_ = false;
        }
        cv_idle_.notify_all();
    }
}

// Per-gate scheduling of one flush.  The recorded program of the group is a DAG of gates (GateDep) over single-assignment buffers
// (record_gate renames every output whose buffer has other recorded users while two_lane is on), so any order that respects the data
// dependences computes the in-order result.  The level-by-level order pays one launch sequence per level: a ripple-carry adder's 32
// carry levels of a few hundred gates each run on the latency shapes with the rest of the chip waiting, after a first level of
// thousands.  Here the gates are list-scheduled, longest remaining path first, onto two lanes that share the device
// (Backend::LaneModel): steps of the low-latency shape on half of the compute units carry the chains while chunks of the throughput
// shape on the other half carry whatever nothing waits for.  The plan is only used when the backend's own cost model says it beats the
// level order by more than 5 %; everything it cannot express (several kinds of gates in the flush, uploads behind the first level,
// a hazard that a level kept apart because it could not be renamed) leaves the flush to the level order.
inline bool DeviceSched::compile_two_lane(Group* g, TwoLanePlan* out)
{
    const int64_t t_compile = now_ns();
    Backend::LaneModel m;
    if (!two_lane || !rename_outputs || nstreams_ < 2 || g->plans.size() < 3 || !be_->lane_model(&m) || m.chain_gates == 0 || m.bulk_gates == 0) return false;
    int kind = -1;
    size_t N = 0;
    std::vector<size_t> off(g->plans.size() + 1, 0);
    for (size_t pi = 0; pi < g->plans.size(); pi++) {
        const Plan* p = g->plans[pi];
        if (p->level_ordered || !p->gates[2].empty() || (pi > 0 && !p->uploads.empty())) return false;
        for (int l = 0; l < 2; l++)
            if (!p->gates[l].empty()) {
                if (kind >= 0 && kind != l) return false;
                kind = l;
            }
        off[pi] = N;
        N += kind >= 0 ? p->gates[kind].size() : 0;
    }
    off[g->plans.size()] = N;
    if (kind < 0 || N < 4 * m.chain_gates) return false;
    // producers (ids inside the group: a producer sits at a lower level, so at a lower id), consumers, weights, heights
    std::vector<int32_t> prod(3 * N, -1);
    std::vector<uint32_t> ncons(N, 0), indeg(N, 0), height(N, 0);
    std::vector<uint8_t> weight(N, 1), is_post(N, 0);
    std::vector<const GateRef*> gref(N, nullptr);
    double level_ms = 0;
    for (size_t pi = 0; pi < g->plans.size(); pi++) {
        const Plan* p = g->plans[pi];
        const size_t n = p->gates[kind].size();
        if (p->deps[kind].size() != n) return false;
        size_t rotations = 0;
        for (size_t gi = 0; gi < n; gi++) {
            const size_t id = off[pi] + gi;
            gref[id] = &p->gates[kind][gi];
            weight[id] = (uint8_t)std::max(0, std::min(255, be_->gate_weight(p->gates[kind][gi].op)));
            rotations += weight[id];
            const GateDep& d = p->deps[kind][gi];
            if (d.home_copy) { is_post[id] = 1; continue; }                // runs behind everything (and must not be anybody's operand: below)
            for (int i = 0; i < 3; i++) {
                if (d.depth[i] < g->first_depth) continue;                 // resident, or produced by an earlier flush (the group's own dependences)
                if (d.depth[i] >= p->depth || d.kind[i] != kind) return false;
                const size_t pp = d.depth[i] - g->first_depth;
                if (d.idx[i] >= g->plans[pp]->gates[kind].size() || is_post[off[pp] + d.idx[i]]) return false;
                bool dup = false;
                for (int j = 0; j < i; j++) dup = dup || prod[3 * id + j] == (int32_t)(off[pp] + d.idx[i]);
                if (dup) continue;
                prod[3 * id + i] = (int32_t)(off[pp] + d.idx[i]);
                ncons[off[pp] + d.idx[i]]++;
                indeg[id]++;
            }
        }
        level_ms += be_->launch_ms(rotations);
    }
    std::vector<uint32_t> cstart(N + 1, 0);
    for (size_t i = 0; i < N; i++) cstart[i + 1] = cstart[i] + ncons[i];
    std::vector<uint32_t> cons(cstart[N]), cfill(cstart.begin(), cstart.end() - 1);
    for (size_t id = 0; id < N; id++)
        for (int i = 0; i < 3; i++)
            if (prod[3 * id + i] >= 0) cons[cfill[(size_t)prod[3 * id + i]]++] = (uint32_t)id;
    for (size_t id = N; id-- > 0;)
        for (int i = 0; i < 3; i++)
            if (prod[3 * id + i] >= 0) height[(size_t)prod[3 * id + i]] = std::max(height[(size_t)prod[3 * id + i]], height[id] + 1);
    uint32_t hmax = 0;
    for (size_t id = 0; id < N; id++) hmax = std::max(hmax, height[id]);
    if (hmax < 4) return false;                                            // no chain worth a lane of its own
    auto gate_of = [&](size_t id) -> const GateRef& { return *gref[id]; };
    // list scheduling over the two lanes, simulated on the backend's cost model
    // gates whose producers are all scheduled, by the time the last of them ends: one entry per launch end (a few dozen), not per gate
    std::map<double, std::vector<uint32_t>> pending;
    // the ready gates by height, each bucket in the order its gates became ready (a flush of 20 000 gates is planned in well under a
    // millisecond: the plan sits on the critical path of the flush)
    std::vector<std::vector<uint32_t>> bucket(hmax + 1);
    std::vector<size_t> bucket_head(hmax + 1, 0);
    size_t ready_count = 0;
    auto push_ready = [&](uint32_t id) { bucket[height[id]].push_back(id); ready_count++; };
    std::vector<double> fin(N, -1.0);
    std::vector<int32_t> lane_of(N, -1), launch_of(N, -1);
    out->post.clear();
    size_t scheduled = 0, unscheduled_with_consumers = 0;
    for (size_t id = 0; id < N; id++) {
        // (a fetch of the value a copy home delivers -- recorded at its level or behind it -- marks its level level_ordered at record time:
        // results are fetched before the copies home run)
        if (is_post[id]) { out->post.push_back(gate_of(id)); lane_of[id] = 2; scheduled++; continue; }
        if (indeg[id] == 0) push_ready((uint32_t)id);
    }
    double tfree[2] = {0.0, 0.0};
    int launches[2] = {0, 0};
    std::vector<uint32_t> by_height(hmax + 1, 0);                          // unscheduled gates per height
    for (size_t id = 0; id < N; id++)
        if (!is_post[id]) by_height[height[id]]++;
    uint32_t hrem = hmax;
    const uint32_t margin = (uint32_t)std::max(1.0, std::ceil(m.bulk_ms / m.chain_ms));
    for (size_t id = 0; id < N; id++) unscheduled_with_consumers += ncons[id] != 0;
    const size_t cap[2] = {m.chain_gates, m.bulk_gates};
    const double dur[2] = {m.chain_ms, m.bulk_ms};
    out->seq.clear();
    while (scheduled < N && unscheduled_with_consumers > 0) {
        const int lane = tfree[0] <= tfree[1] ? 0 : 1;
        const double t = tfree[lane];
        while (!pending.empty() && pending.begin()->first <= t) {
            for (uint32_t id : pending.begin()->second) push_ready(id);
            pending.erase(pending.begin());
        }
        const double next = pending.empty() ? -1.0 : pending.begin()->first;
        if (ready_count == 0) {
            if (next < 0) return false;                                    // cannot happen in a DAG: leave the flush to the level order
            tfree[lane] = std::max(next, t);
            continue;
        }
        // The chain lane takes the most urgent gates first (greatest height).  The bulk lane keeps a gate for a whole chunk (about
        // `margin` chain steps): it may only take gates that the longest remaining chain does not reach for that long -- those at least
        // `margin` below the greatest height still unscheduled -- and among them the ones needed soonest.
        while (hrem > 0 && by_height[hrem] == 0) hrem--;
        int top = (int)hmax;                                               // the highest bucket this lane may take from
        if (lane == 1) {
            top = (int)hrem - (int)margin;
            size_t eligible = 0;
            for (int hh = top; hh >= 0 && eligible < cap[1] / 2; hh--) eligible += bucket[(size_t)hh].size() - bucket_head[(size_t)hh];
            if (eligible < cap[1] / 2) {
                // a chunk of the throughput shape costs its 19 ms whatever it carries: the bulk lane waits for a worthwhile load -- until
                // more gates become available or the chain lane (busy beyond t, or it would have been chosen) has taken its pick
                tfree[1] = next >= 0 ? std::min(next, tfree[0]) : tfree[0];
                continue;
            }
        }
        LaneLaunch L;
        L.lane = lane;
        L.index = launches[lane]++;
        size_t load = 0;
        std::vector<uint32_t> placed;
        bool full = false;
        for (int hh = top; hh >= 0 && !full; hh--) {
            std::vector<uint32_t>& b = bucket[(size_t)hh];
            size_t& head = bucket_head[(size_t)hh];
            while (head < b.size()) {
                const uint32_t id = b[head];
                if (load && load + weight[id] > cap[lane]) { full = true; break; }
                head++;
                ready_count--;
                by_height[height[id]]--;
                load += weight[id];
                L.gates.push_back(gate_of(id));
                placed.push_back(id);
                lane_of[id] = lane;
                launch_of[id] = L.index;
                for (int i = 0; i < 3; i++) {
                    const int32_t pr = prod[3 * id + i];
                    if (pr >= 0 && lane_of[(size_t)pr] == 1 - lane) L.wait_other = std::max(L.wait_other, launch_of[(size_t)pr]);
                }
                scheduled++;
                if (ncons[id]) unscheduled_with_consumers--;
            }
        }
        const double finish = t + dur[lane] * (lane == 0 && 2 * load <= cap[0] ? 0.7 : 1.0);      // a half-empty chain step takes the single-rotation shape
        tfree[lane] = finish;
        out->seq.push_back(std::move(L));
        // consumers whose producers are all scheduled now become available when the latest of them ends
        for (uint32_t id : placed) fin[id] = finish;
        for (uint32_t id : placed)
            for (uint32_t ci = cstart[id]; ci < cstart[id + 1]; ci++) {
                const uint32_t c = cons[ci];
                if (--indeg[c] == 0) {
                    double a = 0;
                    for (int i = 0; i < 3; i++)
                        if (prod[3 * c + i] >= 0) a = std::max(a, fin[(size_t)prod[3 * c + i]]);
                    pending[a].push_back(c);
                }
            }
    }
    // what is left has no consumers left to serve and every producer scheduled: one launch by the backend's own rules behind both lanes
    out->tail.clear();
    size_t tail_rot = 0;
    for (size_t id = 0; id < N; id++)
        if (lane_of[id] < 0) { out->tail.push_back(gate_of(id)); tail_rot += weight[id]; }
    // which launches must signal the other lane
    for (LaneLaunch& L : out->seq)
        if (L.wait_other >= 0)
            for (LaneLaunch& P : out->seq)
                if (P.lane == 1 - L.lane && P.index == L.wait_other) P.signal = true;
    out->kind = kind;
    out->level_ms = level_ms;
    out->est_ms = std::max(tfree[0], tfree[1]) + (tail_rot ? be_->launch_ms(tail_rot) : 0.0);
    static const bool debug = getenv("CUFHE_AMD_SCHED_DEBUG") != nullptr;
    if (debug)
        fprintf(stderr, "[sched] flush of %zu gates in %zu levels: level by level %.1f ms, two lanes %.1f ms (%d chain steps to %.1f, %d bulk chunks to %.1f, remainder %zu, copies home %zu; planned in %.2f ms)\n",
                N, g->plans.size(), level_ms, out->est_ms, launches[0], tfree[0], launches[1], tfree[1], tail_rot, out->post.size(), (now_ns() - t_compile) * 1e-6);
    return two_lane == 2 || out->est_ms < 0.95 * level_ms;
}

inline int DeviceSched::launch(Group* g)
{
    uint64_t ns = 0;
    struct Add { std::atomic<uint64_t>& a; uint64_t& n; ~Add() { a += n; } } add{stats_.launch_ns, ns};
    ScopedNs timer(&ns);
    int rc = 0;
    const int s = g->stream;
    auto step = [&](int r) {
        if (r && !rc) {
            rc = r;
            g->error_text = be_->error_text();
        }
        return !rc;
    };
    g->trace.t_launch_begin = now_ns();
    step(be_->event_create(&g->done->ev));
    for (auto& e : g->deps)
        if (e->ev && rc == 0) step(be_->stream_wait(s, e->ev));
    for (void* cs : g->ext_waits)
        if (rc == 0) step(be_->wait_for_caller_stream(s, cs));
    g->marks[0] = be_->mark(s);
    // Staging.  A backend whose pinned memory is visible to the device (device_alias) needs no copy engine at all: the scatter
    // kernel reads the pinned block over the bus and the gather kernel writes the results straight into it (zero_copy).  The
    // inputs of the group's FIRST level -- nothing in this group precedes them -- are then scattered chunk by chunk while the
    // worker is still gathering the next chunk out of the tlwehosts, so that the device starts its first gate one chunk after
    // the last input was copied instead of after gather + H2D + scatter one behind the other.
    bool zero_copy = false;
    size_t first_done = 0;                                    // uploads of plans[0] already scattered
    if (rc == 0 && g->in_words) {
        const size_t bytes = g->in_words * 4;
        if (step(get_buf(pinned_cache_, bytes, true, &g->pin_in, &g->pin_in_cap))) {
            uint32_t* alias = (uint32_t*)be_->device_alias(g->pin_in);
            zero_copy = alias != nullptr;
            if (zero_copy) g->dev_in = alias;
            else step(get_buf(dev_cache_, bytes, false, (void**)&g->dev_in, &g->dev_in_cap));
        }
        if (rc == 0) {
            // the reference's H2D copies, gathered: tlwehost (or the saved words of a ciphertext that was
            // destroyed meanwhile) -> the pinned staging block
            constexpr size_t kChunk = 512;
            for (size_t pi = 0; pi < g->plans.size() && !rc; pi++) {
                Plan* p = g->plans[pi];
                auto gather = [&](size_t lo, size_t hi) {
                    for (size_t i = lo; i < hi; i++) {
                        cufhe_amd_ctxt* c = p->upload_ctxts[i];
                        const uint32_t* shadow = c->shadow.load(std::memory_order_acquire);
                        memcpy((uint32_t*)g->pin_in + p->in_base + p->uploads[i].slot, shadow ? shadow : c->host,
                               (size_t)be_->words(c->level) * 4);
                        c->host_reads.fetch_sub(1, std::memory_order_release);
                    }
                };
                // a wave of chunks at a time: one chunk per copy thread, then (first level, staging visible to the device) the scatter
                // of the wave is submitted while the next wave is being gathered
                CopyHelpers* pool = p->uploads.size() >= parallel_copy_min ? helpers() : nullptr;
                const size_t wave = kChunk * (pool ? (size_t)copy_threads : 1);
                for (size_t lo = 0; lo < p->uploads.size() && !rc; lo += wave) {
                    const size_t hi = std::min(p->uploads.size(), lo + wave);
                    {
                        std::lock_guard<std::mutex> lk(copy_mu_);
                        if (pool) pool->run((hi - lo + kChunk - 1) / kChunk, [&](size_t j) { gather(lo + j * kChunk, std::min(hi, lo + (j + 1) * kChunk)); });
                        else gather(lo, hi);
                    }
                    if (zero_copy && pi == 0) {
                        step(be_->copy_ctxts(s, p->uploads.data() + lo, hi - lo, g->dev_in + p->in_base, true));
                        first_done = hi;
                    }
                }
            }
            g->trace.t_gather_end = now_ns();
            if (!zero_copy && rc == 0) step(be_->h2d(s, g->dev_in, g->pin_in, bytes));
        }
    }
    if (!g->trace.t_gather_end) g->trace.t_gather_end = now_ns();
    bool zero_copy_out = false;
    if (rc == 0 && g->out_words) {
        const size_t bytes = g->out_words * 4;
        if (step(get_buf(pinned_cache_, bytes, true, &g->pin_out, &g->pin_out_cap))) {
            uint32_t* alias = (uint32_t*)be_->device_alias(g->pin_out);
            zero_copy_out = alias != nullptr;
            if (zero_copy_out) g->dev_out = alias;
            else step(get_buf(dev_cache_, bytes, false, (void**)&g->dev_out, &g->dev_out_cap));
        }
    }
    // the device-side spans of the trace (scatter | gates | gather) are marked for groups of one level; longer groups get one span
    const bool single = g->plans.size() == 1;
    TwoLanePlan tl;
    if (rc == 0 && compile_two_lane(g, &tl)) {
        // Gate by gate on two lanes: the uploads (all in the first level), the launches in the order the plan generated them -- each on
        // its lane's stream, behind the other lane's launch that produces its operands -- the independent remainder as one launch by the
        // backend's own rules, then every result of the flush (the recorded program is single-assignment: nothing was overwritten).
        const int sc = (s + 1) % nstreams_;                            // the chain lane's stream; the bulk lane keeps the group's
        Plan* p0 = g->plans[0];
        if (p0->uploads.size() > first_done)
            step(be_->copy_ctxts(s, p0->uploads.data() + first_done, p0->uploads.size() - first_done, g->dev_in + p0->in_base, true));
        auto new_event = [&](int stream) -> void* {
            void* ev = nullptr;
            if (!step(be_->event_create(&ev))) return nullptr;
            g->lane_events.push_back(ev);
            step(be_->event_record(stream, ev));
            return ev;
        };
        if (void* start = new_event(s)) step(be_->stream_wait(sc, start));      // the chain lane starts behind the uploads and the group's dependences
        std::vector<void*> done_ev[2];
        for (const LaneLaunch& L : tl.seq) {
            if (rc) break;
            const int st = L.lane == 0 ? sc : s;
            if (L.wait_other >= 0) {
                void* ev = done_ev[1 - L.lane][(size_t)L.wait_other];
                if (ev) step(be_->stream_wait(st, ev));
            }
            step(be_->run_gates_lane(st, tl.kind, L.gates.data(), L.gates.size(), L.lane));
            done_ev[L.lane].push_back(L.signal && !rc ? new_event(st) : nullptr);
            stats_.two_lane_launches++;
        }
        if (!rc)
            if (void* joined = new_event(sc)) step(be_->stream_wait(s, joined));
        if (!rc && !tl.tail.empty()) step(be_->run_gates(s, tl.kind, tl.tail.data(), tl.tail.size()));
        // the results first: a copy home (post) rewrites a buffer whose earlier value an earlier level's result may still be waiting in
        for (Plan* p : g->plans)
            if (!rc && !p->downloads.empty())
                step(be_->copy_ctxts(s, p->downloads.data(), p->downloads.size(), g->dev_out + p->out_base, false));
        if (!rc && !tl.post.empty()) step(be_->run_gates(s, tl.kind, tl.post.data(), tl.post.size()));
        stats_.two_lane_groups++;
    } else
    for (size_t pi = 0; pi < g->plans.size(); pi++) {
        Plan* p = g->plans[pi];
        if (rc) break;
        const size_t from = pi == 0 ? first_done : 0;
        if (p->uploads.size() > from) step(be_->copy_ctxts(s, p->uploads.data() + from, p->uploads.size() - from, g->dev_in + p->in_base, true));
        if (single) g->marks[1] = be_->mark(s);
        for (int l = 0; l < kKinds && !rc; l++)
            if (!p->gates[l].empty()) step(be_->run_gates(s, l, p->gates[l].data(), p->gates[l].size()));
        if (single) g->marks[2] = be_->mark(s);
        if (!rc && !p->downloads.empty())
            step(be_->copy_ctxts(s, p->downloads.data(), p->downloads.size(), g->dev_out + p->out_base, false));
    }
    if (rc == 0 && g->out_words && !zero_copy_out) step(be_->d2h(s, g->pin_out, g->dev_out, g->out_words * 4));
    g->zero_copy_in = zero_copy;
    g->zero_copy_out = zero_copy_out;
    g->marks[3] = be_->mark(s);
    if (g->done->ev) {
        const int r = be_->event_record(s, g->done->ev);
        if (r && !rc) {
            rc = r;
            g->error_text = be_->error_text();
        }
    }
    g->error = rc;
    g->trace.t_submit_end = now_ns();
    g->state.store(1, std::memory_order_release);
    return rc;
}

// the group's event has completed: hand results to the host ciphertexts, recycle everything
inline int DeviceSched::retire(Group* g)
{
    if (g->state.load(std::memory_order_acquire) == 2) return 0;
    ScopedNs timer(&stats_.retire_ns);
    g->trace.t_done_seen = now_ns();
    int rc = g->error;
    if (rc) {
        sticky_error_ = rc;
        err_ = g->error_text;
    }
    for (Plan* p : g->plans) {
        auto deliver = [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                const Delivery& dl = p->deliveries[i];
                cufhe_amd_ctxt* c = dl.c;
                if (c->host_token != dl.token) continue;          // superseded by a newer result (at most one entry per ciphertext is current)
                if (c->host && !rc)
                    memcpy(c->host, (const uint32_t*)g->pin_out + p->out_base + dl.slot, (size_t)be_->words(c->level) * 4);
                c->host_dev = -1;
            }
        };
        CopyHelpers* pool = p->deliveries.size() >= parallel_copy_min ? helpers() : nullptr;
        if (pool) {
            constexpr size_t kChunk = 256;
            pool->run((p->deliveries.size() + kChunk - 1) / kChunk, [&](size_t j) { deliver(j * kChunk, std::min(p->deliveries.size(), (j + 1) * kChunk)); });
        } else deliver(0, p->deliveries.size());
        // an upload snapshot lives in this level's staging copy: keep it only for inputs that were re-used
        for (size_t i = 0; i < p->uploads.size(); i++) {
            cufhe_amd_ctxt::PerDev& pd = p->upload_ctxts[i]->d[device_];
            if (pd.snap_plan != (void*)p || pd.snap_off != p->uploads[i].slot) continue;
            if (pd.snap_hits > 0 && pd.snap_version == pd.version) {
                const uint32_t* w = (const uint32_t*)g->pin_in + p->in_base + pd.snap_off;
                pd.snap_own.assign(w, w + be_->words(p->uploads[i].level));
                pd.snap_owned = true;
            }
            pd.snap_plan = nullptr;
        }
        p->reset();
        plan_pool_.push_back(p);
    }
    g->plans.clear();
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (g->pin_in) pinned_cache_.push_back({g->pin_in, g->pin_in_cap});
        if (g->pin_out) pinned_cache_.push_back({g->pin_out, g->pin_out_cap});
        if (g->dev_in && !g->zero_copy_in) dev_cache_.push_back({g->dev_in, g->dev_in_cap});
        if (g->dev_out && !g->zero_copy_out) dev_cache_.push_back({g->dev_out, g->dev_out_cap});
    }
    g->deps.clear();
    if (g->marks[0] && g->marks[3]) {
        if (g->marks[1] && g->marks[2]) {
            g->trace.dev_h2d_ms = be_->elapsed_ms(g->marks[0], g->marks[1]);
            g->trace.dev_body_ms = be_->elapsed_ms(g->marks[1], g->marks[2]);
            g->trace.dev_d2h_ms = be_->elapsed_ms(g->marks[2], g->marks[3]);
        } else {
            g->trace.dev_body_ms = be_->elapsed_ms(g->marks[0], g->marks[3]);
        }
    }
    for (void*& m : g->marks)
        if (m) { be_->event_destroy(m); m = nullptr; }
    for (void* ev : g->lane_events) be_->event_destroy(ev);
    g->lane_events.clear();
    g->trace.t_delivered = now_ns();
    trace_.push_back(g->trace);
    if (trace_.size() > 64) trace_.pop_front();
    g->state.store(2, std::memory_order_release);
    while (!live_.empty() && live_.front()->state.load(std::memory_order_acquire) == 2) {
        delete live_.front();
        live_.pop_front();
    }
    owner_->collect_zombies();
    collect_retired();
    return rc;
}

inline int DeviceSched::synchronize()
{
    restore_homes(nullptr, true);
    if (int rc = flush()) return rc;
    wait_worker_idle();
    be_->bind_thread();
    int rc = 0;
    for (;;) {      // retire() may delete groups: look the next one up afresh every time
        Group* g = nullptr;
        for (Group* x : live_)
            if (x->state.load(std::memory_order_acquire) != 2) {
                g = x;
                break;
            }
        if (!g) break;
        if (g->done->ev && !g->error)
            if (int r = be_->event_sync(g->done->ev)) {
                rc = rc ? rc : r;
                err_ = be_->error_text();
            }
        if (int r = retire(g)) rc = rc ? rc : r;
    }
    if (sticky_error_ && !rc) rc = sticky_error_;
    sticky_error_ = 0;
    return rc;
}

inline int DeviceSched::stream_query(void* stream)
{
    restore_homes(stream);                        // (may be the first thing on record for this stream: a value it only re-uploaded)
    auto it = streams_.find(stream);
    if (it == streams_.end()) return 1;
    StreamState& ss = streams_[stream];           // references survive a rehash, iterators do not
    if (ss.max_depth >= base_depth_)
        if (int rc = flush()) return rc;          // it can only complete once it has been launched
    be_->bind_thread();
    size_t keep = 0;
    int busy = 0;
    for (uint64_t id : ss.open) {
        Group* g = nullptr;
        for (Group* x : live_)
            if (x->id == id) g = x;
        if (!g || g->state.load(std::memory_order_acquire) == 2) continue;
        bool complete = false;
        if (g->state.load(std::memory_order_acquire) == 1) {
            if (g->error || !g->done->ev) complete = true;
            else {
                const int q = be_->event_query(g->done->ev);
                if (q < 0) return fail(q, be_->error_text());
                complete = q == 1;
            }
        }
        if (complete) {
            if (int rc = retire(g)) return rc;
        } else {
            ss.open[keep++] = id;
            busy = 1;
        }
    }
    ss.open.resize(keep);
    if (busy) return 0;
    if (sticky_error_) {
        const int rc = sticky_error_;
        sticky_error_ = 0;
        return rc;
    }
    forget_stream(stream);
    return 1;
}

inline int DeviceSched::stream_fence(void* stream)
{
    ext_streams_.insert(stream);
    fence_epoch_[stream]++;
    restore_homes(stream);
    if (streams_.find(stream) == streams_.end()) return 0;      // nothing of this stream is recorded or in flight
    StreamState& ss = streams_[stream];
    if (ss.max_depth >= base_depth_)
        if (int rc = flush()) return rc;
    wait_worker_idle();                                          // every flushed group is submitted: its completion event is recorded
    be_->bind_thread();
    for (uint64_t id : ss.open) {
        Group* g = nullptr;
        for (Group* x : live_)
            if (x->id == id) g = x;
        if (!g || g->state.load(std::memory_order_acquire) == 2 || g->error || !g->done->ev) continue;
        if (int rc = be_->caller_stream_wait(stream, g->done->ev)) return fail(rc, be_->error_text());
    }
    return 0;
}

inline int DeviceSched::stream_synchronize(void* stream)
{
    restore_homes(stream);
    if (streams_.find(stream) == streams_.end()) return 0;
    {
        StreamState& ss = streams_[stream];
        if (ss.max_depth >= base_depth_)
            if (int rc = flush()) return rc;
    }
    wait_worker_idle();
    be_->bind_thread();
    int rc = 0;
    const std::vector<uint64_t> open = streams_[stream].open;     // retire() deletes groups: look each one up afresh
    for (uint64_t id : open) {
        Group* g = nullptr;
        for (Group* x : live_)
            if (x->id == id) g = x;
        if (!g || g->state.load(std::memory_order_acquire) == 2) continue;
        if (g->done->ev && !g->error)
            if (int r = be_->event_sync(g->done->ev)) {
                rc = rc ? rc : r;
                err_ = be_->error_text();
            }
        if (int r = retire(g)) rc = rc ? rc : r;
    }
    if (sticky_error_ && !rc) rc = sticky_error_;
    sticky_error_ = 0;
    forget_stream(stream);
    return rc;
}

}  // namespace sched
}  // namespace cufhe_amd
