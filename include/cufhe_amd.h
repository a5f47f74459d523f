/*
 * cufhe_amd.h -- C ABI of the MI355X gate-bootstrapping engine.
 *
 * This is the drop-in boundary for the gate path of virtualsecureplatform/cuFHE: every
 * entry point below replaces one piece of the reference's host interface (file:line
 * under /root/reference cited per function).  Plain pointers and sizes only; no HIP,
 * torch or C++ types.  All functions return 0 on success and a negative code on failure
 * (cufhe_amd_last_error() gives the text); the reference aborts instead
 * (include/details/error_gpu.cuh:40-60) and the C++ shim include/cufhe_amd.hpp restores
 * that behaviour.  Codes: -1 bad argument, -2 a HIP call failed, -3 keys not initialised,
 * -5 a kernel reported through the device's fault word that its own result cannot be trusted
 * (a bounded device-side wait expired): returned by cufhe_amd_synchronize / _stream_query /
 * _stream_synchronize and by the call that makes the scheduler observe the completion,
 * sticky until cufhe_amd_cleanup; ciphertexts produced since Initialize must be discarded.
 *
 * Conventions
 *   - "device" is the reference's GPU index in [0, gpuNum) (Stream::device_id(),
 *     include/cufhe_gpu.cuh:152-189).
 *   - "stream" is an opaque hipStream_t (NULL = the device's default stream).
 *   - level 0 ciphertexts are lvl0 TLWEs: n+1 = 631 uint32 words (a[0..n-1], b);
 *     level 1 ciphertexts are lvl1 TLWEs: N+1 = 1025 words (TFHEpp::TLWE<P>,
 *     include/cufhe_gpu.cuh:118).  All ciphertext pointers passed to gate functions are
 *     DEVICE pointers, as in the reference's launchers (src/bootstrap_gpu.cu:834-1292).
 *   - op codes: enum cufhe_amd_op.  For CUFHE_AMD_MUX/NMUX the operands are
 *     (in0, in1, in2) = (inc, in1, in0) of Mux(out, inc, in1, in0).
 */
#ifndef CUFHE_AMD_H
#define CUFHE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum cufhe_amd_op {
    CUFHE_AMD_NAND = 0, CUFHE_AMD_NOR, CUFHE_AMD_XNOR, CUFHE_AMD_AND, CUFHE_AMD_OR, CUFHE_AMD_XOR,
    CUFHE_AMD_ANDNY, CUFHE_AMD_ANDYN, CUFHE_AMD_ORNY, CUFHE_AMD_ORYN,
    CUFHE_AMD_MUX, CUFHE_AMD_NMUX, CUFHE_AMD_NOT, CUFHE_AMD_COPY, CUFHE_AMD_NUM_OPS
};

/* parameter set compiled into the library (TFHEpp lvl0/lvl1/lvl10 params) */
typedef struct cufhe_amd_params {
    uint32_t n, N, nbit, k, l, Bgbit, t, basebit, mu;
    uint32_t lvl0_words, lvl1_words;
    uint64_t bk_words;        /* n * (k+1)l * (k+1) * N torus words   */
    uint64_t ksk_words;       /* k N * t * 2^(basebit-1) * (n+1) words */
    uint64_t bk_ntt_bytes;    /* bytes one blind rotation reads        */
} cufhe_amd_params;
int cufhe_amd_get_params(cufhe_amd_params* out);
const char* cufhe_amd_last_error(void);

/* ---- device management: src/cufhe_gates_gpu.cu:38-65, include/cufhe_gpu.cuh:54-74 ---- */
int cufhe_amd_set_gpu_num(int gpu_num);                 /* SetGPUNum             :38 */
int cufhe_amd_get_gpu_num(void);
int cufhe_amd_device_count(void);                       /* physical GPUs visible     */
/* Which physical GPU logical device `device` is: "pci=<domain:bus:dev.fn> uuid=<hex> hip_device=<index> local_cpus=<list>"
 * (hipDeviceGetPCIBusId / hipDeviceGetUuid; local_cpus = /sys/bus/pci/devices/<pci>/local_cpulist, the CPUs the
 * device's launch worker is pinned to).  The reference has no counterpart: it addresses GPUs by index
 * (cudaSetDevice(i), include/cufhe_gpu.cuh:68-74); bench.py uses this to prove that N ranks / N logical devices
 * are N distinct GPUs (test/test_gate_gpu_multi.cc:36-93 assumes it). */
int cufhe_amd_device_identity(int device, char* buf, size_t len);
/* Compute units of logical device `device` -- the unit of every launch-shape and flush rule of this library (one grid round of the
 * blind rotation = a workgroup of 8 rotations per CU); "cus_override" (cufhe_amd_set_option) replaces it. */
int cufhe_amd_device_cus(int device);
/* hipMemGetInfo of the device: what Initialize leaves free (every key is replicated per GPU, src/bootstrap_gpu.cu:115-137) */
int cufhe_amd_device_mem_info(int device, uint64_t* free_bytes, uint64_t* total_bytes);
int cufhe_amd_initialize_ntt(void);                     /* Initialize()          :40 */
/* Initialize(const EvalKey&) :42-47 = InitializeNTThandlers + BootstrappingKeyToNTT
 * (src/bootstrap_gpu.cu:111-138) + KeySwitchingKeyToDevice (src/keyswitch_gpu.cu:6-16).
 * bk: host, [n][(k+1)l][k+1][N] torus words; ksk: host, [kN][t][2^(basebit-1)][n+1]. */
int cufhe_amd_initialize(const uint32_t* bk, size_t bk_words, const uint32_t* ksk, size_t ksk_words);
int cufhe_amd_cleanup(void);                            /* CleanUp               :49 */
int cufhe_amd_synchronize(void);                        /* Synchronize  cufhe_gpu.cuh:68-74 */

/* ---- streams: class Stream include/cufhe_gpu.cuh:152-189, StreamQuery :55-65 ---- */
int cufhe_amd_stream_create(int device, void** stream);
int cufhe_amd_stream_destroy(int device, void* stream);
int cufhe_amd_stream_query(int device, void* stream);   /* 1 = idle, 0 = busy, <0 = error */
/* cudaStreamSynchronize(st.st()) of a reference program: everything issued on `stream` -- recorded gates included -- is launched,
 * complete and delivered to the tlwehosts on return. */
int cufhe_amd_stream_synchronize(int device, void* stream);
/* What makes the raw handle mean what it means in the reference, where a gate IS enqueued on st.st() at the call
 * (src/cufhe_gates_gpu.cu:148-167): called when the handle is handed to the caller (Stream::st() of include/cufhe_amd.hpp does).
 * Everything recorded on `stream` so far is launched -- values it wrote return to the ciphertexts' own buffers (tlwedevices) first --
 * and the raw stream is made to wait for it: a hipMemcpyAsync / hipEventRecord / hipStreamSynchronize the caller issues on the handle
 * afterwards is ordered behind those gates; and from now on gates recorded on `stream` wait for what the caller has put on the raw
 * stream before them (an upload into tlwedevices, a hipStreamWaitEvent).  Host results (tlwehost) are delivered by the completion
 * calls: cufhe_amd_stream_synchronize / _stream_query / _synchronize. */
int cufhe_amd_stream_fence(int device, void* stream);

/* ---- ciphertext storage: ctxtInitialize/ctxtDelete include/cufhe_gpu.cuh:76-95,
 *      CtxtCopyH2D/D2H :193-207 ---- */
int cufhe_amd_malloc(int device, size_t bytes, void** dptr);
int cufhe_amd_free(int device, void* dptr);
int cufhe_amd_host_register(void* hptr, size_t bytes);
int cufhe_amd_host_unregister(void* hptr);
int cufhe_amd_memcpy_h2d(int device, void* stream, void* dptr, const void* hptr, size_t bytes);
int cufhe_amd_memcpy_d2h(int device, void* stream, void* hptr, const void* dptr, size_t bytes);

/* ---- gates on device-resident ciphertexts ----
 * One gate, one launch sequence on `stream`: the g-prefixed gates of
 * src/cufhe_gates_gpu.cu:160-167 etc. (NandBootstrap ... NMuxBootstrap, NotBootstrap,
 * CopyBootstrap).  in1/in2 may be NULL where the op does not read them. */
int cufhe_amd_gate(int device, void* stream, int op, int level, uint32_t* out,
                   const uint32_t* in0, const uint32_t* in1, const uint32_t* in2);
/* The native batched entry: `count` independent gates in one launch sequence.
 * ops[g * ops_stride] is the op of gate g (ops_stride 0: ops[0] for all; ops is a HOST
 * array).  Operand g of each array is at base + g * stride_words. */
int cufhe_amd_gate_batch(int device, void* stream, int level, size_t count,
                         const int32_t* ops, int ops_stride, uint32_t* out,
                         const uint32_t* in0, const uint32_t* in1, const uint32_t* in2,
                         size_t stride_words);
/* Same with per-gate operand pointers (HOST arrays of DEVICE pointers): what the
 * stream scheduler of include/cufhe_amd.hpp flushes. */
int cufhe_amd_gate_list(int device, void* stream, int level, size_t count, const int32_t* ops,
                        uint32_t* const* outs, const uint32_t* const* in0s,
                        const uint32_t* const* in1s, const uint32_t* const* in2s);

/* ---- the reference's per-gate API on host-visible ciphertexts ----
 * template<class P> struct Ctxt (include/cufhe_gpu.cuh:102-121): `host_words` is the
 * caller-owned tlwehost storage (n+1 or N+1 words, kept alive by the caller); the handle
 * adds the per-GPU device buffers (tlwedevices).  Create after SetGPUNum. */
typedef struct cufhe_amd_ctxt cufhe_amd_ctxt;
int cufhe_amd_ctxt_create(int level, uint32_t* host_words, cufhe_amd_ctxt** out);
int cufhe_amd_ctxt_destroy(cufhe_amd_ctxt* c);
uint32_t* cufhe_amd_ctxt_device_ptr(cufhe_amd_ctxt* c, int device);   /* constant for the life of the ciphertext */
/* Nand(out, in0, in1, st) ... NMux, Not, Copy (copying != 0: src/cufhe_gates_gpu.cu:148-158,
 * inputs taken from tlwehost, result delivered to out's tlwehost) and gNand ... (copying == 0:
 * :160-167, device buffers only).  The gate is RECORDED with its data dependences; the recorded
 * program of a device is launched level by level (all gates whose operands are ready: one
 * blind-rotate + one key-switch launch per level) at Synchronize / StreamQuery / cufhe_amd_flush,
 * or as soon as a level holds two grid rounds of gates (one when the device is idle; a round = 8 gates per compute unit).  A flush of several
 * levels (a recorded netlist) may instead be scheduled gate by gate on two lanes ("sched_two_lane" below).  Stream order, output aliasing,
 * shared inputs and completion semantics are those of the reference (see cufhe_amd/csrc/sched_core.h).
 * cufhe_amd_ctxt_destroy never waits: buffers are recycled when the last recorded gate naming
 * them has retired. */
int cufhe_amd_enqueue_gate(int device, void* stream, int op, int copying, cufhe_amd_ctxt* out,
                           cufhe_amd_ctxt* in0, cufhe_amd_ctxt* in1, cufhe_amd_ctxt* in2);
/* TRLWE-level operations through the same scheduler: struct cuFHETRLWElvl1 (include/cufhe_gpu.cuh:124-134) is a
 * ciphertext handle of level 2 ((k+1) N words; cufhe_amd_ctxt_create(2, ...)), and
 * gGateBootstrappingTLWE2TRLWElvl01NTT / gRefresh / gSampleExtractAndKeySwitch (src/cufhe_gates_gpu.cu:86-146;
 * copying != 0: their upload-and-fetch forms) are recorded with their dependences and launched level by level:
 * 4096 Refresh calls on 800 streams (test/test_perf.cc:63-81) become a handful of launches. */
enum cufhe_amd_trlwe_op { CUFHE_AMD_TL_BOOTSTRAP = 100, CUFHE_AMD_TL_REFRESH = 101, CUFHE_AMD_TL_SEIKS = 102, CUFHE_AMD_TL_CMUX = 103 };
int cufhe_amd_enqueue_trlwe_op(int device, void* stream, int op, int copying, cufhe_amd_ctxt* out, cufhe_amd_ctxt* in);
/* CMUXNTT(res, cs, c1, c0, st) (src/cufhe_gates_gpu.cu:68-85; kernel __CMUXNTT__ src/bootstrap_gpu.cu:197-285): res = c0 + cs [x] (c1 - c0),
 * recorded like a gate and ordered against its operands by the scheduler -- as in the reference it returns at once and the
 * result is in res.trlwehost after Synchronize() / StreamQuery(st).  `cs` is a handle of level 3 (struct cuFHETRGSWNTTlvl1,
 * include/cufhe_gpu.cuh:136-146: cufhe_amd_ctxt_create(3, trgswhost as words, ..): (k+1)l (k+1) N doubles in this library's NTT
 * domain), res / c1 / c0 are TRLWE handles (level 2).  copying != 0: operands from their host members, result delivered to
 * the host (the reference's only form); 0: device buffers only.  Needs Initialize() only, like the reference. */
int cufhe_amd_enqueue_cmux(int device, void* stream, int copying, cufhe_amd_ctxt* res, cufhe_amd_ctxt* cs,
                           cufhe_amd_ctxt* c1, cufhe_amd_ctxt* c0);
/* CtxtCopyH2D / CtxtCopyD2H (include/cufhe_gpu.cuh:193-207), ordered with the recorded gates */
int cufhe_amd_enqueue_copy(int device, void* stream, cufhe_amd_ctxt* c, int to_device);
int cufhe_amd_flush(int device);                        /* launch what is recorded, do not wait */
int cufhe_amd_sched_stream_query(int device, void* stream);  /* scheduler half of StreamQuery */
/* what the scheduler did (no reference counterpart; the reference launches per gate,
 * src/bootstrap_gpu.cu:834-1292) */
typedef struct cufhe_amd_sched_stats {
    uint64_t gates;              /* gates recorded */
    uint64_t groups;             /* flushes handed to the device */
    uint64_t levels;             /* dependence levels launched */
    uint64_t launch_sequences;   /* blind-rotate + key-switch launch pairs */
    uint64_t uploads, uploads_shared, downloads;   /* ciphertext copies; _shared = unchanged inputs not copied again */
    uint64_t forced_syncs;       /* a result had to reach tlwehost before a gate could be recorded */
    uint64_t max_level_gates;
    uint64_t cross_stream_waits; /* event dependences between flushes on different internal streams */
    uint64_t record_ns, retire_ns;  /* host time on the issuing thread: recording gates, delivering results */
    uint64_t launch_ns;             /* host time on the device's launch worker */
    uint64_t renames;               /* outputs that took a fresh device buffer ("sched_rename") */
    uint64_t worker_cpus;           /* CPUs the device's launch worker is pinned to (0: not pinned; "sched_affinity") */
    uint64_t home_copies;           /* renamed values copied back to the ciphertext's own buffer before the host could look */
    uint64_t two_lane_groups;       /* flushes scheduled gate by gate on two lanes ("sched_two_lane") ... */
    uint64_t two_lane_launches;     /* ... and the launches they were cut into */
} cufhe_amd_sched_stats;
int cufhe_amd_sched_get_stats(int device, cufhe_amd_sched_stats* out, int reset);
/* Timeline of the most recent flushes of a device (oldest first, at most 64 kept): host times are std::chrono::steady_clock
 * nanoseconds (time_since_epoch), device spans milliseconds between HIP timing events (0 unless cufhe_amd_profile_enable was
 * on; marked per phase for flushes of one dependence level, as one span otherwise).  This is what the PCIe-inclusive rate of the
 * per-gate API (enqueue -> Synchronize, test/test_util.h:29-72) is made of; the reference has no counterpart. */
typedef struct cufhe_amd_group_trace {
    uint64_t id;
    uint32_t levels, gates, stream, pad;
    uint64_t in_bytes, out_bytes;    /* ciphertext words uploaded / downloaded by this flush */
    int64_t t_queued;                /* issuing thread: handed to the device's launch worker */
    int64_t t_launch_begin;          /* worker: picked up */
    int64_t t_gather_end;            /* worker: inputs copied out of tlwehost into the pinned block */
    int64_t t_submit_end;            /* worker: everything submitted to the stream */
    int64_t t_done_seen;             /* issuing thread: completion observed */
    int64_t t_delivered;             /* issuing thread: results copied into tlwehost */
    float dev_h2d_ms, dev_body_ms, dev_d2h_ms;   /* H2D copy + scatter | gates | gather + D2H copy */
    float pad2;
} cufhe_amd_group_trace;
int cufhe_amd_sched_get_trace(int device, cufhe_amd_group_trace* out, int max, int clear);

/* ---- pieces of the path (TRLWE-level primitives and parity hooks) ----
 * BootstrapTLWE2TRLWE (src/bootstrap_gpu.cu:806-815): tlwe0[count][n+1] -> acc[count][2N]
 * after `steps` CMux steps (steps < 0: all n); also the accumulator parity hook. */
int cufhe_amd_blind_rotate_batch(int device, void* stream, size_t count, const uint32_t* tlwe0,
                                 uint32_t* acc, int steps);
/* Bootstrap (src/bootstrap_gpu.cu:290-301,782-788): refresh lvl0 TLWEs without a gate,
 * in[count][n+1] -> blind rotate (test vector mu = lvl1 mu) -> extract -> key switch -> out[count][n+1] */
int cufhe_amd_bootstrap_batch(int device, void* stream, size_t count, uint32_t* out, const uint32_t* in);
/* SEIandKS (src/keyswitch_gpu.cu:26-40) on already extracted lvl1 TLWEs:
 * tlwe1[count][N+1] -> tlwe0[count][n+1] */
int cufhe_amd_keyswitch_batch(int device, void* stream, size_t count, const uint32_t* tlwe1,
                              uint32_t* tlwe0);
/* SampleExtractAndKeySwitch on TRLWEs (src/cufhe_gates_gpu.cu:126-146): trlwe[count][2N] -> tlwe0 */
int cufhe_amd_sample_extract_keyswitch_batch(int device, void* stream, size_t count,
                                             const uint32_t* trlwe, uint32_t* tlwe0);
/* Refresh (src/cufhe_gates_gpu.cu:106-124, SEIandBootstrap2TRLWE src/bootstrap_gpu.cu:325-364):
 * trlwe_in[count][2N] -> sample extract -> key switch -> blind rotate -> trlwe_out[count][2N] */
int cufhe_amd_refresh_batch(int device, void* stream, size_t count, const uint32_t* trlwe_in,
                            uint32_t* trlwe_out);
/* TRGSW2NTT (src/bootstrap_gpu.cu:75-94): trgsw[count][(k+1)l][k+1][N] torus words ->
 * trgsw_ntt[count][(k+1)l][k+1][N] doubles (this library's NTT domain; opaque to callers) */
int cufhe_amd_trgsw_to_ntt_batch(int device, void* stream, size_t count, const uint32_t* trgsw,
                                 double* trgsw_ntt);
/* TRGSW2NTT on host memory, as the reference's wrapper runs it (H2D, kernel, D2H on `stream`, complete on return): staging
 * comes from the stream's workspace and recycled pinned blocks -- no allocation per call.  trgsw_host: (k+1)l (k+1) N
 * torus words; trgsw_ntt_host: as many doubles. */
int cufhe_amd_trgsw_to_ntt_host(int device, void* stream, const uint32_t* trgsw_host, double* trgsw_ntt_host);
/* TRGSW2NTT on a TRGSW holder (ciphertext handle of level 3, struct cuFHETRGSWNTTlvl1): fills the holder's host words (complete
 * on return) and records their upload to the stream's device, so that both CMUXNTT forms see them -- the reference leaves the
 * result in trgswhost and trgswdevices[st.device_id()] (src/bootstrap_gpu.cu:75-94).  Ordered against recorded uses of the holder. */
int cufhe_amd_trgsw_to_ntt(int device, void* stream, const uint32_t* trgsw_host, cufhe_amd_ctxt* trgswntt);
/* CMUXNTT (src/bootstrap_gpu.cu:197-285): res = c0 + trgsw [x] (c1 - c0), TRLWEs [count][2N]; res may be c0 or c1 */
int cufhe_amd_cmux_batch(int device, void* stream, size_t count, const double* trgsw_ntt,
                         const uint32_t* c1, const uint32_t* c0, uint32_t* res);
/* NTT product check of test/test_polynomial_mult_1024.cu: res = a * b negacyclic mod 2^32,
 * a signed with |a| <= 128 (exactness bound of the field), all [count][N], device. */
int cufhe_amd_polymul_batch(int device, void* stream, size_t count, const int32_t* a,
                            const uint32_t* b, uint32_t* res);
/* the same check for the 512-point transform (SmallForwardNTT_512 / SmallInverseNTT_512,
 * include/ntt_gpu/ntt_gpuntt.cuh:283-329): res = a * b mod (X^512 + 1, 2^32), operands [count][512] */
int cufhe_amd_polymul512_batch(int device, void* stream, size_t count, const int32_t* a,
                               const uint32_t* b, uint32_t* res);

/* ---- tuning ----
 * "device_base": physical HIP device that logical device 0 maps to (default 0).  A process
 * that drives one GPU of a node (one rank per GPU) sets it to its local rank before Initialize.
 * Launch shape of the blind rotation (all variants produce identical words; the rules are in units of the device's CU count,
 * the numbers below are MI355X's 256 CUs).  A launch is cut into whole
 * rounds of the batch kernel's grid (2048 rotations: two per SIMD, highest throughput) plus a tail, and the
 * tail -- or a whole small launch -- takes the cheapest kernel by measured cost: up to 256 rotations (one per CU) the 16-wave
 * workgroup-per-rotation kernel with split transforms (lowest latency: 2.8 ms for one, 2.9 for 64, 3.3 for 256); up to
 * 1536 rounds of 512 on its two-rotations-per-workgroup form (5.2 ms each) plus a last round of up to 256 on the
 * single form; above that a full round (20.7 ms).  "ll2_threshold" (default -1 = by cost; 0 = never; n = for every
 * launch up to n) governs the paired form; "ll_threshold" / "half_threshold" (default -1 = by cost) force the single
 * form / the batch kernel with one rotation per SIMD up to the given count, "tail_split" 0 launches everything above
 * one round as one grid.
 * "ks_split_threshold" (default -1 = by measured cost: 32 on 256 CUs): up to that many ciphertexts per launch each key
 * switch is split over 8 workgroups (0.05 ms); above, the ciphertexts of a workgroup share each step of the key in LDS
 * (keyswitch_kernel): "ks_per_wg" ciphertexts per workgroup (1..16) and the 1024 steps of j cut into "ks_slices" runs (a
 * power of two up to 64, partial sums meeting in the output through atomics), both -1 = the cheapest shape by a model of
 * the measured times: 0.07 ms for 64, 0.12 for 256, 0.31 for 1024, 0.57 for 2048, 0.86 for 3072, 1.04 ms for 4096.
 * "ks_wg_threshold" (default -1 = never) forces one workgroup per ciphertext up to the given count (0.22 ms up to 256,
 * 0.8 ms at 1024).  All variants produce identical words.
 * "ps_batch_threshold" (default -1 = by cost: 1025, 1281 for N = 512, 1537 for sets with key limbs): rotations per launch from which the parameter-set path
 * (cufhe_amd_ps_*) uses the wave-per-rotation kernel instead of a workgroup per rotation.
 * "lvl0_ring": 1024 (default) or 2048 -- the ring through which gates on lvl0 ciphertexts
 * bootstrap: lvl01/lvl10 (cufhe_amd_initialize) or lvl02/lvl20 (cufhe_amd_lvl2_initialize).  With 2048
 * every level-0 entry point (cufhe_amd_gate*, the recorded per-gate API, Nand<lvl0param>() ... in
 * the C++ shim) runs through the N = 2048 path.
 * "lvl2_kernel" (default -1 = by measured cost): blind-rotate kernel of the N = 2048 ring: 1 = four quarter-transform waves per rotation with
 * register sums, two rotations per CU (launches above one rotation per CU); 0 = eight half-transform waves, one rotation per CU.
 * "param_set" (default -1; "lvl0_param_set" is the same option under its old name): index of a cufhe_amd_ps_* parameter set on
 * which the whole per-gate API runs instead -- both ciphertext levels and both gate orders, and the three bootstrapping TRLWE-level
 * operations of cufhe_amd_enqueue_trlwe_op, CMUXNTT and TRGSW2NTT (cufhe_amd_enqueue_cmux / cufhe_amd_trgsw_to_ntt; not on the
 * small-modulus set, as in the reference), as the set chosen when the reference is built serves every entry point
 * (CMakeLists.txt:8-24); cufhe_amd_ps_initialize first (cufhe_amd_initialize_params does both).  Ciphertexts then have the set's sizes
 * (cufhe_amd_ctxt_words: n + 1 and k N + 1 words; include/cufhe_amd.hpp selects the matching parameter structs with
 * -DCUFHE_AMD_PARAM_SET_K2N512 / -DCUFHE_AMD_PARAM_SET_CGGI16 / -DCUFHE_AMD_PARAM_SET_SMALLMOD).  Changing it waits for everything
 * recorded; a ciphertext keeps the host buffer of the set it was created under, and using it while a set with other sizes is active is
 * refused (-1).
 * "share_devices" (default 0): 1 lets SetGPUNum(G) exceed the visible GPU count, logical devices wrapping around the
 * physical ones (every logical device keeps its own key replica, scheduler, launch thread and streams): the
 * reference's multi-GPU programs (test/test_gate_gpu_multi.cc) rehearsed on fewer GPUs.
 * "sched_streams" (default 4): internal HIP streams per device over which independent flushes of the
 * per-gate API overlap; "sched_threads" (default 1): one launch worker thread per device (0: launches
 * happen on the issuing thread).  Both before the first ciphertext is created.
 * "sched_rename" (default 1): an output whose device buffer still has recorded users (an earlier write,
 * readers of the old value) takes a fresh buffer instead of being ordered after them, so that only true data
 * dependences order a recorded program: a temporary re-used down a ripple-carry chain no longer serialises the
 * independent gates of the adders (16-bit adders: 64 dependence levels become 33).  The buffer a ciphertext was created
 * with (cufhe_amd_ctxt_device_ptr at construction = Ctxt::tlwedevices[i], include/cufhe_gpu.cuh:80-84) stays its home: a
 * value still in a renamed buffer when the caller asks for completion (Synchronize, StreamQuery of the stream that wrote
 * it) is copied home by one Copy gate in the flush that request triggers, so the published pointer holds the value
 * whenever the host may look -- results, tlwehost, tlwedevices and every API call behave as with 0 (never rename).
 * "sched_idle_gates" (default -1 = one grid round): gates of a level at which an IDLE device is handed it (never more than "sched_level_gates").
 * (Two rounds -- one flush and one key-switch launch for a burst of 4096 gates -- was measured and is no faster: the device then starts
 * 0.7 ms later, which is what the second key-switch launch costs.)
 * "sched_copy_threads" (default 4; before the first flush): host threads, the calling one included, that share the two copies on a flush's
 * latency path once it moves 1024 ciphertexts or more -- inputs out of the tlwehosts into the pinned staging block (launch worker) and
 * results back into the tlwehosts (issuing thread): 4096 NANDs' 10 MB of inputs are gathered in 0.2 ms instead of 0.7.
 * "sched_two_lane" (default 1): a flush of several dependence levels (a recorded netlist) is scheduled GATE BY GATE instead of level by
 * level when the library's cost model says that is faster: the gates on long dependence chains run as steps of the paired low-latency
 * kernel on half of the compute units while the gates nothing waits for run as chunks of the batch kernel on the other half, on two
 * internal streams (256 sixteen-bit ripple-carry adders, 32 carry levels of 256 gates behind one level of 8192: see DESIGN.md 2a).
 * Needs "sched_rename" (the recorded program is kept single-assignment); results, completion rules and tlwedevices do not change.
 * 0 = never, 2 = whenever a flush is eligible, whatever the cost model says (tests).
 * "br_shape" (of the calling thread; default 0 = the launch-shape rules above): 1 / 2 / 3 put every blind rotation of the thread's
 * launches on the batch kernel / the paired low-latency kernel / the single one -- what the two lanes use, and tools/two_lane_probe.py.
 * "sched_level_gates" (default -1 = two grid rounds of the device, 16 gates per CU, while it has work -- on MI355X 4096: 32 768 gates through
 * the per-gate API 96.1 k -> 99.4 k gates/s --, one round when it is idle) / "sched_total_gates" (default 32768): a dependence level this
 * full is launched at once / bound on the recorded program.
 * "test_fail_alloc" n (default -1 = off): the (n+1)-th device allocation of an Initialize entry point fails like an exhausted device,
 * once -- a test hook for the error paths (the keys that were loaded stay loaded and usable, nothing leaks).
 * "cus_override" (default 0 = the device's own count): every launch-shape rule above and the scheduler's flush rules behave as on a
 * device with that many compute units (a partitioned device, another chip); results do not depend on it.
 * "sched_zero_copy" (default 1): the batched ciphertext traffic of a flush is read and written by the scatter / gather kernels
 * directly in pinned host memory, the inputs of a flush's first level chunk by chunk while the rest is still being gathered
 * from the tlwehosts; 0 = one H2D / D2H copy per flush through device staging buffers (the round-3 path).
 * "sched_affinity" (default 1): the launch worker thread of a device runs on the CPUs local to that GPU
 * (local_cpulist of its PCI function, intersected with the CPUs the process may use); 0 leaves it to the OS. */
int cufhe_amd_set_option(const char* key, long value);

/* ---- N = 2048 ring, 64-bit torus (BASELINE.json configs[4]; lvl2 / lvl02 / lvl20) ----
 * The reference has no N = 2048 path (its NTT is `if constexpr (N == 1024) ... else if
 * (N == 512)`, include/ntt_gpu/ntt_gpuntt.cuh, and its prime has no 4096-th root); these entry
 * points are what its gate templates compute when instantiated at brP = lvl02, iksP = lvl20:
 * __HomGate__ br -> iks (src/bootstrap_gpu.cu:402-421), the Mux of :515-588, Accumulate
 * (include/gatebootstrapping_gpu.cuh:115-285) and KeySwitchFromTLWE (include/keyswitch_gpu.cuh:
 * 83-134, 64-bit domain, 32-bit target).  Gates take and return lvl0 ciphertexts. */
typedef struct cufhe_amd_lvl2_params {
    uint32_t n, N, nbit, k, l, Bgbit, t, basebit;
    uint32_t lvl0_words, lvl2_words;   /* lvl2 TLWE: N + 1 uint64 words */
    uint64_t mu;                       /* 2^61 */
    uint64_t bk_words;                 /* n * (k+1)l * (k+1) * N uint64 words */
    uint64_t ksk_words;                /* k N * t * 2^(basebit-1) * (n+1) uint32 words */
    uint64_t bk_ntt_bytes;             /* bytes one blind rotation reads */
} cufhe_amd_lvl2_params;
int cufhe_amd_lvl2_get_params(cufhe_amd_lvl2_params* out);
/* bk: host, [n][(k+1)l][k+1][N] uint64 torus words (TRGSW of the lvl0 key bits under the lvl2
 * key); ksk: host, [kN][t][2^(basebit-1)][n+1] uint32.  Independent of cufhe_amd_initialize. */
int cufhe_amd_lvl2_initialize(const uint64_t* bk, size_t bk_words, const uint32_t* ksk, size_t ksk_words);
/* same contract as cufhe_amd_gate_batch at level 0 */
int cufhe_amd_lvl2_gate_batch(int device, void* stream, size_t count, const int32_t* ops, int ops_stride,
                              uint32_t* out, const uint32_t* in0, const uint32_t* in1, const uint32_t* in2,
                              size_t stride_words);
/* tlwe0[count][n+1] -> acc[count][2N] (uint64) after `steps` CMux steps (< 0: all n) */
int cufhe_amd_lvl2_blind_rotate_batch(int device, void* stream, size_t count, const uint32_t* tlwe0,
                                      uint64_t* acc, int steps);
/* tlwe2[count][N+1] (uint64) -> tlwe0[count][n+1] */
int cufhe_amd_lvl2_keyswitch_batch(int device, void* stream, size_t count, const uint64_t* tlwe2,
                                   uint32_t* tlwe0);

/* ---- other parameter sets (CMakeLists.txt:8-24: USE_80BIT_SECURITY / USE_CGGI19 / USE_CONCRETE select TFHEpp
 * parameter headers at build time; k > 1: src/bootstrap_gpu.cu:402-421; N = 512: include/ntt_gpu/ntt_gpuntt.cuh:283-329)
 * Every set of cufhe_amd/csrc/kernels_ps.hip.h is compiled in and chosen by index: 0 = the default set (the same
 * numbers as cufhe_amd_get_params, computed by the generic kernels), 1 = "k2n512" (k = 2 over the N = 512 ring),
 * 2 = "cggi16" (the original TFHE 80-bit set; its external product exceeds the exact range of the FP64 field, so
 * the key is taken in two 16-bit limbs), 3 = "smallmod" (the default numbers through the reference's optional small NTT modulus,
 * CMakeLists.txt:12 -DUSE_SMALL_NTT_MODULUS, include/ntt_gpu/ntt_small_modulus.cuh: the bootstrapping key and every CMux increment are
 * switched between the 2^32 and the P = 625 * 2^20 + 1 discretisation of the torus -- approximate by design, word-for-word the
 * reference's arithmetic).  The numeric parameters of TFHEpp's headers are not in the reference tree
 * (SURVEY.md F3): a set is what cufhe_amd_ps_get_params reports.  Gates take and return ciphertexts of the set at
 * either level (n + 1 or k N + 1 words), keys use TFHEpp's layouts with the set's dimensions. */
typedef struct cufhe_amd_ps_params {
    char name[32];
    uint32_t n, N, nbit, k, l, Bgbit, t, basebit, key_limbs, key_limb_bits, mu;
    uint32_t lvl0_words, lvl1_words;
    uint32_t small_ntt_modulus;     /* 0: exact products; P: the set runs the reference's -DUSE_SMALL_NTT_MODULUS arithmetic mod P */
    uint64_t bk_words, ksk_words, bk_ntt_bytes;
} cufhe_amd_ps_params;
/* The numbers a caller was compiled with (TFHEpp's lvl0param::n; lvl1param::nbit, k, l, Bgbit; lvl10param::t, basebit) and whether it
 * was built with the reference's -DUSE_SMALL_NTT_MODULUS (then small_ntt_modulus = 625 * 2^20 + 1, else 0).
 * cufhe_amd_find_param_set: index of the compiled set with exactly these numbers, or -1 (text names the numbers).
 * cufhe_amd_initialize_params: Initialize(const EvalKey&) (src/cufhe_gates_gpu.cu:42-47) with the reference's single selector -- the
 * set is FOUND from the numbers, its keys are loaded on every device and the whole per-gate API runs on it (set 0 takes the
 * hand-scheduled kernels of cufhe_amd_initialize, the others cufhe_amd_ps_initialize + "param_set").  A caller whose Bgbit, t or
 * basebit differ from every compiled set is refused here instead of being served by kernels of other numbers that happen to have
 * keys of the same size.  include/cufhe_amd.hpp calls nothing else, and derives the same match at compile time. */
typedef struct cufhe_amd_param_numbers {
    uint32_t n, nbit, k, l, Bgbit, t, basebit, small_ntt_modulus;
} cufhe_amd_param_numbers;
int cufhe_amd_find_param_set(const cufhe_amd_param_numbers* numbers);
int cufhe_amd_initialize_params(const cufhe_amd_param_numbers* numbers, const uint32_t* bk, size_t bk_words, const uint32_t* ksk, size_t ksk_words);
int cufhe_amd_ps_count(void);
int cufhe_amd_ps_get_params(int set, cufhe_amd_ps_params* out);
int cufhe_amd_ps_initialize(int set, const uint32_t* bk, size_t bk_words, const uint32_t* ksk, size_t ksk_words);
/* words of a level-0 / level-1 ciphertext, (level 2) of a TRLWE or (level 3) of a TRGSW holder in the NTT domain (uint32 words: two per
 * double; the active set's key limbs included), of the per-gate API as configured now ("param_set") */
int cufhe_amd_ctxt_words(int level);
/* the same gates on ciphertexts of `level`: 0 = blind rotate then key switch on n + 1 words, 1 = key switch then blind rotate on
 * k N + 1 words (the reference's two __HomGate__ orders, src/bootstrap_gpu.cu:383-421, Mux :515-588 / :706-780) */
int cufhe_amd_ps_gate_batch_level(int set, int device, void* stream, int level, size_t count, const int32_t* ops, int ops_stride,
                                  uint32_t* out, const uint32_t* in0, const uint32_t* in1, const uint32_t* in2, size_t stride_words);
/* same contract as cufhe_amd_gate_batch at level 0 */
int cufhe_amd_ps_gate_batch(int set, int device, void* stream, size_t count, const int32_t* ops, int ops_stride,
                            uint32_t* out, const uint32_t* in0, const uint32_t* in1, const uint32_t* in2,
                            size_t stride_words);
/* tlwe0[count][n+1] -> acc[count][(k+1)N] after `steps` CMux steps (< 0: all n) */
int cufhe_amd_ps_blind_rotate_batch(int set, int device, void* stream, size_t count, const uint32_t* tlwe0,
                                    uint32_t* acc, int steps);
/* tlwe1[count][kN+1] -> tlwe0[count][n+1] */
int cufhe_amd_ps_keyswitch_batch(int set, int device, void* stream, size_t count, const uint32_t* tlwe1,
                                 uint32_t* tlwe0);
/* TRGSW2NTT / CMUXNTT on a set, device-resident (src/bootstrap_gpu.cu:75-94,197-285: in the reference the set chosen at build time
 * serves them too; the small-modulus set has none, :73-95): trgsw[count][(k+1)l][k+1][N] torus words -> trgsw_ntt[count][key_limbs]
 * [(k+1)l][k+1][N] doubles (cufhe_amd_ctxt_words(3) / 2 per TRGSW while the set is active); res = c0 + trgsw [x] (c1 - c0) on TRLWEs
 * [count][(k+1)N], res may be c0 or c1.  Need Initialize() only, like the reference. */
int cufhe_amd_ps_trgsw_to_ntt_batch(int set, int device, void* stream, size_t count, const uint32_t* trgsw, double* trgsw_ntt);
int cufhe_amd_ps_cmux_batch(int set, int device, void* stream, size_t count, const double* trgsw_ntt, const uint32_t* c1,
                            const uint32_t* c0, uint32_t* res);
/* the bootstrapping TRLWE-level operations on a set, device-resident: op = CUFHE_AMD_TL_BOOTSTRAP (in: tlwe0[count][n+1], out:
 * trlwe[count][(k+1)N]; BootstrapTLWE2TRLWE, src/bootstrap_gpu.cu:806-815), CUFHE_AMD_TL_REFRESH (trlwe -> trlwe; :325-364) or
 * CUFHE_AMD_TL_SEIKS (trlwe -> tlwe0; SEIandKS, src/keyswitch_gpu.cu:26-40) */
int cufhe_amd_ps_trlwe_op_batch(int set, int device, void* stream, int op, size_t count, uint32_t* out, const uint32_t* in);

/* ---- measurement ----
 * When enabled, every blind-rotate / key-switch launch is bracketed by HIP events on the
 * stream it runs on; get_profile synchronises and returns accumulated kernel time. */
typedef struct cufhe_amd_profile {
    double blind_rotate_ms; uint64_t blind_rotate_launches; uint64_t blind_rotations;
    double keyswitch_ms;    uint64_t keyswitch_launches;    uint64_t keyswitches;
} cufhe_amd_profile;
int cufhe_amd_profile_enable(int device, int on);
int cufhe_amd_profile_get(int device, cufhe_amd_profile* out, int reset);
/* Shader clock (Hz) the device holds under an FP64 load, measured now by a short calibration kernel (shader cycles over
 * the constant 100 MHz counter, median over all waves).  Measurement aid: no counterpart in the reference. */
int cufhe_amd_probe_clock(int device, double* hz);

#ifdef __cplusplus
}
#endif
#endif
