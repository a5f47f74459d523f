// capi.hip -- implementation of the C ABI declared in include/cufhe_amd.h.
// Host side of the gate path: device/key management (reference: src/cufhe_gates_gpu.cu:38-65,
// src/bootstrap_gpu.cu:97-155, src/keyswitch_gpu.cu:6-24) and the launch sequences that
// replace the per-gate launchers of src/bootstrap_gpu.cu:834-1292.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/cufhe_amd.h"
#include "sched_core.h"
#include "kernels.hip.h"
#include "kernels_lvl2.hip.h"
#include "kernels_lvl2q.hip.h"
#include "kernels_ks2.hip.h"
#define CUFHE_AMD_LL_DECLARATIONS_ONLY      // defined in kernels_ll.hip
#include "kernels_ll.hip.h"
#include "kernels_ps.hip.h"

using namespace cufhe_amd;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(-2, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                                std::to_string(__LINE__) + ")");                               \
    } while (0)

struct EventPair { hipEvent_t a, b; uint64_t units; };

// Pinned host staging for descriptor uploads.  hipMemcpyAsync from pageable memory may
// return before the bytes have been read, so descriptors are staged in pinned blocks that
// are recycled only once the event recorded behind their copy has completed.
// A block is free again when its owner has RECORDED `done` behind the last device access, that event has completed, and the owner
// is not itself still reading the block on the host (`held`, cufhe_amd_trgsw_to_ntt_host): completion of a never-recorded or stale
// event says nothing about the current owner.
struct PinnedBlock { void* host = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; bool busy = false, recorded = false, held = false; };

// Per-stream device workspace (temporaries + descriptor arrays of one launch sequence).
// Work on one stream is ordered, so the next sequence on the same stream may overwrite the
// workspace of the previous one; distinct streams get distinct workspaces.  Grow-only.
struct Workspace { char* base = nullptr; size_t bytes = 0; };

struct DeviceState {
    bool ntt_ready = false, keys_ready = false;
    int cus = 0;                         // hipDeviceProp_t::multiProcessorCount (launch-shape rules: one grid round = a workgroup per CU)
    NttTables* tables = nullptr;
    NttTables* tables_r4 = nullptr;      // the same transform's tables for the radix-4 passes (ntt_r4.h): blind_rotate_kernel
    Ntt512Tables* tables512 = nullptr;   // [3]: the two 512-point halves (low-latency kernel), stand-alone N = 512
    double* bk_ntt = nullptr;
    uint32_t* ksk = nullptr;
    // device fault word: pinned, host-coherent, mapped into the device (kernels_common.hip.h: kFault*); sticky until CleanUp
    uint32_t* fault_host = nullptr;
    uint32_t* fault = nullptr;
    bool profiling = false;
    bool br_lds_opt_in = false, ks_lds_opt_in = false;
    // N = 2048 / 64-bit torus (lvl2.inc.h)
    bool keys2_ready = false, br2_lds_opt_in = false, ks2_lds_opt_in = false;
    NttTables* tables2 = nullptr;      // [2]: the two half transforms
    double* bk2_ntt = nullptr;
    Ntt512Tables* tables2q = nullptr;  // [4]: the four quarter transforms (kernels_lvl2q.hip.h)
    double* bk2q_ntt = nullptr;        // the same key in the quarter layout
    bool br2q_lds_opt_in = false;
    uint32_t* ksk2 = nullptr;
    std::vector<EventPair> br_events, ks_events;
    cufhe_amd_profile prof{};
    std::deque<PinnedBlock> staging;
    std::map<hipStream_t, Workspace> workspaces;
    std::mutex staging_mu;
};

// "cus_override" > 0: every launch-shape and flush rule behaves as on a device with that many CUs (a partitioned device, another
// chip; the rules are all in CU units) -- what tests use to check that nothing is tied to MI355X's 256
long g_cus_override = 0;
int cus_of(const struct DeviceState& s);
int g_gpu_num = 1;
int g_device_base = 0;         // physical HIP device of logical device 0 (one process per GPU: LOCAL_RANK)
long g_share_devices = 0;      // 1: logical devices beyond the visible GPUs wrap around (SetGPUNum(G) rehearsed on fewer GPUs)
int g_phys_count = 0;
int phys_device(int device)
{
    if (g_share_devices && g_phys_count > 0) return (g_device_base + device) % g_phys_count;
    return device + g_device_base;
}
long g_ks_split_threshold = -1; // key switches per launch up to which each ciphertext is split over 8 workgroups
// Key switch launch shape, -1 = the measured rule (tools/ks_slices.py, MI355X, ms per launch of n key switches):
//   8 workgroups per ciphertext   0.046 (n = 1)  0.049 (16)  0.051 (32)  0.078 (64)  0.13 (128)  0.24 (256)  0.45 (512)  0.85 (1024)
//   a workgroup per ciphertext    0.22 (n <= 256)  0.42 (512)  0.80 (1024)  1.19 (1536)  1.63 (2048)  3.3 (4096)
//   table through LDS (keyswitch_kernel), 16 ciphertexts per workgroup and the steps of j cut into runs that fill the CUs:
//                                 0.040 (1)  0.052 (16)  0.059 (32)  0.070 (64)  0.084 (128)  0.12 (256)  0.19 (512)  0.32 (1024)
//                                 0.56 (1536)  0.58 (2048)  0.86 (3072)  1.04 - 1.09 (4096)
// so: split up to 32, the shared-table kernel above -- on 256 CUs; in units of the device's CU count: 1/8 ciphertext per CU.
// The workgroup-per-ciphertext kernel is no longer chosen by the rule ("ks_wg_threshold" still forces it).
inline long ks_auto_split(int cus) { return std::max(1, cus) / 8; }
inline long ks_auto_wg(int) { return 0; }
long g_ll2_threshold = -1;      // two-rotations-per-workgroup low-latency kernel: -1 by cost, 0 never, > 0 for launches up to this size
long g_ks_wg_threshold = -1;    // key switches per launch up to which the workgroup-per-ciphertext kernel is used
long g_ks_per_wg = -1;          // ciphertexts per workgroup of the shared-table key switch: -1 by count, else 1..16
long g_ks_slices = -1;          // runs the shared-table key switch cuts j into: -1 by count, else a power of two 1..64
// The shape of a shared-table launch (keyswitch_kernel: per_wg ciphertexts per workgroup, the kn steps of j cut into `slices` runs):
// the cheapest by a model of the measured times -- a workgroup of 16 live waves takes 1.03 us per step (0.68 + 0.022 per live wave),
// 12 us around its steps; the workgroups run in rounds of one per CU (hipDeviceProp_t::multiProcessorCount, cached in
// DeviceState); a launch with runs zeroes the outputs first.  kn = 1024 -- 4096 ciphertexts: 256 workgroups x 1024 steps; 3072:
// 768 x 256 (three rounds); 2048: 256 x 512; 256: 256 x 64.  min_slices: a workgroup keeps the digit words of at most 1024 steps.
void ks_auto_shape(size_t count, int cus, int kn, int min_slices, int* per_wg, int* slices)
{
    const size_t c = cus > 0 ? (size_t)cus : 256;
    if (g_ks_slices > 0 || g_ks_per_wg > 0) {            // forced (tests, sweeps): the other one by the round-5 rule
        const size_t p = (count + c - 1) / c;
        *per_wg = g_ks_per_wg > 0 ? (int)g_ks_per_wg : (int)(p < 1 ? 1 : p > 16 ? 16 : p);
        *slices = std::max(min_slices, g_ks_slices > 0 ? (int)g_ks_slices : 1);
        return;
    }
    auto cost = [&](int p, int sl) {                      // us
        const size_t wgs = (count + p - 1) / p * sl, rounds = (wgs + c - 1) / c;
        return rounds * (kn / sl * (0.68 + 0.022 * p) + 12.0) + (sl > 1 ? 20.0 : 15.0);
    };
    const size_t fit = (count * min_slices + c - 1) / c;  // fewest ciphertexts per workgroup that still fit one round
    int best_p = (int)(fit < 1 ? 1 : fit > 16 ? 16 : fit), best_sl = min_slices;
    double best = cost(best_p, best_sl);
    for (int sl = min_slices; sl <= 64; sl *= 2) {
        const double t = cost(kKsWaves, sl);
        if (t < best) { best = t; best_p = kKsWaves; best_sl = sl; }
    }
    *per_wg = best_p;
    *slices = best_sl;
}
long g_ll_threshold = -1;      // rotations per launch up to which the 16-wave split-transform kernel is used; -1: by measured cost (below)
long g_half_threshold = -1;    // ... up to which the batch kernel runs one rotation per SIMD (4 per workgroup); -1: by measured cost
// "br_shape": 0 = by the rules of launch_blind_rotate; 1 / 2 / 3 = every launch whole on the batch kernel (8 rotations per workgroup) / the
// paired low-latency kernel (2 per workgroup) / the single one.  thread_local: the scheduler's launch worker picks a shape per launch.
thread_local long g_br_shape = 0;
long g_tail_split = 1;         // 1: launches above one grid round are cut into full rounds + a tail that takes the cheapest kernel
// N = 2048 blind rotation: 1 = four quarter waves per rotation, two rotations per CU (kernels_lvl2q.hip.h); 0 = eight half waves, one
// rotation per CU (kernels_lvl2.hip.h); -1 = by measured cost: a launch that leaves CUs with a single rotation (<= one per CU) is
// faster on the eight-wave kernel (256 rotations: 14.5 ms against 17.4), everything above on the four-wave one (512: 27.3 against 28.0,
// 4096: 192 against 223)
long g_lvl2_kernel = -1;
long g_lvl0_ring = 1024;       // ring through which gates on lvl0 ciphertexts bootstrap: 1024 (lvl01/lvl10) or 2048 (lvl02/lvl20)
constexpr int kMaxLogicalDevices = 64;    // SetGPUNum bound (per-device tables of fixed size: paramsets.inc.h)
std::deque<DeviceState> g_dev(1);    // re-created only while no device is initialised (SetGPUNum)
int cus_of(const DeviceState& s) { return g_cus_override > 0 ? (int)g_cus_override : s.cus; }
std::mutex g_mu;

// ---- exact host arithmetic for the tables ----
typedef unsigned __int128 u128;
uint64_t mulmod_u64(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % fpf::P_U64); }
uint64_t powmod_u64(uint64_t a, uint64_t e)
{
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mulmod_u64(r, a);
        a = mulmod_u64(a, a);
        e >>= 1;
    }
    return r;
}
double balanced(uint64_t v) { return v > fpf::P_U64 / 2 ? -(double)(fpf::P_U64 - v) : (double)v; }
uint32_t bitrev(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}
double n_inverse_balanced() { return balanced(powmod_u64(kN, fpf::P_U64 - 2)); }

// Device tables of one 1024-point transform from its root arrays, fwd[m + g] / inv[m + g]
// being the twiddle of group g at the stage with m groups.
// r4: the tables of the radix-4 passes (ntt_r4.h).  A 16-register block uses the twiddles [w | u, I u | w_0..w_3 | u_0, I u_0, ..];
// the radix-4 butterfly never multiplies by the second twiddle of a fine pair (root[2m + 1] = I root[2m], -I for the inverse)
// but by the product of the first with the coarse twiddle: slot 2 = u w, slot 8 + 2g = u_g w_g (tc: slot 5 + 2g = u_g w_g).
double mul_balanced(double a, double b)
{
    const uint64_t ua = a < 0 ? fpf::P_U64 - (uint64_t)(-a) : (uint64_t)a, ub = b < 0 ? fpf::P_U64 - (uint64_t)(-b) : (uint64_t)b;
    return balanced(mulmod_u64(ua, ub));
}
void to_r4_block(double* t, int stride)       // t[k * stride], k < 15: one block of four stages
{
    t[2 * stride] = mul_balanced(t[1 * stride], t[0]);
    for (int g = 0; g < 4; g++) t[(8 + 2 * g) * stride] = mul_balanced(t[(7 + 2 * g) * stride], t[(3 + g) * stride]);
}
void fill_tables(NttTables& t, const std::vector<double>& fwd, const std::vector<double>& inv, bool r4 = false)
{
    memset(&t, 0, sizeof(t));
    for (int k = 0; k < 15; k++) {
        int lvl = 0;
        while ((2 << lvl) <= k + 1) lvl++;
        const int j = k + 1 - (1 << lvl);
        t.tu_fwd[k] = fwd[(1 << lvl) + j];
        t.tu_inv[k] = inv[(1 << lvl) + j];
        for (int lam = 0; lam < 16; lam++) {
            const int idx = (16 << lvl) + (lam << lvl) + j;
            t.tb_fwd[k * 16 + lam] = fwd[idx];
            t.tb_inv[k * 16 + lam] = inv[idx];
        }
    }
    for (int k = 0; k < kTcCount; k++)
        for (int lane = 0; lane < 64; lane++) {
            const int lam = lane & 15, h = lane >> 4;
            const int idx = k < 4 ? 256 + ((lam << 4) | (h << 2) | k) : 512 + ((lam << 5) | (h << 3) | (k - 4));
            t.tc_fwd[k * 64 + lane] = fwd[idx];
            t.tc_inv[k * 64 + lane] = inv[idx];
        }
    if (r4) {
        to_r4_block(t.tu_fwd, 1);
        to_r4_block(t.tu_inv, 1);
        for (int lam = 0; lam < 16; lam++) {
            to_r4_block(t.tb_fwd + lam, 16);
            to_r4_block(t.tb_inv + lam, 16);
        }
        for (int lane = 0; lane < 64; lane++)
            for (int g = 0; g < 4; g++) {
                t.tc_fwd[(5 + 2 * g) * 64 + lane] = mul_balanced(t.tc_fwd[(4 + 2 * g) * 64 + lane], t.tc_fwd[g * 64 + lane]);
                t.tc_inv[(5 + 2 * g) * 64 + lane] = mul_balanced(t.tc_inv[(4 + 2 * g) * 64 + lane], t.tc_inv[g * 64 + lane]);
            }
        // packed per lane (ntt_wave.h: NttTables::tbp_fwd ..)
        for (int lam = 0; lam < 16; lam++)
            for (int k = 0; k < kTbCount; k++) {
                t.tbp_fwd[lam * kTbpStride + k] = t.tb_fwd[k * 16 + lam];
                t.tbp_inv[lam * kTbpStride + k] = t.tb_inv[k * 16 + lam];
            }
        for (int lane = 0; lane < 64; lane++)
            for (int k = 0; k < kTcCount; k++) {
                t.tcp_fwd[lane * kTcpStride + k] = t.tc_fwd[k * 64 + lane];
                t.tcp_inv[lane * kTcpStride + k] = t.tc_inv[k * 64 + lane];
            }
    }
}

// root[i] = psi^bitrev(i): the reference's table order, src/ntt_gpu/ntt_gpuntt.cu:88-111
void build_tables(NttTables& t, bool r4 = false)
{
    std::vector<double> fwd(kN), inv(kN);
    const uint64_t psi = fpf::PSI_2048, psi_inv = powmod_u64(psi, fpf::P_U64 - 2);
    for (uint32_t i = 0; i < (uint32_t)kN; i++) {
        fwd[i] = balanced(powmod_u64(psi, bitrev(i, 10)));
        inv[i] = balanced(powmod_u64(psi_inv, bitrev(i, 10)));
    }
    fill_tables(t, fwd, inv, r4);
}

// exact product mod p of two balanced residues held in doubles, balanced again
double prod_balanced(double a, double b)
{
    const __int128 p = (__int128)fpf::P_U64;
    __int128 v = ((__int128)(int64_t)a * (__int128)(int64_t)b) % p;
    if (v < 0) v += p;
    return balanced((uint64_t)v);
}

// Radix-4 form of a 512-point transform (ntt_wave512.h: q4): the product of a block's stage-a and first stage-b twiddle, u w
// (forward) and v w (inverse) -- in the spare slot 7 of tu_fwd / tu_inv for the wave-uniform block, in uwb_* / uwc_* per lam and per
// lane.  The second stage-b twiddle must be I times (forward) / -I times (inverse) the first: false if it is not.
bool fill_r4_products_512(Ntt512Tables& t)
{
    auto apart = [&](double w1, double w2, bool inverse) { return prod_balanced(w1, inverse ? -fpf::ROOT4 : fpf::ROOT4) == w2; };
    if (!apart(t.tu_fwd[1], t.tu_fwd[2], false) || !apart(t.tu_inv[1], t.tu_inv[2], true)) return false;
    t.tu_fwd[7] = prod_balanced(t.tu_fwd[0], t.tu_fwd[1]);
    t.tu_inv[7] = prod_balanced(t.tu_inv[0], t.tu_inv[1]);
    for (int lam = 0; lam < 8; lam++) {
        if (!apart(t.tb_fwd[8 + lam], t.tb_fwd[16 + lam], false) || !apart(t.tb_inv[8 + lam], t.tb_inv[16 + lam], true)) return false;
        t.uwb_fwd[lam] = prod_balanced(t.tb_fwd[lam], t.tb_fwd[8 + lam]);
        t.uwb_inv[lam] = prod_balanced(t.tb_inv[lam], t.tb_inv[8 + lam]);
    }
    for (int lane = 0; lane < 64; lane++) {
        if (!apart(t.tc_fwd[64 + lane], t.tc_fwd[128 + lane], false) || !apart(t.tc_inv[64 + lane], t.tc_inv[128 + lane], true)) return false;
        t.uwc_fwd[lane] = prod_balanced(t.tc_fwd[lane], t.tc_fwd[64 + lane]);
        t.uwc_inv[lane] = prod_balanced(t.tc_inv[lane], t.tc_inv[64 + lane]);
    }
    return true;
}

// Tables of one 512-point transform from accessors rf(idx) / ri(idx) = forward / inverse twiddle of
// group g at the stage with m groups, idx = m + g.
template <class RF, class RI>
void fill_tables_512(Ntt512Tables& t, RF rf, RI ri)
{
    memset(&t, 0, sizeof(t));
    for (int k = 0; k < 7; k++) {
        int lvl = 0;
        while ((2 << lvl) <= k + 1) lvl++;
        const int j = k + 1 - (1 << lvl);
        t.tu_fwd[k] = rf((1 << lvl) + j);
        t.tu_inv[k] = ri((1 << lvl) + j);
        for (int lam = 0; lam < 8; lam++) {
            const int idx = (8 << lvl) + (lam << lvl) + j;
            t.tb_fwd[k * 8 + lam] = rf(idx);
            t.tb_inv[k * 8 + lam] = ri(idx);
        }
        for (int lane = 0; lane < 64; lane++) {
            const int mu = 8 * (lane & 7) + (lane >> 3);
            const int idx = (64 << lvl) + (mu << lvl) + j;
            t.tc_fwd[k * 64 + lane] = rf(idx);
            t.tc_inv[k * 64 + lane] = ri(idx);
        }
    }
}

// t[0], t[1]: the two 512-point halves of the 1024-point transform, root_h[m + g] = root[2m + h m + g];
// t[2]: the stand-alone 512-point negacyclic transform (psi_1024 = psi_2048^2)
void build_tables_512(Ntt512Tables (&t)[3])
{
    std::vector<double> fwd(kN), inv(kN);
    const uint64_t psi = fpf::PSI_2048, psi_inv = powmod_u64(psi, fpf::P_U64 - 2);
    for (uint32_t i = 0; i < (uint32_t)kN; i++) {
        fwd[i] = balanced(powmod_u64(psi, bitrev(i, 10)));
        inv[i] = balanced(powmod_u64(psi_inv, bitrev(i, 10)));
    }
    auto top = [](int idx) { int m = 1; while (2 * m <= idx) m *= 2; return m; };
    for (int h = 0; h < 2; h++)
        fill_tables_512(t[h], [&](int idx) { const int m = top(idx); return fwd[2 * m + h * m + (idx - m)]; },
                        [&](int idx) { const int m = top(idx); return inv[2 * m + h * m + (idx - m)]; });
    const uint64_t psi2 = mulmod_u64(psi, psi), psi2_inv = powmod_u64(psi2, fpf::P_U64 - 2);
    fill_tables_512(t[2], [&](int idx) { return balanced(powmod_u64(psi2, bitrev((uint32_t)idx, 9))); },
                    [&](int idx) { return balanced(powmod_u64(psi2_inv, bitrev((uint32_t)idx, 9))); });
}

// /sys/bus/pci/devices/<domain:bus:dev.fn>/local_cpulist of a physical HIP device ("" when it cannot be read)
std::string device_local_cpulist(int phys)
{
    char pci[64];
    if (hipDeviceGetPCIBusId(pci, (int)sizeof pci, phys) != hipSuccess) { (void)hipGetLastError(); return ""; }
    std::string id(pci);
    for (auto& ch : id) ch = (char)tolower((unsigned char)ch);
    FILE* f = fopen(("/sys/bus/pci/devices/" + id + "/local_cpulist").c_str(), "r");
    if (!f) return "";
    char line[1024] = "";
    const bool ok = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    std::string out(ok ? line : "");
    while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();
    return out;
}

// Device allocations of the Initialize entry points go through here: "test_fail_alloc" n makes the (n+1)-th one fail like an exhausted
// device (then disarms itself), so that the error paths -- the old keys stay loaded and usable, nothing leaks -- can be exercised
long g_fail_alloc_countdown = -1;
hipError_t init_malloc(void** p, size_t bytes)
{
    if (g_fail_alloc_countdown >= 0 && g_fail_alloc_countdown-- == 0) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    return hipMalloc(p, bytes);
}

int check_device(int device)
{
    if (device < 0 || device >= g_gpu_num) return fail(-1, "device index out of range (SetGPUNum first)");
    return 0;
}
int use_device(int device)
{
    if (int rc = check_device(device)) return rc;
    HIP_TRY(hipSetDevice(phys_device(device)));
    return 0;
}

int ensure_ntt(int device)
{
    DeviceState& s = g_dev[device];
    if (s.ntt_ready) return 0;
    HIP_TRY(hipSetDevice(phys_device(device)));
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, phys_device(device)));
        s.cus = prop.multiProcessorCount;
    }
    // every allocation of this function is released again when a later one fails: a retry starts from nothing
    auto undo = [&]() {
        if (s.fault_host) { (void)hipHostFree(s.fault_host); s.fault_host = nullptr; s.fault = nullptr; }
        if (s.tables) { (void)hipFree(s.tables); s.tables = nullptr; }
        if (s.tables_r4) { (void)hipFree(s.tables_r4); s.tables_r4 = nullptr; }
        if (s.tables512) { (void)hipFree(s.tables512); s.tables512 = nullptr; }
    };
#define CUFHE_AMD_TRY_UNDO(expr)                                                               \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            undo();                                                                            \
            return fail(-2, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
        }                                                                                      \
    } while (0)
    CUFHE_AMD_TRY_UNDO(hipHostMalloc((void**)&s.fault_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *s.fault_host = 0;
    CUFHE_AMD_TRY_UNDO(hipHostGetDevicePointer((void**)&s.fault, s.fault_host, 0));
    NttTables host;
    build_tables(host);
    CUFHE_AMD_TRY_UNDO(hipMalloc((void**)&s.tables, sizeof(NttTables)));
    CUFHE_AMD_TRY_UNDO(hipMemcpy(s.tables, &host, sizeof(NttTables), hipMemcpyHostToDevice));
    build_tables(host, true);
    CUFHE_AMD_TRY_UNDO(hipMalloc((void**)&s.tables_r4, sizeof(NttTables)));
    CUFHE_AMD_TRY_UNDO(hipMemcpy(s.tables_r4, &host, sizeof(NttTables), hipMemcpyHostToDevice));
    static Ntt512Tables host512[3];
    build_tables_512(host512);
    for (int h = 0; h < 2; h++)
        if (!fill_r4_products_512(host512[h])) return fail(-2, "half-transform tables: the stage-b twiddles of a block are not I apart (radix-4 form)");
    CUFHE_AMD_TRY_UNDO(hipMalloc((void**)&s.tables512, sizeof(host512)));
    CUFHE_AMD_TRY_UNDO(hipMemcpy(s.tables512, host512, sizeof(host512), hipMemcpyHostToDevice));
#undef CUFHE_AMD_TRY_UNDO
    s.ntt_ready = true;
    return 0;
}

// A kernel that found its own result untrustworthy has set a bit in the device's fault word (today: the bounded LDS
// rendezvous of blind_rotate_ll2_kernel).  Called wherever the host observes completion (Synchronize, StreamQuery,
// StreamSynchronize, the scheduler's event polls): status -5 with text, where the reference would have printed
// "CuCheckError() failed" and exited (include/details/error_gpu.cuh:40-60; the C++ shim aborts on any negative status).
// Sticky, like a CUDA context error, until CleanUp().
int device_fault(int device)
{
    DeviceState& s = g_dev[device];
    if (!s.fault_host) return 0;
    const uint32_t bits = __atomic_load_n(s.fault_host, __ATOMIC_ACQUIRE);
    if (bits == 0) return 0;
    std::string what;
    if (bits & kFaultLl2SyncTimeout) what += " blind_rotate_ll2_kernel: the inverse waves' LDS rendezvous timed out;";
    if (bits & ~kFaultLl2SyncTimeout) what += " unknown bits;";
    char hex[16];
    snprintf(hex, sizeof hex, "0x%x", bits);
    return fail(-5, std::string("device ") + std::to_string(device) + " fault word " + hex + ":" + what +
                        " ciphertexts produced since Initialize() are not trustworthy -- CleanUp() and Initialize() again");
}


struct Scratch {   // bump allocator over the stream's workspace
    char* cur;
    char* end;
    hipStream_t st;
    int alloc(void** p, size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        if (cur + bytes > end) return fail(-4, "internal: workspace under-sized");
        *p = cur;
        cur += bytes;
        return 0;
    }
};

int open_scratch(DeviceState& s, hipStream_t st, size_t bytes, Scratch* sc)
{
    std::lock_guard<std::mutex> lk(s.staging_mu);
    Workspace& w = s.workspaces[st];
    if (w.bytes < bytes) {
        if (w.base) {
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipFree(w.base));
            w.base = nullptr; w.bytes = 0;
        }
        const size_t want = bytes + bytes / 4 + (1 << 20);
        HIP_TRY(hipMalloc((void**)&w.base, want));
        w.bytes = want;
    }
    sc->cur = w.base; sc->end = w.base + w.bytes; sc->st = st;
    return 0;
}

int acquire_staging(DeviceState& s, size_t bytes, PinnedBlock** out)
{
    std::lock_guard<std::mutex> lk(s.staging_mu);
    for (auto& b : s.staging) {
        if (b.busy && b.recorded && !b.held && hipEventQuery(b.done) == hipSuccess) b.busy = false;
        if (!b.busy && b.bytes >= bytes) { b.busy = true; b.recorded = false; b.held = false; *out = &b; return 0; }
    }
    PinnedBlock nb;
    nb.bytes = bytes < 65536 ? 65536 : bytes;
    HIP_TRY(hipHostMalloc(&nb.host, nb.bytes, hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&nb.done, hipEventDisableTiming));
    nb.busy = true;
    s.staging.push_back(nb);
    *out = &s.staging.back();
    return 0;
}
// the owner's last device access to the block is on `st`: from its completion on the block may be handed out again
int staging_done_after(DeviceState& s, PinnedBlock* blk, hipStream_t st)
{
    HIP_TRY(hipEventRecord(blk->done, st));
    std::lock_guard<std::mutex> lk(s.staging_mu);
    blk->recorded = true;
    return 0;
}
void staging_hold(DeviceState& s, PinnedBlock* blk, bool held)
{
    std::lock_guard<std::mutex> lk(s.staging_mu);
    blk->held = held;
}
// The owner of a block between acquire_staging and staging_done_after.  A failing HIP call in between returns early: without this
// guard the block would stay busy and unrecorded for ever (a pinned-memory leak, and every later call would allocate a new block).
// On that path the device may still be reading or writing the block, so it is handed back only after the device has drained.
struct StagingOwner {
    DeviceState& s;
    PinnedBlock* b;
    bool ok = false;
    void recorded() { ok = true; }
    ~StagingOwner()
    {
        if (!ok) (void)hipDeviceSynchronize();
        std::lock_guard<std::mutex> lk(s.staging_mu);
        b->held = false;
        if (!ok) { b->busy = false; b->recorded = false; }
    }
};

template <class Desc>
int upload_descs(DeviceState& s, Scratch& sc, const std::vector<Desc>& h, Desc** d)
{
    *d = nullptr;
    if (h.empty()) return 0;
    const size_t bytes = h.size() * sizeof(Desc);
    if (int rc = sc.alloc((void**)d, bytes)) return rc;
    PinnedBlock* blk = nullptr;
    if (int rc = acquire_staging(s, bytes, &blk)) return rc;
    StagingOwner owner{s, blk};
    memcpy(blk->host, h.data(), bytes);
    HIP_TRY(hipMemcpyAsync(*d, blk->host, bytes, hipMemcpyHostToDevice, sc.st));
    if (int rc = staging_done_after(s, blk, sc.st)) return rc;
    owner.recorded();
    return 0;
}

// HIP events around a launch sequence on its own stream (cufhe_amd_profile_enable): begin before, end after
int prof_begin(DeviceState& s, hipStream_t st, EventPair& ev)
{
    if (!s.profiling) return 0;
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    HIP_TRY(hipEventRecord(ev.a, st));
    return 0;
}
int prof_end(DeviceState& s, hipStream_t st, EventPair& ev, size_t units, bool keyswitch)
{
    if (!s.profiling || !ev.a) return 0;
    HIP_TRY(hipEventRecord(ev.b, st));
    ev.units = units;
    std::lock_guard<std::mutex> lk(s.staging_mu);
    (keyswitch ? s.ks_events : s.br_events).push_back(ev);
    return 0;
}

int launch_blind_rotate(DeviceState& s, hipStream_t st, const LinDesc* d, size_t count, int steps, uint32_t* acc_dump)
{
    if (count == 0) return 0;
    EventPair ev{};
    if (s.profiling) {
        HIP_TRY(hipEventCreate(&ev.a));
        HIP_TRY(hipEventCreate(&ev.b));
        HIP_TRY(hipEventRecord(ev.a, st));
    }
    if (!s.br_lds_opt_in) {      // > 64 KiB of dynamic LDS needs an opt-in, per device
        HIP_TRY(hipFuncSetAttribute((const void*)blind_rotate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kBrLdsBytes));
        HIP_TRY(hipFuncSetAttribute((const void*)blind_rotate_ll_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLlLdsBytes));
        HIP_TRY(hipFuncSetAttribute((const void*)blind_rotate_ll2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLl2LdsBytes));
        s.br_lds_opt_in = true;
    }
    // One round of the batch kernel's grid is a workgroup of 8 rotations per CU and takes ~19 ms however few of its
    // wave slots are used, so a launch of a few rotations past a whole round used to cost two rounds.  Launches are cut into
    // whole rounds plus a tail, and the tail takes the cheapest of: the low-latency kernel (a CU per
    // rotation, 3.3 ms per started round of one rotation per CU), its paired form (two per CU, 5.3 - 5.8 ms), the batch
    // kernel with one rotation per SIMD (~12 ms per round of four per CU), a full round.  All variants compute identical
    // words.  Every rule below is in units of the device's CU count (MI355X: 256; the measured milliseconds are that chip's).
    const size_t cu = (size_t)std::max(1, cus_of(s));
    const size_t kRound = cu * kBrWavesPerBlock;
    auto launch_batch = [&](const LinDesc* dd, size_t n, int active, uint32_t* dump) {
        const unsigned blocks = (unsigned)((n + active - 1) / active);
        hipLaunchKernelGGL(blind_rotate_kernel, dim3(blocks), dim3(kBrThreads), kBrLdsBytes, st, dd, (int)n,
                           s.bk_ntt, s.tables_r4, steps, dump, active);
    };
    // Measured on MI355X, 256 CUs (tools/latency_sweep.py, tools/ll_times.py; profiles/r02_latency_sweep.txt, r05_ll_ab.txt), ms per
    // launch of n rotations, key switch included:
    //   low-latency kernel  2.9 (n <= 64), 3.3 / 6.7 / 10.0 / 13.3 / 16.6 per started round of one rotation per CU
    //   its paired form     5.0 - 5.3 per started round of two per CU (10.4 for four, 17.1 for six, 22.9 for eight)
    //   one rotation per SIMD 12.7 (n <= 4 per CU)          two per SIMD 20.7 (n <= 8 per CU)
    // so: low-latency up to one rotation per CU, rounds of two per CU on the paired kernel (+ a last round of up to one per CU on
    // the single one) up to six per CU, a full round of the batch kernel above.  ("ll2_threshold" 0 gives the rules without the
    // paired kernel: low-latency up to three per CU, one-per-SIMD up to four, both up to five.)
    const bool auto_ll = g_ll_threshold < 0, auto_half = g_half_threshold < 0;
    auto launch_ll = [&](const LinDesc* dd, size_t n, uint32_t* dump) {
        // smallest batches: one 16-wave workgroup per rotation, transforms split in halves (kernels_ll.hip.h)
        hipLaunchKernelGGL(blind_rotate_ll_kernel, dim3((unsigned)n), dim3(kLlThreads), kLlLdsBytes, st, dd, (int)n,
                           s.bk_ntt, s.tables512, steps, dump);
    };
    auto launch_ll2 = [&](const LinDesc* dd, size_t n, uint32_t* dump) {
        // two rotations per workgroup: the row phase of one beside the inverse transforms of the other (kernels_ll.hip.h)
        hipLaunchKernelGGL(blind_rotate_ll2_kernel, dim3((unsigned)((n + 1) / 2)), dim3(kLlThreads), kLl2LdsBytes, st, dd, (int)n,
                           s.bk_ntt, s.tables512, steps, dump, s.fault);
    };
    auto launch_small = [&](const LinDesc* dd, size_t n, uint32_t* dump) {
        if (g_ll2_threshold > 0 && (long)n <= g_ll2_threshold) {
            launch_ll2(dd, n, dump);
            return;
        }
        if (g_ll2_threshold < 0 && auto_ll && auto_half && n > cu && n <= 6 * cu) {
            // rounds of two rotations per CU on the paired kernel and a last started round of up to one per CU on the single one
            const size_t rem = n % (2 * cu), paired = (rem == 0 || rem > cu) ? n : n - rem;
            launch_ll2(dd, paired, dump);
            if (paired < n) launch_ll(dd + paired, n - paired, dump ? dump + paired * 2 * kN : nullptr);
            return;
        }
        if (auto_ll && auto_half && n > 4 * cu && n <= 5 * cu) {
            // (without the paired kernel) four per CU at one rotation per SIMD (12.0 ms) and the rest on the low-latency kernel (3.2):
            // 15.5 ms against 16.6 for five rounds of the low-latency kernel and 20 for a full round
            launch_batch(dd, 4 * cu, kBrWavesPerBlock / 2, dump);
            launch_ll(dd + 4 * cu, n - 4 * cu, dump ? dump + 4 * cu * 2 * kN : nullptr);
            return;
        }
        const bool use_ll = auto_ll ? n <= 3 * cu : (long)n <= g_ll_threshold;
        const bool use_half = auto_half ? n <= 4 * cu : (long)n <= g_half_threshold;
        if (use_ll) {
            launch_ll(dd, n, dump);
        } else if (use_half) {
            launch_batch(dd, n, kBrWavesPerBlock / 2, dump);
        } else {
            launch_batch(dd, n, kBrWavesPerBlock, dump);
        }
    };
    if (g_br_shape > 0) {
        // a caller that places launches itself (the two-lane scheduler, tools/two_lane_probe.py): the whole launch on one kernel
        if (g_br_shape == 1) launch_batch(d, count, kBrWavesPerBlock, acc_dump);
        else if (g_br_shape == 2) launch_ll2(d, count, acc_dump);
        else launch_ll(d, count, acc_dump);
        HIP_TRY(hipGetLastError());
        if (s.profiling) {
            HIP_TRY(hipEventRecord(ev.b, st));
            ev.units = count;
            std::lock_guard<std::mutex> lk(s.staging_mu);
            s.br_events.push_back(ev);
        }
        return 0;
    }
    const size_t tail = count % kRound;
    const long tail_max = std::max(auto_half ? (long)(4 * cu) : g_half_threshold, auto_ll ? (long)((g_ll2_threshold < 0 ? 6 : 5) * cu) : g_ll_threshold);
    if (g_tail_split && count > kRound && tail != 0 && (long)tail <= tail_max) {
        const size_t full = count - tail;
        launch_batch(d, full, kBrWavesPerBlock, acc_dump);
        launch_small(d + full, tail, acc_dump ? acc_dump + full * 2 * kN : nullptr);
    } else {
        launch_small(d, count, acc_dump);
    }
    HIP_TRY(hipGetLastError());
    if (s.profiling) {
        HIP_TRY(hipEventRecord(ev.b, st));
        ev.units = count;
        std::lock_guard<std::mutex> lk(s.staging_mu);
        s.br_events.push_back(ev);
    }
    return 0;
}
// keyswitch_kernel<S> over `ksk_padded` ([kn][t][2][row_pad] u32); *opted_in: the instantiation's dynamic-LDS opt-in on this device
template <class S>
int launch_keyswitch_shared(DeviceState& s, hipStream_t st, const typename S::Desc* d, size_t count, const uint32_t* ksk_padded, bool* opted_in)
{
    if (!*opted_in) {
        HIP_TRY(hipFuncSetAttribute((const void*)keyswitch_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, KsDims<S>::lds_bytes));
        *opted_in = true;
    }
    int per_wg, slices;
    ks_auto_shape(count, cus_of(s), S::kn, KsDims<S>::min_slices, &per_wg, &slices);
    if (slices > 1) hipLaunchKernelGGL(keyswitch_zero_kernel<S>, dim3((unsigned)count), dim3(256), 0, st, d, (int)count);
    const unsigned ks_blocks = (unsigned)((count + per_wg - 1) / per_wg) * (unsigned)slices;
    hipLaunchKernelGGL(keyswitch_kernel<S>, dim3(ks_blocks), dim3(kKsThreads), KsDims<S>::lds_bytes, st, d, (int)count, ksk_padded, per_wg, slices);
    return 0;
}
int launch_keyswitch(DeviceState& s, hipStream_t st, const LinDesc* d, size_t count)
{
    if (count == 0) return 0;
    EventPair ev{};
    if (s.profiling) {
        HIP_TRY(hipEventCreate(&ev.a));
        HIP_TRY(hipEventCreate(&ev.b));
        HIP_TRY(hipEventRecord(ev.a, st));
    }
    const long split_max = g_ks_split_threshold < 0 ? ks_auto_split(cus_of(s)) : g_ks_split_threshold;
    const long wg_max = g_ks_wg_threshold < 0 ? ks_auto_wg(cus_of(s)) : g_ks_wg_threshold;
    if ((long)count <= split_max) {
        hipLaunchKernelGGL(keyswitch_split_zero_kernel, dim3((unsigned)count), dim3(256), 0, st, d, (int)count);
        hipLaunchKernelGGL(keyswitch_split_kernel, dim3((unsigned)count * kKsSplit), dim3(kKsThreads), 0, st, d, (int)count, s.ksk);
    } else if ((long)count <= wg_max) {
        hipLaunchKernelGGL(keyswitch_wg_kernel, dim3((unsigned)count), dim3(kKsThreads), 0, st, d, (int)count, s.ksk);
    } else {
        if (int rc = launch_keyswitch_shared<KsShapeDefault>(s, st, d, count, s.ksk, &s.ks_lds_opt_in)) return rc;
    }
    HIP_TRY(hipGetLastError());
    if (s.profiling) {
        HIP_TRY(hipEventRecord(ev.b, st));
        ev.units = count;
        std::lock_guard<std::mutex> lk(s.staging_mu);
        s.ks_events.push_back(ev);
    }
    return 0;
}
int launch_lincomb(hipStream_t st, const LinDesc* d, size_t count, int words)
{
    if (count == 0) return 0;
    const unsigned blocks = (unsigned)(count < 4096 ? count : 4096);
    hipLaunchKernelGGL(lincomb_kernel, dim3(blocks), dim3(256), 0, st, d, (int)count, words);
    HIP_TRY(hipGetLastError());
    return 0;
}

// (ca, cb, offset/mu) of the ten two-input gates, src/bootstrap_gpu.cu:424-512,591-679
const int kGateTab[10][3] = {
    {-1, -1, 1}, {-1, -1, -1}, {-2, -2, -2}, {1, 1, -1}, {1, 1, 1},
    {2, 2, 2},   {-1, 1, -1},  {1, -1, -1},  {-1, 1, 1}, {1, -1, 1},
};

using sched::GateRef;

template <class GetGate>
int run_gates_lvl2(int device, void* stream, size_t count, GetGate get);   // lvl2.inc.h
template <class GetGate>
int run_gates_ps(int set, int device, void* stream, int level, size_t count, GetGate get);   // paramsets.inc.h
int ps_ctxt_words(int set, int level);
void lvl2_release_host_key();          // lvl2.inc.h
// a TRGSW holder's device slot (ciphertext handle of level 3) fits the NTT-domain TRGSW of every compiled set, key limbs included
template <class PS> constexpr int kTrgswNttWordsOf = (int)(2 * PsDims<PS>::bk_ntt_step_doubles);
constexpr int kMaxTrgswNttWords = std::max({(int)(2 * kBkStepDoubles), kTrgswNttWordsOf<PsDefault>, kTrgswNttWordsOf<PsK2N512>, kTrgswNttWordsOf<PsCggi16>});
int run_trlwe_ops_ps(int set, int device, void* stream, const GateRef* g, size_t n);         // paramsets.inc.h
int ps_trgsw_to_ntt_host(int set, int device, void* stream, const uint32_t* trgsw_host, double* trgsw_ntt_host);
// >= 0: the per-gate API (both ciphertext levels, both gate orders) runs on this compiled parameter set -- the reference's build-time
// choice (CMakeLists.txt:8-24) serves every entry point the same way; ciphertexts then have the set's sizes (cufhe_amd_ctxt_words)
long g_param_set = -1;

template <class GetGate>
int run_gates(int device, void* stream, int level, size_t count, GetGate get)
{
    if (level == 0 && g_lvl0_ring == 2048) return run_gates_lvl2(device, stream, count, get);
    if ((level == 0 || level == 1) && g_param_set >= 0) return run_gates_ps((int)g_param_set, device, stream, level, count, get);
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    if (!s.keys_ready) return fail(-3, "Initialize(ek) has not been called for this device");
    if (level != 0 && level != 1) return fail(-1, "level must be 0 or 1");
    if (count == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t negmu = 0u - kMu;

    std::vector<LinDesc> rot, ks, lin;
    rot.reserve(count * 2); ks.reserve(count * 2); lin.reserve(count);
    // first pass: count temporaries
    size_t nrot = 0;
    for (size_t g = 0; g < count; g++) {
        const int op = get(g).op;
        if (op < 0 || op >= CUFHE_AMD_NUM_OPS) return fail(-1, "unknown gate op");
        if (op == CUFHE_AMD_MUX || op == CUFHE_AMD_NMUX) nrot += 2;
        else if (op < CUFHE_AMD_MUX) nrot += 1;
    }
    Scratch sc;
    {
        const size_t need = nrot * (kLvl1Words + kLvl0Words) * sizeof(uint32_t) +
                            (count * 5 + 8) * sizeof(LinDesc) + 4096;
        if (int rc = open_scratch(s, st, need, &sc)) return rc;
    }
    uint32_t *tmp1 = nullptr, *tmp0 = nullptr;   // lvl1 / lvl0 temporaries, one per rotation
    if (nrot) {
        if (int rc = sc.alloc((void**)&tmp1, nrot * kLvl1Words * sizeof(uint32_t))) return rc;
        if (level == 1)
            if (int rc = sc.alloc((void**)&tmp0, nrot * kLvl0Words * sizeof(uint32_t))) return rc;
    }
    size_t ir = 0;
    for (size_t g = 0; g < count; g++) {
        const GateRef gr = get(g);
        if (!gr.out || !gr.in0) return fail(-1, "null ciphertext pointer");
        if (gr.op == CUFHE_AMD_NOT || gr.op == CUFHE_AMD_COPY) {
            lin.push_back({gr.in0, gr.in0, gr.out, gr.op == CUFHE_AMD_NOT ? -1 : 1, 0, 0u, 0u});
            continue;
        }
        if (!gr.in1) return fail(-1, "gate needs a second operand");
        if (gr.op == CUFHE_AMD_MUX || gr.op == CUFHE_AMD_NMUX) {
            if (!gr.in2) return fail(-1, "mux needs a third operand");
            uint32_t* t1a = tmp1 + (ir + 0) * kLvl1Words;
            uint32_t* t1b = tmp1 + (ir + 1) * kLvl1Words;
            const bool neg = gr.op == CUFHE_AMD_NMUX;
            if (level == 0) {   // src/bootstrap_gpu.cu:515-588
                rot.push_back({gr.in0, gr.in1, t1a, 1, 1, negmu, 0u});
                rot.push_back({gr.in0, gr.in2, t1b, -1, 1, negmu, 0u});
                ks.push_back({t1a, t1b, gr.out, neg ? -1 : 1, neg ? -1 : 1, neg ? negmu : kMu, 0u});
            } else {            // src/bootstrap_gpu.cu:706-780
                uint32_t* t0a = tmp0 + (ir + 0) * kLvl0Words;
                uint32_t* t0b = tmp0 + (ir + 1) * kLvl0Words;
                ks.push_back({gr.in0, gr.in1, t0a, 1, 1, negmu, 0u});
                ks.push_back({gr.in0, gr.in2, t0b, -1, 1, negmu, 0u});
                rot.push_back({t0a, t0a, t1a, 1, 0, 0u, 0u});
                rot.push_back({t0b, t0b, t1b, 1, 0, 0u, 0u});
                lin.push_back({t1a, t1b, gr.out, neg ? -1 : 1, neg ? -1 : 1, neg ? negmu : kMu, 0u});
            }
            ir += 2;
            continue;
        }
        const int ca = kGateTab[gr.op][0], cb = kGateTab[gr.op][1];
        const uint32_t off = (uint32_t)kGateTab[gr.op][2] * kMu;
        if (level == 0) {       // __HomGate__ br->iks, src/bootstrap_gpu.cu:402-421
            uint32_t* t1 = tmp1 + ir * kLvl1Words;
            rot.push_back({gr.in0, gr.in1, t1, ca, cb, off, 0u});
            ks.push_back({t1, t1, gr.out, 1, 0, 0u, 0u});
        } else {                // __HomGate__ iks->br, src/bootstrap_gpu.cu:383-400
            uint32_t* t0 = tmp0 + ir * kLvl0Words;
            ks.push_back({gr.in0, gr.in1, t0, ca, cb, off, 0u});
            rot.push_back({t0, t0, gr.out, 1, 0, 0u, 0u});
        }
        ir += 1;
    }
    // Mux/NMux at level 1 write their rotations to temporaries, two-input gates at level 1
    // write straight to `out`; a lincomb that reads tmp1 must run after the rotations.
    LinDesc *drot, *dks, *dlin;
    if (int rc = upload_descs(s, sc, rot, &drot)) return rc;
    if (int rc = upload_descs(s, sc, ks, &dks)) return rc;
    if (int rc = upload_descs(s, sc, lin, &dlin)) return rc;
    const int words = level ? kLvl1Words : kLvl0Words;
    if (level == 0) {
        if (int rc = launch_blind_rotate(s, st, drot, rot.size(), kLvl0N, nullptr)) return rc;
        if (int rc = launch_keyswitch(s, st, dks, ks.size())) return rc;
    } else {
        if (int rc = launch_keyswitch(s, st, dks, ks.size())) return rc;
        if (int rc = launch_blind_rotate(s, st, drot, rot.size(), kLvl0N, nullptr)) return rc;
    }
    if (int rc = launch_lincomb(st, dlin, lin.size(), words)) return rc;
    return 0;
}

// TRLWE-level operations recorded through the scheduler (include/cufhe_gpu.cuh:124-146,209-216,282-285;
// src/cufhe_gates_gpu.cu:86-146): any mix of
//   CUFHE_AMD_TL_BOOTSTRAP  lvl0 TLWE -> TRLWE     __BlindRotateGlobal__, src/bootstrap_gpu.cu:317-323
//   CUFHE_AMD_TL_REFRESH    TRLWE -> TRLWE         __SEIandBootstrap2TRLWE__, :325-364
//   CUFHE_AMD_TL_SEIKS      TRLWE -> lvl0 TLWE     __SEIandKS__, src/keyswitch_gpu.cu:26-40
// as ONE launch sequence: sample extracts, one key-switch launch, one blind-rotate launch, scatter.
int run_trlwe_ops(int device, void* stream, const GateRef* g, size_t n)
{
    if (g_param_set >= 0) return run_trlwe_ops_ps((int)g_param_set, device, stream, g, n);
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    size_t n_se = 0, n_rot = 0, n_t0 = 0, n_cmux = 0;
    for (size_t i = 0; i < n; i++) {
        if (!g[i].out || !g[i].in0) return fail(-1, "null operand");
        switch (g[i].op) {
            case CUFHE_AMD_TL_BOOTSTRAP: n_rot++; break;
            case CUFHE_AMD_TL_REFRESH: n_se++; n_rot++; n_t0++; break;
            case CUFHE_AMD_TL_SEIKS: n_se++; break;
            case CUFHE_AMD_TL_CMUX:
                if (!g[i].in1 || !g[i].in2) return fail(-1, "CMUXNTT: null operand");
                n_cmux++;
                break;
            default: return fail(-1, "unknown TRLWE-level op");
        }
    }
    if (n_cmux < n && !s.keys_ready) return fail(-3, "Initialize(ek) has not been called for this device");
    if (n_cmux && !s.ntt_ready) return fail(-3, "Initialize() has not been called for this device");
    Scratch sc;
    const size_t need = n_se * kLvl1Words * 4 + n_t0 * kLvl0Words * 4 + n_rot * 2 * kN * 4 + (3 * n + 8) * sizeof(LinDesc) +
                        n_cmux * sizeof(CmuxDesc) + 16384;
    if (int rc = open_scratch(s, st, need, &sc)) return rc;
    if (n_cmux) {      // the CMUXNTT calls of this level: independent of the other operations of the level (the scheduler's contract)
        std::vector<CmuxDesc> cm;
        cm.reserve(n_cmux);
        for (size_t i = 0; i < n; i++)
            if (g[i].op == CUFHE_AMD_TL_CMUX) cm.push_back({g[i].in0, g[i].in1, g[i].out, (const double*)g[i].in2});
        CmuxDesc* dcm;
        if (int rc = upload_descs(s, sc, cm, &dcm)) return rc;
        const unsigned blocks = (unsigned)((cm.size() + kNttWavesPerBlock - 1) / kNttWavesPerBlock);
        hipLaunchKernelGGL(cmux_desc_kernel, dim3(blocks), dim3(kNttThreads), kNttLdsBytes, st, dcm, (int)cm.size(), s.tables);
        HIP_TRY(hipGetLastError());
        if (n_cmux == n) return 0;
    }
    uint32_t *t1 = nullptr, *t0 = nullptr, *dump = nullptr;
    if (n_se) if (int rc = sc.alloc((void**)&t1, n_se * kLvl1Words * 4)) return rc;
    if (n_t0) if (int rc = sc.alloc((void**)&t0, n_t0 * kLvl0Words * 4)) return rc;
    if (n_rot) if (int rc = sc.alloc((void**)&dump, n_rot * 2 * kN * 4)) return rc;
    std::vector<LinDesc> se, ks, rot, scat;
    size_t i_se = 0, i_t0 = 0, i_rot = 0;
    for (size_t i = 0; i < n; i++) {
        if (g[i].op == CUFHE_AMD_TL_CMUX) continue;
        if (g[i].op == CUFHE_AMD_TL_BOOTSTRAP) {
            rot.push_back({g[i].in0, g[i].in0, nullptr, 1, 0, 0u, 0u});
        } else {
            uint32_t* a = t1 + i_se++ * kLvl1Words;
            se.push_back({g[i].in0, g[i].in0, a, 1, 0, 0u, 0u});
            if (g[i].op == CUFHE_AMD_TL_SEIKS) {
                ks.push_back({a, a, g[i].out, 1, 0, 0u, 0u});
                continue;
            }
            uint32_t* b = t0 + i_t0++ * kLvl0Words;
            ks.push_back({a, a, b, 1, 0, 0u, 0u});
            rot.push_back({b, b, nullptr, 1, 0, 0u, 0u});
        }
        uint32_t* d = dump + i_rot++ * 2 * kN;
        scat.push_back({d, d, g[i].out, 1, 0, 0u, 0u});
    }
    LinDesc *dse, *dks, *drot, *dscat;
    if (int rc = upload_descs(s, sc, se, &dse)) return rc;
    if (int rc = upload_descs(s, sc, ks, &dks)) return rc;
    if (int rc = upload_descs(s, sc, rot, &drot)) return rc;
    if (int rc = upload_descs(s, sc, scat, &dscat)) return rc;
    if (!se.empty()) {
        hipLaunchKernelGGL(sample_extract_desc_kernel, dim3((unsigned)(se.size() < 2048 ? se.size() : 2048)), dim3(256), 0, st, dse, (int)se.size());
        HIP_TRY(hipGetLastError());
    }
    if (int rc = launch_keyswitch(s, st, dks, ks.size())) return rc;
    if (int rc = launch_blind_rotate(s, st, drot, rot.size(), kLvl0N, dump)) return rc;
    return launch_lincomb(st, dscat, scat.size(), 2 * kN);
}

}  // namespace

#include "sched_hip.inc.h"
#include "lvl2.inc.h"
#include "paramsets.inc.h"

extern "C" {

const char* cufhe_amd_last_error(void) { return g_err.c_str(); }

int cufhe_amd_get_params(cufhe_amd_params* p)
{
    if (!p) return fail(-1, "null");
    p->n = kLvl0N; p->N = kN; p->nbit = kNbit; p->k = 1; p->l = kL; p->Bgbit = kBgbit;
    p->t = kKsT; p->basebit = kKsBasebit; p->mu = kMu;
    p->lvl0_words = kLvl0Words; p->lvl1_words = kLvl1Words;
    p->bk_words = (uint64_t)kLvl0N * kBkStepDoubles;
    p->ksk_words = (uint64_t)kN * kKsT * kKsNumBase * kKsRowWords;
    p->bk_ntt_bytes = (uint64_t)kLvl0N * kBkStepDoubles * sizeof(double);
    return 0;
}

int cufhe_amd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int cufhe_amd_device_identity(int device, char* buf, size_t len)
{
    if (int rc = check_device(device)) return rc;
    if (!buf || len < 2) return fail(-1, "null / short buffer");
    const int phys = phys_device(device);
    char pci[64] = "?";
    HIP_TRY(hipDeviceGetPCIBusId(pci, (int)sizeof pci, phys));
    hipUUID uuid;
    memset(&uuid, 0, sizeof uuid);
    char hex[2 * sizeof(uuid.bytes) + 1] = "";
    if (hipDeviceGetUuid(&uuid, phys) == hipSuccess)
        for (size_t i = 0; i < sizeof(uuid.bytes); i++) snprintf(hex + 2 * i, 3, "%02x", (unsigned char)uuid.bytes[i]);
    (void)hipGetLastError();
    // CPUs close to the device (what the launch worker of this device is pinned to, "sched_affinity")
    std::string cpus = device_local_cpulist(phys);
    snprintf(buf, len, "pci=%s uuid=%s hip_device=%d local_cpus=%s", pci, hex[0] ? hex : "?", phys, cpus.empty() ? "?" : cpus.c_str());
    return 0;
}

int cufhe_amd_device_mem_info(int device, uint64_t* free_bytes, uint64_t* total_bytes)
{
    if (int rc = use_device(device)) return rc;
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

int cufhe_amd_device_cus(int device)
{
    if (int rc = check_device(device)) return rc;
    if (g_cus_override > 0) return (int)g_cus_override;
    int n = 0;
    HIP_TRY(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, phys_device(device)));
    return n;
}

int cufhe_amd_set_gpu_num(int gpu_num)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (gpu_num < 1) return fail(-1, "gpu_num must be >= 1");
    if (gpu_num > kMaxLogicalDevices) return fail(-1, "gpu_num exceeds the 64 logical devices this build is sized for");
    for (auto& d : g_dev)
        if (d.ntt_ready || d.keys_ready || d.tables2) return fail(-1, "SetGPUNum after Initialize: call CleanUp first");
    int have = cufhe_amd_device_count();
    g_phys_count = have;
    if (have < 1) return fail(-1, "no GPU visible");
    if (!g_share_devices && gpu_num + g_device_base > have) return fail(-1, "gpu_num (plus device_base) exceeds the visible device count");
    if (gpu_num != g_gpu_num) {
        std::lock_guard<std::mutex> lk2(g_sched_mu);
        sched_retire_generation();
    }
    g_gpu_num = gpu_num;
    g_dev.clear();
    g_dev.resize(gpu_num);
    return 0;
}
int cufhe_amd_get_gpu_num(void) { return g_gpu_num; }

int cufhe_amd_initialize_ntt(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < g_gpu_num; i++)
        if (int rc = ensure_ntt(i)) return rc;
    return 0;
}

int cufhe_amd_initialize(const uint32_t* bk, size_t bk_words, const uint32_t* ksk, size_t ksk_words)
{
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t want_bk = (size_t)kLvl0N * kBkStepDoubles;
    const size_t want_ksk = (size_t)kN * kKsT * kKsNumBase * kKsRowWords;
    if (!bk || !ksk) return fail(-1, "null key pointer");
    if (bk_words != want_bk) return fail(-1, "bootstrapping key has the wrong size for this parameter set");
    if (ksk_words != want_ksk) return fail(-1, "key-switching key has the wrong size for this parameter set");
    // Build first, swap last: the new keys of EVERY device are allocated and converted beside whatever is loaded; only when all of that
    // has succeeded do they replace the old ones.  A failure on the way (a full device: 104 MB per replica) frees what this call
    // allocated and leaves every device with the keys -- and the results -- it had.
    struct Built { double* bk_ntt = nullptr; uint32_t* ksk = nullptr; uint32_t* d_bk = nullptr; };
    std::vector<Built> built((size_t)g_gpu_num);
    struct Undo {
        std::vector<Built>& b; bool armed = true;
        ~Undo()
        {
            for (size_t i = 0; i < b.size(); i++) {
                if (!b[i].bk_ntt && !b[i].ksk && !b[i].d_bk) continue;
                (void)hipSetDevice(phys_device((int)i));
                (void)hipFree(b[i].d_bk);
                if (armed) { (void)hipFree(b[i].bk_ntt); (void)hipFree(b[i].ksk); }
            }
        }
    } undo{built};
    for (int i = 0; i < g_gpu_num; i++) {
        if (int rc = ensure_ntt(i)) return rc;
        DeviceState& s = g_dev[i];
        Built& b = built[(size_t)i];
        HIP_TRY(hipSetDevice(phys_device(i)));
        HIP_TRY(init_malloc((void**)&b.bk_ntt, want_bk * sizeof(double)));
        const size_t ksk_rows = want_ksk / kKsRowWords;
        HIP_TRY(init_malloc((void**)&b.ksk, ksk_rows * kKsRowPad * sizeof(uint32_t)));
        HIP_TRY(init_malloc((void**)&b.d_bk, want_bk * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(b.d_bk, bk, want_bk * sizeof(uint32_t), hipMemcpyHostToDevice));
        // KeySwitchingKeyToDevice (src/keyswitch_gpu.cu:6-16), rows padded 631 -> 640 words
        HIP_TRY(hipMemset(b.ksk, 0, ksk_rows * kKsRowPad * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy2D(b.ksk, kKsRowPad * sizeof(uint32_t), ksk, kKsRowWords * sizeof(uint32_t),
                            kKsRowWords * sizeof(uint32_t), ksk_rows, hipMemcpyHostToDevice));
        const size_t polys = want_bk / kN;
        const unsigned blocks = (unsigned)((polys + kNttWavesPerBlock - 1) / kNttWavesPerBlock);
        hipLaunchKernelGGL(bk_to_ntt_kernel, dim3(blocks), dim3(kNttThreads), kNttLdsBytes, 0, b.bk_ntt, b.d_bk,
                           polys, s.tables, n_inverse_balanced());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());        // also: nothing on this device still reads the keys that are about to go
    }
    for (int i = 0; i < g_gpu_num; i++) {
        DeviceState& s = g_dev[i];
        (void)hipSetDevice(phys_device(i));
        if (s.keys_ready) { (void)hipFree(s.bk_ntt); (void)hipFree(s.ksk); }
        s.bk_ntt = built[(size_t)i].bk_ntt;
        s.ksk = built[(size_t)i].ksk;
        s.keys_ready = true;
    }
    undo.armed = false;        // the guard still frees the torus-domain staging copies
    return 0;
}

int cufhe_amd_cleanup(void)
{
    {
        std::lock_guard<std::mutex> lk2(g_sched_mu);
        sched_quiesce();
    }
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < g_gpu_num; i++) {
        DeviceState& s = g_dev[i];
        if (!s.ntt_ready && !s.keys_ready && !s.tables2) continue;
        HIP_TRY(hipSetDevice(phys_device(i)));
        HIP_TRY(hipDeviceSynchronize());
        for (auto* v : {&s.br_events, &s.ks_events}) {
            for (auto& e : *v) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
            v->clear();
        }
        if (s.keys_ready) { HIP_TRY(hipFree(s.bk_ntt)); HIP_TRY(hipFree(s.ksk)); }
        ps_release(i);
        if (s.keys2_ready) { if (s.bk2_ntt) HIP_TRY(hipFree(s.bk2_ntt)); HIP_TRY(hipFree(s.bk2q_ntt)); HIP_TRY(hipFree(s.ksk2)); }
        if (s.tables2) HIP_TRY(hipFree(s.tables2));
        if (s.tables2q) HIP_TRY(hipFree(s.tables2q));
        s.keys2_ready = s.br2_lds_opt_in = s.br2q_lds_opt_in = s.ks2_lds_opt_in = false;
        s.tables2 = nullptr; s.tables2q = nullptr; s.bk2_ntt = nullptr; s.bk2q_ntt = nullptr; s.ksk2 = nullptr;
        if (s.ntt_ready) { HIP_TRY(hipFree(s.tables)); HIP_TRY(hipFree(s.tables_r4)); HIP_TRY(hipFree(s.tables512)); }
        if (s.fault_host) { (void)hipHostFree(s.fault_host); s.fault_host = nullptr; s.fault = nullptr; }   // the fault, if any, ends with the keys
        for (auto& b : s.staging) { (void)hipEventDestroy(b.done); (void)hipHostFree(b.host); }
        s.staging.clear();
        for (auto& kv : s.workspaces) (void)hipFree(kv.second.base);
        s.workspaces.clear();
        s.ntt_ready = s.keys_ready = false;
        s.br_lds_opt_in = s.ks_lds_opt_in = false;
        s.tables = nullptr; s.tables_r4 = nullptr; s.tables512 = nullptr; s.bk_ntt = nullptr; s.ksk = nullptr;
        s.prof = cufhe_amd_profile{};
    }
    lvl2_release_host_key();
    return 0;
}

int cufhe_amd_synchronize(void)
{
    {
        std::lock_guard<std::mutex> lk2(g_sched_mu);
        if (int rc = sched_synchronize_all()) return rc;
    }
    for (int i = 0; i < g_gpu_num; i++) {
        HIP_TRY(hipSetDevice(phys_device(i)));
        HIP_TRY(hipDeviceSynchronize());
        if (int rc = device_fault(i)) return rc;
    }
    return 0;
}

int cufhe_amd_stream_create(int device, void** stream)
{
    if (int rc = use_device(device)) return rc;
    hipStream_t st;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = (void*)st;
    return 0;
}
int cufhe_amd_stream_destroy(int device, void* stream)
{
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    {
        std::lock_guard<std::mutex> lk(s.staging_mu);
        auto it = s.workspaces.find((hipStream_t)stream);
        if (it != s.workspaces.end()) {
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            (void)hipFree(it->second.base);
            s.workspaces.erase(it);
        }
    }
    {
        std::lock_guard<std::mutex> lk(g_sched_mu);
        if (g_scheduler && device < g_scheduler->gpu_num()) {
            (void)g_scheduler->dev(device).retire_external_stream(stream);      // no queued launch may still name the raw handle
            g_scheduler->dev(device).forget_stream(stream);
            if (device < (int)g_sched_backends.size()) g_sched_backends[device]->forget_caller_stream(stream);
        }
    }
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return device_fault(device);      // the workspace release above may have waited for the stream
}
int cufhe_amd_stream_query(int device, void* stream)
{
    if (int rc = use_device(device)) return rc;
    if (sched_active()) {
        int q = cufhe_amd_sched_stream_query(device, stream);
        if (q <= 0) return q;
    }
    hipError_t e = hipStreamQuery((hipStream_t)stream);
    if (e == hipSuccess) {
        if (int rc = device_fault(device)) return rc;
        return 1;
    }
    if (e == hipErrorNotReady) return 0;
    return fail(-2, std::string("hipStreamQuery: ") + hipGetErrorString(e));
}
int cufhe_amd_stream_synchronize(int device, void* stream)
{
    if (int rc = use_device(device)) return rc;
    if (sched_active()) {       // what was recorded on the stream through the per-gate API: launched, complete, delivered to the tlwehosts
        std::lock_guard<std::mutex> lk(g_sched_mu);
        if (int rc = g_scheduler->dev(device).stream_synchronize(stream)) return sched_error(g_scheduler->dev(device), rc);
    }
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return device_fault(device);
}

int cufhe_amd_malloc(int device, size_t bytes, void** dptr)
{
    if (int rc = use_device(device)) return rc;
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 16));
    return 0;
}
int cufhe_amd_free(int device, void* dptr)
{
    if (int rc = use_device(device)) return rc;
    HIP_TRY(hipFree(dptr));
    return 0;
}
int cufhe_amd_host_register(void* hptr, size_t bytes)
{
    HIP_TRY(hipHostRegister(hptr, bytes, hipHostRegisterDefault));
    return 0;
}
int cufhe_amd_host_unregister(void* hptr)
{
    HIP_TRY(hipHostUnregister(hptr));
    return 0;
}
int cufhe_amd_memcpy_h2d(int device, void* stream, void* dptr, const void* hptr, size_t bytes)
{
    if (int rc = use_device(device)) return rc;
    HIP_TRY(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}
int cufhe_amd_memcpy_d2h(int device, void* stream, void* hptr, const void* dptr, size_t bytes)
{
    if (int rc = use_device(device)) return rc;
    HIP_TRY(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}

int cufhe_amd_gate(int device, void* stream, int op, int level, uint32_t* out, const uint32_t* in0,
                   const uint32_t* in1, const uint32_t* in2)
{
    return run_gates(device, stream, level, 1, [&](size_t) { return GateRef{op, out, in0, in1, in2}; });
}

int cufhe_amd_gate_batch(int device, void* stream, int level, size_t count, const int32_t* ops, int ops_stride,
                         uint32_t* out, const uint32_t* in0, const uint32_t* in1, const uint32_t* in2,
                         size_t stride_words)
{
    if (!ops) return fail(-1, "null ops");
    // the kernels move the ciphertext size of the set the entry point runs on ("param_set" / "lvl0_ring"): a smaller stride would make
    // neighbouring ciphertexts overlap and the last one run past the buffer
    if (count > 1 && (level == 0 || level == 1) && (long)stride_words < (long)cufhe_amd_ctxt_words(level))
        return fail(-1, "stride_words is smaller than a ciphertext of the active parameter set (cufhe_amd_ctxt_words)");
    return run_gates(device, stream, level, count, [&](size_t g) {
        return GateRef{ops[g * (size_t)ops_stride], out + g * stride_words, in0 ? in0 + g * stride_words : nullptr,
                       in1 ? in1 + g * stride_words : nullptr, in2 ? in2 + g * stride_words : nullptr};
    });
}

int cufhe_amd_gate_list(int device, void* stream, int level, size_t count, const int32_t* ops,
                        uint32_t* const* outs, const uint32_t* const* in0s, const uint32_t* const* in1s,
                        const uint32_t* const* in2s)
{
    if (!ops || !outs || !in0s) return fail(-1, "null array");
    return run_gates(device, stream, level, count, [&](size_t g) {
        return GateRef{ops[g], outs[g], in0s[g], in1s ? in1s[g] : nullptr, in2s ? in2s[g] : nullptr};
    });
}

int cufhe_amd_bootstrap_batch(int device, void* stream, size_t count, uint32_t* out, const uint32_t* in)
{
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    if (!s.keys_ready) return fail(-3, "Initialize(ek) has not been called for this device");
    if (!out || !in) return fail(-1, "null pointer");
    if (count == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    Scratch sc;
    if (int rc = open_scratch(s, st, count * (kLvl1Words * sizeof(uint32_t) + 2 * sizeof(LinDesc)) + 8192, &sc)) return rc;
    uint32_t* t1;
    if (int rc = sc.alloc((void**)&t1, count * kLvl1Words * sizeof(uint32_t))) return rc;
    std::vector<LinDesc> rot(count), ks(count);
    for (size_t g = 0; g < count; g++) {
        rot[g] = {in + g * kLvl0Words, in + g * kLvl0Words, t1 + g * kLvl1Words, 1, 0, 0u, 0u};
        ks[g] = {t1 + g * kLvl1Words, t1 + g * kLvl1Words, out + g * kLvl0Words, 1, 0, 0u, 0u};
    }
    LinDesc *drot, *dks;
    if (int rc = upload_descs(s, sc, rot, &drot)) return rc;
    if (int rc = upload_descs(s, sc, ks, &dks)) return rc;
    if (int rc = launch_blind_rotate(s, st, drot, count, kLvl0N, nullptr)) return rc;
    return launch_keyswitch(s, st, dks, count);
}

int cufhe_amd_blind_rotate_batch(int device, void* stream, size_t count, const uint32_t* tlwe0, uint32_t* acc, int steps)
{
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    if (!s.keys_ready) return fail(-3, "Initialize(ek) has not been called for this device");
    if (!tlwe0 || !acc) return fail(-1, "null pointer");
    if (steps < 0 || steps > kLvl0N) steps = kLvl0N;
    hipStream_t st = (hipStream_t)stream;
    std::vector<LinDesc> rot(count);
    for (size_t g = 0; g < count; g++) rot[g] = {tlwe0 + g * kLvl0Words, tlwe0 + g * kLvl0Words, nullptr, 1, 0, 0u, 0u};
    Scratch sc;
    if (int rc = open_scratch(s, st, count * sizeof(LinDesc) + 4096, &sc)) return rc;
    LinDesc* d;
    if (int rc = upload_descs(s, sc, rot, &d)) return rc;
    return launch_blind_rotate(s, st, d, count, steps, acc);
}

int cufhe_amd_keyswitch_batch(int device, void* stream, size_t count, const uint32_t* tlwe1, uint32_t* tlwe0)
{
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    if (!s.keys_ready) return fail(-3, "Initialize(ek) has not been called for this device");
    if (!tlwe0 || !tlwe1) return fail(-1, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    std::vector<LinDesc> ks(count);
    for (size_t g = 0; g < count; g++)
        ks[g] = {tlwe1 + g * kLvl1Words, tlwe1 + g * kLvl1Words, tlwe0 + g * kLvl0Words, 1, 0, 0u, 0u};
    Scratch sc;
    if (int rc = open_scratch(s, st, count * sizeof(LinDesc) + 4096, &sc)) return rc;
    LinDesc* d;
    if (int rc = upload_descs(s, sc, ks, &d)) return rc;
    return launch_keyswitch(s, st, d, count);
}

int cufhe_amd_trgsw_to_ntt_batch(int device, void* stream, size_t count, const uint32_t* trgsw, double* trgsw_ntt)
{
    if (int rc = use_device(device)) return rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = ensure_ntt(device)) return rc;
    }
    if (!trgsw || !trgsw_ntt) return fail(-1, "null pointer");
    if (count == 0) return 0;
    const size_t polys = count * kBkPolysPerStep;
    const unsigned blocks = (unsigned)((polys + kNttWavesPerBlock - 1) / kNttWavesPerBlock);
    hipLaunchKernelGGL(bk_to_ntt_kernel, dim3(blocks), dim3(kNttThreads), kNttLdsBytes, (hipStream_t)stream, trgsw_ntt,
                       trgsw, polys, g_dev[device].tables, n_inverse_balanced());
    HIP_TRY(hipGetLastError());
    return 0;
}

int cufhe_amd_trgsw_to_ntt_host(int device, void* stream, const uint32_t* trgsw_host, double* trgsw_ntt_host)
{
    if (int rc = use_device(device)) return rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = ensure_ntt(device)) return rc;
    }
    if (!trgsw_host || !trgsw_ntt_host) return fail(-1, "null pointer");
    if (g_param_set >= 0) return ps_trgsw_to_ntt_host((int)g_param_set, device, stream, trgsw_host, trgsw_ntt_host);     // the active set's sizes and limbs
    DeviceState& s = g_dev[device];
    hipStream_t st = (hipStream_t)stream;
    // torus words in, NTT-domain doubles out: both staged in the stream's grow-only workspace and in recycled pinned
    // blocks, nothing is allocated or freed per call (the reference's TRGSW2NTT is allocation-free as well,
    // src/bootstrap_gpu.cu:75-94)
    constexpr size_t in_bytes = kBkStepDoubles * sizeof(uint32_t), out_bytes = kBkStepDoubles * sizeof(double);
    Scratch sc;
    if (int rc = open_scratch(s, st, in_bytes + out_bytes + 4096, &sc)) return rc;
    uint32_t* d_in;
    double* d_out;
    if (int rc = sc.alloc((void**)&d_in, in_bytes)) return rc;
    if (int rc = sc.alloc((void**)&d_out, out_bytes)) return rc;
    PinnedBlock* blk = nullptr;
    if (int rc = acquire_staging(s, in_bytes + out_bytes, &blk)) return rc;
    StagingOwner owner{s, blk};          // releases `held`; hands the block back if a call below fails before the event is recorded
    staging_hold(s, blk, true);          // the host reads the result out of the block after the stream has finished with it
    memcpy(blk->host, trgsw_host, in_bytes);
    HIP_TRY(hipMemcpyAsync(d_in, blk->host, in_bytes, hipMemcpyHostToDevice, st));
    const unsigned blocks = (unsigned)((kBkPolysPerStep + kNttWavesPerBlock - 1) / kNttWavesPerBlock);
    hipLaunchKernelGGL(bk_to_ntt_kernel, dim3(blocks), dim3(kNttThreads), kNttLdsBytes, st, d_out, d_in, (size_t)kBkPolysPerStep,
                       s.tables, n_inverse_balanced());
    HIP_TRY(hipGetLastError());
    char* pin_out = (char*)blk->host + in_bytes;
    HIP_TRY(hipMemcpyAsync(pin_out, d_out, out_bytes, hipMemcpyDeviceToHost, st));
    if (int rc = staging_done_after(s, blk, st)) return rc;
    owner.recorded();
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(trgsw_ntt_host, pin_out, out_bytes);
    return device_fault(device);
}

int cufhe_amd_cmux_batch(int device, void* stream, size_t count, const double* trgsw_ntt, const uint32_t* c1,
                         const uint32_t* c0, uint32_t* res)
{
    if (int rc = use_device(device)) return rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = ensure_ntt(device)) return rc;
    }
    if (!trgsw_ntt || !c1 || !c0 || !res) return fail(-1, "null pointer");
    if (count == 0) return 0;
    const unsigned blocks = (unsigned)((count + kNttWavesPerBlock - 1) / kNttWavesPerBlock);
    hipLaunchKernelGGL(cmux_kernel, dim3(blocks), dim3(kNttThreads), kNttLdsBytes, (hipStream_t)stream, res, trgsw_ntt,
                       c1, c0, (int)count, g_dev[device].tables);
    HIP_TRY(hipGetLastError());
    return 0;
}

int cufhe_amd_sample_extract_keyswitch_batch(int device, void* stream, size_t count, const uint32_t* trlwe, uint32_t* tlwe0)
{
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    if (!s.keys_ready) return fail(-3, "Initialize(ek) has not been called for this device");
    if (!trlwe || !tlwe0) return fail(-1, "null pointer");
    if (count == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    Scratch sc;
    if (int rc = open_scratch(s, st, count * (kLvl1Words * sizeof(uint32_t) + sizeof(LinDesc)) + 8192, &sc)) return rc;
    uint32_t* t1;
    if (int rc = sc.alloc((void**)&t1, count * kLvl1Words * sizeof(uint32_t))) return rc;
    hipLaunchKernelGGL(sample_extract_kernel, dim3((unsigned)(count < 2048 ? count : 2048)), dim3(256), 0, st, t1, trlwe, (int)count);
    HIP_TRY(hipGetLastError());
    std::vector<LinDesc> ks(count);
    for (size_t g = 0; g < count; g++)
        ks[g] = {t1 + g * kLvl1Words, t1 + g * kLvl1Words, tlwe0 + g * kLvl0Words, 1, 0, 0u, 0u};
    LinDesc* d;
    if (int rc = upload_descs(s, sc, ks, &d)) return rc;
    return launch_keyswitch(s, st, d, count);
}

int cufhe_amd_refresh_batch(int device, void* stream, size_t count, const uint32_t* trlwe_in, uint32_t* trlwe_out)
{
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    if (!s.keys_ready) return fail(-3, "Initialize(ek) has not been called for this device");
    if (!trlwe_in || !trlwe_out) return fail(-1, "null pointer");
    if (count == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    Scratch sc;
    if (int rc = open_scratch(s, st, count * ((kLvl1Words + kLvl0Words) * sizeof(uint32_t) + 2 * sizeof(LinDesc)) + 16384, &sc)) return rc;
    uint32_t *t1, *t0;
    if (int rc = sc.alloc((void**)&t1, count * kLvl1Words * sizeof(uint32_t))) return rc;
    if (int rc = sc.alloc((void**)&t0, count * kLvl0Words * sizeof(uint32_t))) return rc;
    hipLaunchKernelGGL(sample_extract_kernel, dim3((unsigned)(count < 2048 ? count : 2048)), dim3(256), 0, st, t1, trlwe_in, (int)count);
    HIP_TRY(hipGetLastError());
    std::vector<LinDesc> ks(count), rot(count);
    for (size_t g = 0; g < count; g++) {
        ks[g] = {t1 + g * kLvl1Words, t1 + g * kLvl1Words, t0 + g * kLvl0Words, 1, 0, 0u, 0u};
        rot[g] = {t0 + g * kLvl0Words, t0 + g * kLvl0Words, nullptr, 1, 0, 0u, 0u};
    }
    LinDesc *dks, *drot;
    if (int rc = upload_descs(s, sc, ks, &dks)) return rc;
    if (int rc = upload_descs(s, sc, rot, &drot)) return rc;
    if (int rc = launch_keyswitch(s, st, dks, count)) return rc;
    return launch_blind_rotate(s, st, drot, count, kLvl0N, trlwe_out);
}

int cufhe_amd_polymul_batch(int device, void* stream, size_t count, const int32_t* a, const uint32_t* b, uint32_t* res)
{
    if (int rc = use_device(device)) return rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = ensure_ntt(device)) return rc;
    }
    if (count == 0) return 0;
    const unsigned blocks = (unsigned)((count + kNttWavesPerBlock - 1) / kNttWavesPerBlock);
    hipLaunchKernelGGL(polymul_kernel, dim3(blocks), dim3(kNttThreads), kNttLdsBytes, (hipStream_t)stream, res, a, b,
                       (int)count, g_dev[device].tables, n_inverse_balanced());
    HIP_TRY(hipGetLastError());
    return 0;
}

int cufhe_amd_polymul512_batch(int device, void* stream, size_t count, const int32_t* a, const uint32_t* b, uint32_t* res)
{
    if (int rc = use_device(device)) return rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = ensure_ntt(device)) return rc;
    }
    if (count == 0) return 0;
    const unsigned blocks = (unsigned)((count + kNttWavesPerBlock - 1) / kNttWavesPerBlock);
    hipLaunchKernelGGL(polymul512_kernel, dim3(blocks), dim3(kNttThreads), kPoly512LdsBytes, (hipStream_t)stream, res, a, b,
                       (int)count, g_dev[device].tables512 + 2, balanced(powmod_u64(kH, fpf::P_U64 - 2)));
    HIP_TRY(hipGetLastError());
    return 0;
}

int cufhe_amd_set_option(const char* key, long value)
{
    if (!key) return fail(-1, "null key");
    if (!strcmp(key, "device_base")) {
        for (auto& d : g_dev)
            if (d.ntt_ready || d.keys_ready || d.tables2) return fail(-1, "device_base must be set before Initialize");
        if (value < 0 || value + g_gpu_num > cufhe_amd_device_count()) return fail(-1, "device_base out of range");
        if (value != g_device_base) {
            std::lock_guard<std::mutex> lk2(g_sched_mu);
            sched_retire_generation();
        }
        g_device_base = (int)value;
        return 0;
    }
    if (!strcmp(key, "sched_streams") || !strcmp(key, "sched_threads")) {
        // structure of the scheduler: takes effect for the next scheduler generation
        std::lock_guard<std::mutex> lk2(g_sched_mu);
        if (value < 0 || value > 64) return fail(-1, "value out of range");
        if (g_scheduler && g_scheduler->live_ctxts()) return fail(-1, "sched_streams / sched_threads must be set before the first ciphertext is created");
        sched_retire_generation();
        (key[6] == 's' ? g_sched_streams : g_sched_threads) = value;
        return 0;
    }
    if (!strcmp(key, "sched_level_gates") || !strcmp(key, "sched_total_gates")) {
        std::lock_guard<std::mutex> lk2(g_sched_mu);
        if (value < 1 && !(key[6] == 'l' && value == -1)) return fail(-1, "value out of range (sched_level_gates: -1 = two grid rounds of the device)");
        (key[6] == 'l' ? g_sched_level_gates : g_sched_total_gates) = value;
        sched_apply_settings();
        return 0;
    }
    if (!strcmp(key, "test_fail_alloc")) { g_fail_alloc_countdown = value; return 0; }
    if (!strcmp(key, "cus_override")) {
        if (value < 0 || value > 4096) return fail(-1, "cus_override must be 0 (the device's own CU count) or a CU count");
        g_cus_override = value;
        std::lock_guard<std::mutex> lk2(g_sched_mu);
        sched_apply_settings();        // the scheduler's flush rules are in grid rounds of 8 rotations per CU
        return 0;
    }
    if (!strcmp(key, "sched_idle_gates")) {
        if (value < 1 && value != -1) return fail(-1, "sched_idle_gates must be -1 (one grid round) or a gate count");
        std::lock_guard<std::mutex> lk2(g_sched_mu);
        g_sched_idle_gates = value;
        sched_apply_settings();
        return 0;
    }
    if (!strcmp(key, "sched_copy_threads")) {
        if (value < 1 || value > 16) return fail(-1, "sched_copy_threads must be 1..16");
        std::lock_guard<std::mutex> lk2(g_sched_mu);
        g_sched_copy_threads = value;
        sched_apply_settings();          // takes effect for devices that have not flushed yet (the helpers start with the first large copy)
        return 0;
    }
    if (!strcmp(key, "sched_two_lane")) {
        std::lock_guard<std::mutex> lk2(g_sched_mu);
        if (int rc = sched_synchronize_all()) return rc;      // what is recorded was recorded under the old renaming policy
        if (value < 0 || value > 2) return fail(-1, "sched_two_lane must be 0 (never), 1 (by the cost model) or 2 (whenever a flush is eligible)");
        g_sched_two_lane = value;
        sched_apply_settings();
        return 0;
    }
    if (!strcmp(key, "sched_rename")) {
        std::lock_guard<std::mutex> lk2(g_sched_mu);
        g_sched_rename = value != 0;
        sched_apply_settings();
        return 0;
    }
    if (!strcmp(key, "sched_affinity")) { g_sched_affinity = value != 0; return 0; }
    if (!strcmp(key, "sched_zero_copy")) { g_sched_zero_copy = value != 0; return 0; }
    if (!strcmp(key, "share_devices")) { g_share_devices = value; g_phys_count = cufhe_amd_device_count(); return 0; }
    if (!strcmp(key, "ll_threshold")) { g_ll_threshold = value; return 0; }
    if (!strcmp(key, "half_threshold")) { g_half_threshold = value; return 0; }
    if (!strcmp(key, "tail_split")) { g_tail_split = value; return 0; }
    if (!strcmp(key, "br_shape")) {
        if (value < 0 || value > 3) return fail(-1, "br_shape must be 0 (rules), 1 (batch kernel), 2 (paired low-latency kernel) or 3 (single)");
        g_br_shape = value;        // of the calling thread
        return 0;
    }
    if (!strcmp(key, "ks_wg_threshold")) { g_ks_wg_threshold = value; return 0; }
    if (!strcmp(key, "ks_split_threshold")) { g_ks_split_threshold = value; return 0; }
    if (!strcmp(key, "ps_batch_threshold")) { g_ps_batch_threshold = value; return 0; }
    if (!strcmp(key, "ll2_threshold")) { g_ll2_threshold = value; return 0; }
    if (!strcmp(key, "param_set") || !strcmp(key, "lvl0_param_set")) {
        if (value >= 0) {
            cufhe_amd_ps_params p;
            if (int rc = cufhe_amd_ps_get_params((int)value, &p)) return rc;
            // ciphertext device slots are carved for the largest compiled sizes (HipBackend::slot_words), so a set only has to fit them
            if ((int)p.lvl0_words > kLvl0Words || (int)p.lvl1_words > kLvl1Words) return fail(-1, "param_set: the set's ciphertexts exceed the per-gate API's buffers");
        }
        if (value != g_param_set) {
            // recorded gates were sized and routed for the old set: they complete first
            std::lock_guard<std::mutex> lk2(g_sched_mu);
            if (int rc = sched_synchronize_all()) return rc;
        }
        g_param_set = value;
        return 0;
    }
    if (!strcmp(key, "ks_slices")) {
        if (value != -1 && (value < 1 || value > 64 || (value & (value - 1)))) return fail(-1, "ks_slices must be -1 or a power of two 1..64");
        g_ks_slices = value;
        return 0;
    }
    if (!strcmp(key, "ks_per_wg")) {
        if (value != -1 && (value < 1 || value > 16)) return fail(-1, "ks_per_wg must be -1 or 1..16");
        g_ks_per_wg = value;
        return 0;
    }
    if (!strcmp(key, "lvl2_kernel")) {
        if (value < -1 || value > 1) return fail(-1, "lvl2_kernel must be -1 (by cost), 0 (eight half waves) or 1 (four quarter waves)");
        g_lvl2_kernel = value;
        return 0;
    }
    if (!strcmp(key, "lvl0_ring")) {
        if (value != 1024 && value != 2048) return fail(-1, "lvl0_ring must be 1024 or 2048");
        g_lvl0_ring = value;
        return 0;
    }
    return fail(-1, std::string("unknown option ") + key);
}

int cufhe_amd_profile_enable(int device, int on)
{
    if (int rc = check_device(device)) return rc;
    g_dev[device].profiling = on != 0;
    return 0;
}

int cufhe_amd_probe_clock(int device, double* hz)
{
    if (int rc = use_device(device)) return rc;
    if (!hz) return fail(-1, "null");
    constexpr int kBlocks = 256, kWaves = 8;
    double* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, (kBlocks * kWaves + 1) * sizeof(double)));
    std::vector<double> h(kBlocks * kWaves);
    for (int rep = 0; rep < 2; rep++)        // the second launch runs on a chip that is already under load
        hipLaunchKernelGGL(clock_probe_kernel, dim3(kBlocks), dim3(64 * kWaves), 0, 0, d, d + kBlocks * kWaves, 3000);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(-2, std::string("clock probe: ") + hipGetErrorString(e));
    std::sort(h.begin(), h.end());
    *hz = h[h.size() / 2];
    return 0;
}

int cufhe_amd_profile_get(int device, cufhe_amd_profile* out, int reset)
{
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    auto drain = [&](std::vector<EventPair>& v, double& ms, uint64_t& launches, uint64_t& units) -> int {
        for (auto& e : v) {
            HIP_TRY(hipEventSynchronize(e.b));
            float t = 0;
            HIP_TRY(hipEventElapsedTime(&t, e.a, e.b));
            ms += t; launches += 1; units += e.units;
            (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
        }
        v.clear();
        return 0;
    };
    std::vector<EventPair> br, ks;
    {
        std::lock_guard<std::mutex> lk(s.staging_mu);     // the scheduler's worker thread records launches too
        br.swap(s.br_events);
        ks.swap(s.ks_events);
    }
    if (int rc = drain(br, s.prof.blind_rotate_ms, s.prof.blind_rotate_launches, s.prof.blind_rotations)) return rc;
    if (int rc = drain(ks, s.prof.keyswitch_ms, s.prof.keyswitch_launches, s.prof.keyswitches)) return rc;
    if (out) *out = s.prof;
    if (reset) s.prof = cufhe_amd_profile{};
    return device_fault(device);      // hipEventSynchronize above observed completions
}

}  // extern "C"
