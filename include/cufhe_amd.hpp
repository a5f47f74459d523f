// cufhe_amd.hpp -- the cuFHE gate API (namespace cufhe) on top of the C ABI of cufhe_amd.h.
//
// Drop-in for the gate path of /root/reference/include/cufhe_gpu.cuh: same names, same
// argument order, same completion rules (results are in `tlwehost` after Synchronize() or
// once StreamQuery(st) is true), same abort-on-error behaviour
// (include/details/error_gpu.cuh:40-60).  Host code stays C++; this header contains no HIP.
//
//   reference (include/cufhe_gpu.cuh)                 here
//   SetGPUNum / Initialize / CleanUp / Synchronize    :54-74      same
//   class Stream, StreamQuery                         :152-191    same (Create/Destroy/st/device_id)
//   template<class P> struct Ctxt                     :102-121    same members tlwehost, tlwedevices
//   CtxtCopyH2D / CtxtCopyD2H / CopyOnHost            :193-255    same
//   And ... NMux, Not, Copy and g-variants            :218-313    same, P in {lvl0param, lvl1param}
//
// TFHEpp is not vendored in the reference tree (thirdparties/TFHEpp is an empty submodule),
// so the parameter structs and TLWE<P> used by the reference's signatures are declared
// here with the same member names; define CUFHE_AMD_USE_TFHEPP before including this
// header to take them from <params.hpp>/<cloudkey.hpp> instead (then Initialize(ek) accepts
// a TFHEpp::EvalKey exactly as the reference does, src/cufhe_gates_gpu.cu:42-47).
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "cufhe_amd.h"

// The reference has ONE selector for its parameter set: the macro TFHEpp is built with (CMakeLists.txt:8-24: USE_80BIT_SECURITY,
// USE_CGGI19, USE_CONCRETE ...) fixes the numbers of lvl0param / lvl1param / lvl10param, and every kernel is a template over those
// structs (include/bootstrap_gpu.cuh:51-53).  Here the library carries its sets compiled in (cufhe_amd_ps_get_params), and this header
// finds the one the CALLER's structs describe -- at compile time, by their numbers (detail::kParamSet below: a static_assert with the
// numbers in its message when no compiled set has them), and again at run time (cufhe_amd_initialize_params refuses keys whose
// numbers match no set: sizes alone do not see Bgbit, and t * 2^(basebit-1) is 16 for both (8, 2) and (4, 3)).  Nothing else selects:
//   with -DCUFHE_AMD_USE_TFHEPP the structs are TFHEpp's own, whatever macro TFHEpp was built with;
//   without it they are declared below with the BASELINE numbers, or those of
//     -DCUFHE_AMD_PARAM_SET_K2N512   n = 630, N = 512,  k = 2, l = 3, Bg = 2^6
//     -DCUFHE_AMD_PARAM_SET_CGGI16   n = 500, N = 1024, k = 1, l = 2, Bg = 2^10 (the original TFHE 80-bit set)
//   the reference's own -DUSE_SMALL_NTT_MODULUS (CMakeLists.txt:12,26-28; or -DCUFHE_AMD_PARAM_SET_SMALLMOD) keeps the numbers and asks
//   for the small NTT modulus P = 625 * 2^20 + 1 with torus discretisation switching (include/ntt_gpu/ntt_small_modulus.cuh) --
//   approximate, as there; such a build has no CMUXNTT / TRGSW2NTT, as in the reference (src/cufhe_gates_gpu.cu:68-86).
// Initialize(bk, ..) / Initialize(ek) loads the keys into that set and the whole per-gate API runs on it: Ctxt<lvl0param> has n + 1
// words, Ctxt<lvl1param> k N + 1, And ... NMux run blind rotate -> key switch on the first and key switch -> blind rotate on the
// second (src/bootstrap_gpu.cu:383-421); GateBootstrappingTLWE2TRLWElvl01NTT / Refresh / SampleExtractAndKeySwitch / CMUXNTT /
// TRGSW2NTT run on the set as well (src/cufhe_gates_gpu.cu:68-146).  The N = 2048 ring (namespace lvl2) has its own keys.
#if defined(CUFHE_AMD_PARAM_SET_SMALLMOD) || defined(USE_SMALL_NTT_MODULUS)
#define CUFHE_AMD_SMALL_NTT_MODULUS 1
#endif

#ifdef CUFHE_AMD_USE_TFHEPP
#include <cloudkey.hpp>
#include <params.hpp>
#else
namespace TFHEpp {
struct lvl0param {
    using T = uint32_t;
#if defined(CUFHE_AMD_PARAM_SET_CGGI16)
    static constexpr uint32_t n = 500, k = 1;
#else
    static constexpr uint32_t n = 630, k = 1;
#endif
    static constexpr T mu = 1u << 29;
    static constexpr T μ = mu;             // TFHEpp's spelling
};
struct lvl1param {
    using T = uint32_t;
#if defined(CUFHE_AMD_PARAM_SET_K2N512)
    static constexpr uint32_t nbit = 9, n = 1u << nbit, k = 2, l = 3, Bgbit = 6, Bg = 1u << Bgbit;
#elif defined(CUFHE_AMD_PARAM_SET_CGGI16)
    static constexpr uint32_t nbit = 10, n = 1u << nbit, k = 1, l = 2, Bgbit = 10, Bg = 1u << Bgbit;
#else
    static constexpr uint32_t nbit = 10, n = 1u << nbit, k = 1, l = 3, Bgbit = 6, Bg = 1u << Bgbit;
#endif
    static constexpr T mu = 1u << 29;
    static constexpr T μ = mu;
};
struct lvl01param { using domainP = lvl0param; using targetP = lvl1param; };
struct lvl10param {
    using domainP = lvl1param; using targetP = lvl0param;
    static constexpr uint32_t t = 8, basebit = 2;
};
template <class P> using TLWE = std::array<typename P::T, P::k * P::n + 1>;
template <class P> using TRLWE = std::array<std::array<typename P::T, P::n>, P::k + 1>;
template <class P> using TRGSW = std::array<TRLWE<P>, (P::k + 1) * P::l>;
}  // namespace TFHEpp
#endif

namespace cufhe {

namespace detail {
inline void check(int rc, const char* what, const char* file, int line)
{
    if (rc < 0) {   // CuSafeCall semantics: print and exit(-1)
        std::fprintf(stderr, "%s failed at %s:%d : %s\n", what, file, line, cufhe_amd_last_error());
        std::exit(-1);
    }
}
// which ciphertext a parameter struct stands for, by TYPE (sets whose lvl0 n equals their lvl1 n exist in TFHEpp's family)
template <class P> constexpr int level_of()
{
    static_assert(std::is_same<P, TFHEpp::lvl0param>::value || std::is_same<P, TFHEpp::lvl1param>::value,
                  "cufhe_amd: Ctxt<P> and the gates are specialised for P = TFHEpp::lvl0param and TFHEpp::lvl1param (include/cufhe_gpu.cuh:218-313)");
    return std::is_same<P, TFHEpp::lvl1param>::value ? 1 : 0;
}

// The sets compiled into libcufhe_amd.so (cufhe_amd/csrc/kernels_ps.hip.h; cufhe_amd_ps_get_params reports the same numbers, and
// tests/test_capi.py holds this table against the library): index = the library's set index.
struct SetNumbers { uint32_t n, nbit, k, l, Bgbit, t, basebit, key_limbs, small_ntt_modulus; };
constexpr uint32_t kSmallNttP = (625u << 20) + 1;
constexpr SetNumbers kCompiledSets[] = {
    {630, 10, 1, 3, 6, 8, 2, 1, 0},              // 0 "default": BASELINE.json
    {630, 9, 2, 3, 6, 8, 2, 1, 0},               // 1 "k2n512"
    {500, 10, 1, 2, 10, 8, 2, 2, 0},             // 2 "cggi16": two 16-bit key limbs
    {630, 10, 1, 3, 6, 8, 2, 1, kSmallNttP},     // 3 "smallmod"
};
#ifdef CUFHE_AMD_SMALL_NTT_MODULUS
constexpr uint32_t kCallerSmallNttModulus = kSmallNttP;
#else
constexpr uint32_t kCallerSmallNttModulus = 0;
#endif
/// the numbers this translation unit was compiled with: TFHEpp's structs (or the stand-ins above)
constexpr cufhe_amd_param_numbers kCallerNumbers = {TFHEpp::lvl0param::n, TFHEpp::lvl1param::nbit, TFHEpp::lvl1param::k, TFHEpp::lvl1param::l,
                                                    TFHEpp::lvl1param::Bgbit, TFHEpp::lvl10param::t, TFHEpp::lvl10param::basebit,
                                                    kCallerSmallNttModulus};
constexpr int match_param_set()
{
    for (int i = 0; i < (int)(sizeof(kCompiledSets) / sizeof(kCompiledSets[0])); i++) {
        const SetNumbers& s = kCompiledSets[i];
        if (s.n == kCallerNumbers.n && s.nbit == kCallerNumbers.nbit && s.k == kCallerNumbers.k && s.l == kCallerNumbers.l &&
            s.Bgbit == kCallerNumbers.Bgbit && s.t == kCallerNumbers.t && s.basebit == kCallerNumbers.basebit &&
            s.small_ntt_modulus == kCallerNumbers.small_ntt_modulus)
            return i;
    }
    return -1;
}
constexpr int kParamSet = match_param_set();
static_assert(kParamSet >= 0,
              "cufhe_amd: no parameter set compiled into libcufhe_amd.so has the numbers of this build's TFHEpp::lvl0param::n, "
              "lvl1param::{nbit, k, l, Bgbit} and lvl10param::{t, basebit} (compiled: n=630 N=1024 k=1 l=3 Bgbit=6 | n=630 N=512 k=2 l=3 Bgbit=6 | "
              "n=500 N=1024 k=1 l=2 Bgbit=10, all with t=8 basebit=2): the gate kernels are specialised per set and would compute garbage on "
              "another one -- add the set to cufhe_amd/csrc/kernels_ps.hip.h and to detail::kCompiledSets");
static_assert(TFHEpp::lvl0param::k == 1 && TFHEpp::lvl1param::n == (1u << TFHEpp::lvl1param::nbit) &&
              std::is_same<TFHEpp::lvl0param::T, uint32_t>::value && std::is_same<TFHEpp::lvl1param::T, uint32_t>::value &&
              TFHEpp::lvl0param::mu == (1u << 29) && TFHEpp::lvl1param::mu == (1u << 29),
              "cufhe_amd: lvl0 / lvl1 ciphertexts are 32-bit torus words with mu = 2^29 and lvl0param::k = 1");
}  // namespace detail
/// the library's index of the parameter set this build runs on (0: BASELINE numbers, hand-scheduled kernels) and its key limbs
constexpr int kParamSetIndex = detail::kParamSet;
constexpr uint32_t kKeyLimbs = detail::kCompiledSets[detail::kParamSet >= 0 ? detail::kParamSet : 0].key_limbs;
#define CUFHE_AMD_CHECK(expr) ::cufhe::detail::check((expr), #expr, __FILE__, __LINE__)

namespace detail {
/// Ciphertexts take the sizes of the set the per-gate API runs on ("param_set"): a Ctxt constructed BEFORE Initialize(ek) -- legal in
/// the reference, whose sizes are compile-time -- must already see this build's set.
inline void select_param_set()
{
    static const int once = [] {
        CUFHE_AMD_CHECK(cufhe_amd_set_option("param_set", kParamSetIndex == 0 ? -1 : kParamSetIndex));
        return 0;
    }();
    (void)once;
}
}  // namespace detail

inline int& stream_count() { static int c = 0; return c; }
namespace detail {
inline int& gpu_num_mirror() { static int n = cufhe_amd_get_gpu_num(); return n; }
}  // namespace detail
/// `extern int _gpuNum; extern int streamCount;` of include/cufhe_gpu.cuh:44-46 (defined in src/cufhe_gates_gpu.cu:35-36):
/// public globals of the reference that source code reads (loops over devices, stream bookkeeping).  Here they are
/// references to the shim's state: `_gpuNum` mirrors the library's GPU count (set by SetGPUNum, as in the reference),
/// `streamCount` is the counter the default Stream constructor round-robins with.
inline int& _gpuNum = detail::gpu_num_mirror();
inline int& streamCount = stream_count();

inline void SetGPUNum(int gpuNum)
{
    CUFHE_AMD_CHECK(cufhe_amd_set_gpu_num(gpuNum));
    _gpuNum = gpuNum;
}
inline int GetGPUNum() { return cufhe_amd_get_gpu_num(); }
inline void Initialize() { CUFHE_AMD_CHECK(cufhe_amd_initialize_ntt()); }
/// bk: [n][(k+1)l][k+1][N], ksk: [kN][t][2^(basebit-1)][n+1] torus words (TFHEpp's in-memory layouts).  The library is handed the
/// NUMBERS of this build's parameter structs and loads the keys into the compiled set that has them (cufhe_amd_initialize_params).
inline void Initialize(const uint32_t* bk, size_t bk_words, const uint32_t* ksk, size_t ksk_words)
{
    CUFHE_AMD_CHECK(cufhe_amd_initialize_params(&detail::kCallerNumbers, bk, bk_words, ksk, ksk_words));
    if (cufhe_amd_ctxt_words(0) != (int)(TFHEpp::lvl0param::k * TFHEpp::lvl0param::n + 1) ||
        cufhe_amd_ctxt_words(1) != (int)(TFHEpp::lvl1param::k * TFHEpp::lvl1param::n + 1) ||
        cufhe_amd_find_param_set(&detail::kCallerNumbers) != kParamSetIndex) {
        std::fprintf(stderr, "cufhe_amd.hpp: this header's table of compiled sets does not match libcufhe_amd.so (set %d)\n", kParamSetIndex);
        std::exit(-1);
    }
}
#ifdef CUFHE_AMD_USE_TFHEPP
inline void Initialize(const TFHEpp::EvalKey& ek)
{
    const auto& bk = ek.getbk<TFHEpp::lvl01param>();
    const auto& ksk = ek.getiksk<TFHEpp::lvl10param>();
    Initialize(reinterpret_cast<const uint32_t*>(bk.data()), sizeof(bk) / sizeof(uint32_t),
               reinterpret_cast<const uint32_t*>(ksk.data()), sizeof(ksk) / sizeof(uint32_t));
}
#endif
inline void CleanUp() { CUFHE_AMD_CHECK(cufhe_amd_cleanup()); }
inline void Synchronize() { CUFHE_AMD_CHECK(cufhe_amd_synchronize()); }

/// N = 2048 ring / 64-bit torus (lvl02 blind rotate, lvl20 key switch; no reference counterpart,
/// see include/cufhe_amd.h).  Gates take and return lvl0 ciphertexts in device memory.
namespace lvl2 {
/// bk: [n][(k+1)l][k+1][N] uint64 torus words, ksk: [kN][t][2^(basebit-1)][n+1] uint32
inline void Initialize(const uint64_t* bk, size_t bk_words, const uint32_t* ksk, size_t ksk_words)
{
    CUFHE_AMD_CHECK(cufhe_amd_lvl2_initialize(bk, bk_words, ksk, ksk_words));
}
#ifdef CUFHE_AMD_USE_TFHEPP
inline void Initialize(const TFHEpp::EvalKey& ek)
{
    const auto& bk = ek.getbk<TFHEpp::lvl02param>();
    const auto& ksk = ek.getiksk<TFHEpp::lvl20param>();
    Initialize(reinterpret_cast<const uint64_t*>(bk.data()), sizeof(bk) / sizeof(uint64_t),
               reinterpret_cast<const uint32_t*>(ksk.data()), sizeof(ksk) / sizeof(uint32_t));
}
#endif
/// `count` gates of one op on contiguous lvl0 ciphertexts (stride n + 1 words) of device `device`
inline void GateBatch(int op, size_t count, uint32_t* out, const uint32_t* in0, const uint32_t* in1,
                      const uint32_t* in2, int device = 0, void* stream = nullptr)
{
    const int32_t o = op;
    cufhe_amd_lvl2_params p;
    CUFHE_AMD_CHECK(cufhe_amd_lvl2_get_params(&p));
    CUFHE_AMD_CHECK(cufhe_amd_lvl2_gate_batch(device, stream, count, &o, 0, out, in0, in1, in2, p.lvl0_words));
}
}  // namespace lvl2

/// Bootstrap(out, in, mu, st, gpuNum) of include/bootstrap_gpu.cuh:64-65 on device pointers
/// (lvl0 TLWE in, refreshed lvl0 TLWE out); mu must be lvl1param::mu, the only test vector built in
inline void Bootstrap(uint32_t* out, const uint32_t* in, uint32_t mu, void* st, int gpuNum)
{
    if (mu != TFHEpp::lvl1param::μ) { std::fprintf(stderr, "Bootstrap: unsupported mu\n"); std::exit(-1); }
    CUFHE_AMD_CHECK(cufhe_amd_bootstrap_batch(gpuNum, st, 1, out, in));
}

}  // namespace cufhe
/// The type `Stream::st()` returns (`cudaStream_t` in the reference, include/cufhe_gpu.cuh:183).  hipStream_t IS
/// `struct ihipStream_t*`; declaring the tag here gives the very same type without pulling the HIP headers into host
/// code, so a caller that does include <hip/hip_runtime_api.h> can hand `st.st()` straight to hipMemcpyAsync & co.
struct ihipStream_t;
typedef struct ihipStream_t* cufheStream_t;
namespace cufhe {

/// class Stream, include/cufhe_gpu.cuh:152-189 (passed by value, never auto-destroyed)
class Stream {
   public:
    inline Stream() : st_(nullptr), _device_id(streamCount % _gpuNum) { streamCount++; }
    inline Stream(int device_id) : st_(nullptr), _device_id(device_id) { streamCount++; }
    inline ~Stream() {}
    inline void Create()
    {
        void* s = nullptr;
        CUFHE_AMD_CHECK(cufhe_amd_stream_create(_device_id, &s));
        st_ = static_cast<cufheStream_t>(s);
    }
    inline void Destroy() { CUFHE_AMD_CHECK(cufhe_amd_stream_destroy(_device_id, st_)); st_ = nullptr; }
    /// The stream the gates of this Stream were enqueued on, as in the reference (include/cufhe_gpu.cuh:183): work the caller puts on
    /// the handle after this call -- hipMemcpyAsync from `out.tlwedevices[d]`, hipEventRecord, hipStreamSynchronize -- runs behind every
    /// gate issued on the Stream so far, and gates issued afterwards run behind what the caller has put on the handle before them
    /// (cufhe_amd_stream_fence).  Ask again after issuing more gates: a handle kept from an earlier call is only ordered behind the gates
    /// issued before THAT call.  tlwehost is filled by Synchronize() / StreamQuery(st) / StreamSynchronize(st), not by waiting on the handle.
    inline cufheStream_t st()
    {
        if (st_) CUFHE_AMD_CHECK(cufhe_amd_stream_fence(_device_id, st_));
        return st_;
    }
    /// the bare handle, for the library's own entry points (no ordering side effect)
    inline cufheStream_t raw() const { return st_; }
    inline int device_id() const { return _device_id; }

   private:
    cufheStream_t st_;
    int _device_id;
};

inline bool StreamQuery(Stream st)
{
    int q = cufhe_amd_stream_query(st.device_id(), st.raw());
    CUFHE_AMD_CHECK(q);
    return q == 1;
}
/// cudaStreamSynchronize(st.st()) of a reference program: everything issued on `st` is complete, results are in the tlwehosts and in
/// the ciphertexts' own device buffers (tlwedevices)
inline void StreamSynchronize(Stream st) { CUFHE_AMD_CHECK(cufhe_amd_stream_synchronize(st.device_id(), st.raw())); }

/// template<class P> struct Ctxt, include/cufhe_gpu.cuh:102-121
template <class P>
struct Ctxt {
    Ctxt()
    {
        detail::select_param_set();
        CUFHE_AMD_CHECK(cufhe_amd_ctxt_create(detail::level_of<P>(), tlwehost.data(), &handle));
        tlwedevices.resize(GetGPUNum());
        for (int i = 0; i < GetGPUNum(); i++) tlwedevices[i] = cufhe_amd_ctxt_device_ptr(handle, i);
    }
    ~Ctxt() { cufhe_amd_ctxt_destroy(handle); }
    Ctxt(const Ctxt&) = delete;
    Ctxt& operator=(const Ctxt&) = delete;

    alignas(64) TFHEpp::TLWE<P> tlwehost;
    std::vector<typename P::T*> tlwedevices;   // the ciphertext's own device buffers: they hold its value whenever completion has been observed
    cufhe_amd_ctxt* handle = nullptr;
};

template <class P> inline void CtxtCopyH2D(Ctxt<P>& c, Stream st) { CUFHE_AMD_CHECK(cufhe_amd_enqueue_copy(st.device_id(), st.raw(), c.handle, 1)); }
template <class P> inline void CtxtCopyD2H(Ctxt<P>& c, Stream st) { CUFHE_AMD_CHECK(cufhe_amd_enqueue_copy(st.device_id(), st.raw(), c.handle, 0)); }
template <class P> inline void CopyOnHost(Ctxt<P>& out, Ctxt<P>& in) { out.tlwehost = in.tlwehost; }

#define CUFHE_AMD_GATE2(Name, OP)                                                                     \
    template <class P> inline void Name(Ctxt<P>& out, Ctxt<P>& in0, Ctxt<P>& in1, Stream st)          \
    { CUFHE_AMD_CHECK(cufhe_amd_enqueue_gate(st.device_id(), st.raw(), OP, 1, out.handle, in0.handle, in1.handle, nullptr)); } \
    template <class P> inline void g##Name(Ctxt<P>& out, Ctxt<P>& in0, Ctxt<P>& in1, Stream st)       \
    { CUFHE_AMD_CHECK(cufhe_amd_enqueue_gate(st.device_id(), st.raw(), OP, 0, out.handle, in0.handle, in1.handle, nullptr)); }
#define CUFHE_AMD_GATE1(Name, OP)                                                                     \
    template <class P> inline void Name(Ctxt<P>& out, Ctxt<P>& in, Stream st)                         \
    { CUFHE_AMD_CHECK(cufhe_amd_enqueue_gate(st.device_id(), st.raw(), OP, 1, out.handle, in.handle, nullptr, nullptr)); } \
    template <class P> inline void g##Name(Ctxt<P>& out, Ctxt<P>& in, Stream st)                      \
    { CUFHE_AMD_CHECK(cufhe_amd_enqueue_gate(st.device_id(), st.raw(), OP, 0, out.handle, in.handle, nullptr, nullptr)); }
#define CUFHE_AMD_GATE3(Name, OP)                                                                     \
    template <class P> inline void Name(Ctxt<P>& out, Ctxt<P>& inc, Ctxt<P>& in1, Ctxt<P>& in0, Stream st) \
    { CUFHE_AMD_CHECK(cufhe_amd_enqueue_gate(st.device_id(), st.raw(), OP, 1, out.handle, inc.handle, in1.handle, in0.handle)); } \
    template <class P> inline void g##Name(Ctxt<P>& out, Ctxt<P>& inc, Ctxt<P>& in1, Ctxt<P>& in0, Stream st) \
    { CUFHE_AMD_CHECK(cufhe_amd_enqueue_gate(st.device_id(), st.raw(), OP, 0, out.handle, inc.handle, in1.handle, in0.handle)); }

CUFHE_AMD_GATE2(And, CUFHE_AMD_AND)
CUFHE_AMD_GATE2(AndYN, CUFHE_AMD_ANDYN)
CUFHE_AMD_GATE2(AndNY, CUFHE_AMD_ANDNY)
CUFHE_AMD_GATE2(Or, CUFHE_AMD_OR)
CUFHE_AMD_GATE2(OrYN, CUFHE_AMD_ORYN)
CUFHE_AMD_GATE2(OrNY, CUFHE_AMD_ORNY)
CUFHE_AMD_GATE2(Nand, CUFHE_AMD_NAND)
CUFHE_AMD_GATE2(Nor, CUFHE_AMD_NOR)
CUFHE_AMD_GATE2(Xor, CUFHE_AMD_XOR)
CUFHE_AMD_GATE2(Xnor, CUFHE_AMD_XNOR)
CUFHE_AMD_GATE1(Not, CUFHE_AMD_NOT)
CUFHE_AMD_GATE1(Copy, CUFHE_AMD_COPY)
CUFHE_AMD_GATE3(Mux, CUFHE_AMD_MUX)
CUFHE_AMD_GATE3(NMux, CUFHE_AMD_NMUX)

// ---- TRLWE-level primitives, include/cufhe_gpu.cuh:123-146,209-216,282-285 ----
// Same names and operands as the reference.  GateBootstrappingTLWE2TRLWElvl01NTT, Refresh,
// SampleExtractAndKeySwitch and CMUXNTT (and their g-forms) are RECORDED like gates and launched in batches: as in the
// reference, results are in the host members after Synchronize() or StreamQuery(st) == true.  TRGSW2NTT completes
// before returning (the reference waits for its D2H copy too).

// All of them run on the parameter set of this build (kParamSetIndex).  CMUXNTT and TRGSW2NTT are not declared in a small-modulus
// build -- the reference leaves them out there too (src/cufhe_gates_gpu.cu:68-86, src/bootstrap_gpu.cu:73-95).

/// struct cuFHETRLWElvl1, include/cufhe_gpu.cuh:124-134
struct cuFHETRLWElvl1 {
    TFHEpp::TRLWE<TFHEpp::lvl1param> trlwehost;
    std::vector<TFHEpp::lvl1param::T*> trlwedevices;
    cufhe_amd_ctxt* handle = nullptr;
    cuFHETRLWElvl1()
    {
        detail::select_param_set();
        CUFHE_AMD_CHECK(cufhe_amd_ctxt_create(2, trlwehost[0].data(), &handle));
        trlwedevices.resize(GetGPUNum());
        for (int i = 0; i < GetGPUNum(); i++) trlwedevices[i] = cufhe_amd_ctxt_device_ptr(handle, i);
    }
    ~cuFHETRLWElvl1() { cufhe_amd_ctxt_destroy(handle); }
    cuFHETRLWElvl1(const cuFHETRLWElvl1&) = delete;
    cuFHETRLWElvl1& operator=(const cuFHETRLWElvl1&) = delete;
};
static_assert(sizeof(TFHEpp::TRLWE<TFHEpp::lvl1param>) == (TFHEpp::lvl1param::k + 1) * TFHEpp::lvl1param::n * sizeof(uint32_t), "TRLWE is (k+1) N contiguous words");

#ifndef CUFHE_AMD_SMALL_NTT_MODULUS

/// struct cuFHETRGSWNTTlvl1, :136-146.  The NTT-domain words are this library's (exact
/// residues mod a 50-bit prime carried in doubles); like the reference's FFP words they are
/// only meaningful to CMUXNTT.  The device buffers belong to a scheduler handle (level 3), so that CMUXNTT is
/// ordered against TRGSW2NTT and against other uses of the same TRGSW like any recorded gate.
struct cuFHETRGSWNTTlvl1 {
    // (k+1) l rows of k+1 polynomials, once per key limb of the set (cggi16: two; cufhe_amd_ctxt_words(3) / 2 doubles)
    alignas(64) std::array<double, kKeyLimbs * (TFHEpp::lvl1param::k + 1) * TFHEpp::lvl1param::l * (TFHEpp::lvl1param::k + 1) * TFHEpp::lvl1param::n> trgswhost;
    std::vector<double*> trgswdevices;
    cufhe_amd_ctxt* handle = nullptr;
    cuFHETRGSWNTTlvl1()
    {
        detail::select_param_set();
        CUFHE_AMD_CHECK(cufhe_amd_ctxt_create(3, reinterpret_cast<uint32_t*>(trgswhost.data()), &handle));
        if (cufhe_amd_ctxt_words(3) != (int)(2 * trgswhost.size())) {
            std::fprintf(stderr, "cuFHETRGSWNTTlvl1: the library's TRGSW holder has %d words, this build's %zu\n", cufhe_amd_ctxt_words(3), 2 * trgswhost.size());
            std::exit(-1);
        }
        trgswdevices.resize(GetGPUNum());
        for (int i = 0; i < GetGPUNum(); i++) trgswdevices[i] = reinterpret_cast<double*>(cufhe_amd_ctxt_device_ptr(handle, i));
    }
    ~cuFHETRGSWNTTlvl1() { cufhe_amd_ctxt_destroy(handle); }
    cuFHETRGSWNTTlvl1(const cuFHETRGSWNTTlvl1&) = delete;
    cuFHETRGSWNTTlvl1& operator=(const cuFHETRGSWNTTlvl1&) = delete;
};

/// TRGSW2NTT, src/bootstrap_gpu.cu:75-94: torus-domain TRGSW -> trgswntt.trgswhost, complete on return (the reference
/// waits for its D2H as well), and -- as in the reference -- trgswntt.trgswdevices[st.device_id()]: the upload is recorded on
/// the holder's handle, so gCMUXNTT finds the words on the device and a holder refilled between two CMUXNTT calls is
/// re-uploaded for the second.  Staging is pooled inside the library: nothing is allocated per call.
inline void TRGSW2NTT(cuFHETRGSWNTTlvl1& trgswntt, const TFHEpp::TRGSW<TFHEpp::lvl1param>& trgsw, Stream& st)
{
    CUFHE_AMD_CHECK(cufhe_amd_trgsw_to_ntt(st.device_id(), st.raw(), reinterpret_cast<const uint32_t*>(trgsw.data()), trgswntt.handle));
}
#endif  // CUFHE_AMD_SMALL_NTT_MODULUS
/// gGateBootstrappingTLWE2TRLWElvl01NTT / GateBootstrappingTLWE2TRLWElvl01NTT, src/cufhe_gates_gpu.cu:86-104
inline void gGateBootstrappingTLWE2TRLWElvl01NTT(cuFHETRLWElvl1& out, Ctxt<TFHEpp::lvl0param>& in, Stream st)
{
    CUFHE_AMD_CHECK(cufhe_amd_enqueue_trlwe_op(st.device_id(), st.raw(), CUFHE_AMD_TL_BOOTSTRAP, 0, out.handle, in.handle));
}
inline void GateBootstrappingTLWE2TRLWElvl01NTT(cuFHETRLWElvl1& out, Ctxt<TFHEpp::lvl0param>& in, Stream st)
{
    CUFHE_AMD_CHECK(cufhe_amd_enqueue_trlwe_op(st.device_id(), st.raw(), CUFHE_AMD_TL_BOOTSTRAP, 1, out.handle, in.handle));
}
/// gRefresh / Refresh, src/cufhe_gates_gpu.cu:106-124
inline void gRefresh(cuFHETRLWElvl1& out, cuFHETRLWElvl1& in, Stream st)
{
    CUFHE_AMD_CHECK(cufhe_amd_enqueue_trlwe_op(st.device_id(), st.raw(), CUFHE_AMD_TL_REFRESH, 0, out.handle, in.handle));
}
inline void Refresh(cuFHETRLWElvl1& out, cuFHETRLWElvl1& in, Stream st)
{
    CUFHE_AMD_CHECK(cufhe_amd_enqueue_trlwe_op(st.device_id(), st.raw(), CUFHE_AMD_TL_REFRESH, 1, out.handle, in.handle));
}
/// gSampleExtractAndKeySwitch / SampleExtractAndKeySwitch, src/cufhe_gates_gpu.cu:126-146
/// (both upload `in.trlwehost`, as the reference does; only the second fetches the result)
inline void gSampleExtractAndKeySwitch(Ctxt<TFHEpp::lvl0param>& out, const cuFHETRLWElvl1& in, Stream st)
{
    CUFHE_AMD_CHECK(cufhe_amd_enqueue_copy(st.device_id(), st.raw(), in.handle, 1));
    CUFHE_AMD_CHECK(cufhe_amd_enqueue_trlwe_op(st.device_id(), st.raw(), CUFHE_AMD_TL_SEIKS, 0, out.handle, in.handle));
}
inline void SampleExtractAndKeySwitch(Ctxt<TFHEpp::lvl0param>& out, const cuFHETRLWElvl1& in, Stream st)
{
    gSampleExtractAndKeySwitch(out, in, st);
    CUFHE_AMD_CHECK(cufhe_amd_enqueue_copy(st.device_id(), st.raw(), out.handle, 0));
}
#ifndef CUFHE_AMD_SMALL_NTT_MODULUS
/// CMUXNTT, src/cufhe_gates_gpu.cu:68-85: res = cs ? c1 : c0.  Like the reference it uploads cs, c1, c0 from their host
/// members in stream order, returns at once, and res.trlwehost holds the result after Synchronize() / StreamQuery(st);
/// operands that are results of earlier recorded operations are picked up by the scheduler's dependence tracking (no
/// global synchronisation).  gCMUXNTT: the same on device buffers only.
inline void CMUXNTT(cuFHETRLWElvl1& res, cuFHETRGSWNTTlvl1& cs, cuFHETRLWElvl1& c1, cuFHETRLWElvl1& c0, Stream st)
{
    CUFHE_AMD_CHECK(cufhe_amd_enqueue_cmux(st.device_id(), st.raw(), 1, res.handle, cs.handle, c1.handle, c0.handle));
}
inline void gCMUXNTT(cuFHETRLWElvl1& res, cuFHETRGSWNTTlvl1& cs, cuFHETRLWElvl1& c1, cuFHETRLWElvl1& c0, Stream st)
{
    CUFHE_AMD_CHECK(cufhe_amd_enqueue_cmux(st.device_id(), st.raw(), 0, res.handle, cs.handle, c1.handle, c0.handle));
}
#endif  // CUFHE_AMD_SMALL_NTT_MODULUS

#undef CUFHE_AMD_GATE1
#undef CUFHE_AMD_GATE2
#undef CUFHE_AMD_GATE3

}  // namespace cufhe
