// kernels_common.hip.h -- parameters, descriptor and LDS maps shared by the kernel files (kernels.hip.h and the
// separately compiled kernels_ll.hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ntt_wave.h"

namespace cufhe_amd {

// TFHE parameters.  Another set is NOT just a change of these lines: the
// static_asserts below tie l, Bgbit and N to the FP64 prime (exactness of the external product) and
// to the lazy-reduction schedule of ntt_wave.h; a set that fails them needs the limb split of
// kernels_lvl2.hip.h.
constexpr int kLvl0N = 630;            // lvl0param::n
constexpr int kNbit = 10;              // lvl1param::nbit
constexpr int kL = 3;                  // lvl1param::l
constexpr int kBgbit = 6;              // lvl1param::Bgbit
constexpr int kKsT = 8;                // lvl10param::t
constexpr int kKsBasebit = 2;          // lvl10param::basebit
constexpr uint32_t kMu = 1u << 29;     // lvl0/lvl1 mu
constexpr int kLvl0Words = kLvl0N + 1;
constexpr int kLvl1Words = kN + 1;
constexpr int kBkRows = 2 * kL;                         // (k+1) l
constexpr int kBkPolysPerStep = kBkRows * 2;            // (k+1)^2 l = 12
constexpr size_t kBkStepDoubles = (size_t)kBkPolysPerStep * kN;   // 12288 doubles = 98304 B
constexpr int kKsRowWords = kLvl0Words;                 // 631
constexpr int kKsNumBase = 1 << (kKsBasebit - 1);       // 2

// Exactness of the external product over the FP64 prime (fpfield.h): the true integer sum of one
// CMux output coefficient, |sum| <= (k+1) l N (Bg/2) 2^31 with the key read as signed words, must
// stay below p/2 or the centred lift returns a wrong torus word.
constexpr double kExtProdSumBound = 2.0 * kL * kN * (double)(1u << (kBgbit - 1)) * 2147483648.0;
static_assert(kExtProdSumBound < fpf::P / 2, "(k+1) l N (Bg/2) 2^31 >= p/2: this parameter set needs the limb split (kernels_lvl2.hip.h)");
// ... and of the lazy-reduction schedule: digits of magnitude Bg/2 through the forward transform,
// then (k+1) l unreduced wide products per accumulator
constexpr double kDigitSpectrumBound = forward_digit_spectrum_bound((double)(1u << (kBgbit - 1)));
static_assert(kDigitSpectrumBound > 0, "forward NTT of gadget digits: a stage input exceeds its multiplication's range");
static_assert(pointwise_sum_fits(kDigitSpectrumBound, kBkRows), "(k+1) l unreduced pointwise products exceed 2^53");
static_assert(kL * kBgbit <= 32 - 1, "decomposition wider than the torus word");

// out = ca * in0 + cb * in1 + (0, ..., 0, off): the linear part of every gate
struct LinDesc {
    const uint32_t* in0;
    const uint32_t* in1;   // never null (equal to in0 when cb == 0)
    uint32_t* out;
    int32_t ca, cb;
    uint32_t off;
    uint32_t pad;
};

// Device fault word (one uint32 per device in host-visible memory, DeviceState::fault): bits a kernel sets when it
// detects that its own result cannot be trusted.  The host turns any set bit into status -5 (capi.hip: device_fault).
constexpr uint32_t kFaultLl2SyncTimeout = 1u;      // blind_rotate_ll2_kernel: the inverse waves' LDS rendezvous timed out

// NTT-only kernels (key conversion, product check): 4 waves per workgroup
constexpr int kNttWavesPerBlock = 4;
constexpr int kNttThreads = 64 * kNttWavesPerBlock;
constexpr int kNttLdsBytes = kLdsTableBytes + kNttWavesPerBlock * kTileBytes;   // 49920

// Blind rotate: ONE 8-wave workgroup per CU (2 waves per SIMD).  Its LDS holds the twiddle
// tables, one transpose tile per wave, the abar list of every wave and three 16 KiB
// buffers for the TRGSW row shared by the 8 waves (the BK tile staged in LDS).
constexpr int kBrWavesPerBlock = 8;
constexpr int kBrThreads = 64 * kBrWavesPerBlock;                               // 512
constexpr int kBkRowBytes = 2 * kN * 8;                                         // 16384: one TRGSW row (2 polys)
constexpr int kAbarBytes = 1280;                                                // 630 x u16, padded
constexpr int kBkRowBuffers = 3;
// LDS map: [row buffers][twiddle tables][tiles][abar lists].  The row buffers come first so
// that "buffer + piece" offsets (< 64 KiB) fold into the DS instructions' offset field.
constexpr int kBrLdsBk = 0;
constexpr int kBrLdsTables = kBrLdsBk + kBkRowBuffers * kBkRowBytes;            // 49152
constexpr int kBrLdsTiles = kBrLdsTables + kLdsTablePackedBytes;                // the packed r4 tables (ntt_wave.h)
constexpr int kBrLdsAbar = kBrLdsTiles + kBrWavesPerBlock * kTileBytes;
constexpr int kBrLdsBytes = kBrLdsAbar + kBrWavesPerBlock * kAbarBytes;         // 145920

// gadget decomposition constants, include/gatebootstrapping_gpu.cuh:18-27,145-150
__host__ __device__ constexpr uint32_t decomp_offset()
{
    uint32_t o = 0;
    for (int i = 1; i <= kL; i++) o += (1u << (kBgbit - 1)) << (32 - i * kBgbit);
    return o + (1u << (32 - kL * kBgbit - 1));     // + roundoffset
}
// digit_d = field_d(t) - Bg/2 = sign-extended field_d(t ^ mask): flipping the top bit of each
// Bgbit-wide field adds Bg/2 modulo Bg without carrying into the next digit
__host__ __device__ constexpr uint32_t decomp_signmask()
{
    uint32_t m = 0;
    for (int i = 1; i <= kL; i++) m |= (1u << (kBgbit - 1)) << (32 - i * kBgbit);
    return m;
}

}  // namespace cufhe_amd
