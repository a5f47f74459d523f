// The lvl20 key switch for large launches (kept apart from kernels_lvl2.hip.h, whose hash stamps the profiled
// blind rotation in profiles/kernel_facts.json).
#pragma once
#include "kernels_lvl2.hip.h"

namespace cufhe_amd {

// ----------------------------------------------------------------------------------
// The lvl20 key switch with the table shared through LDS: keyswitch_kernel (kernels.hip.h) over the lvl20 shape -- 2048 values
// a'_j of 64 bits per ciphertext, t = 7 digits each, 14 candidate rows of 640 words per j (35 KiB contiguous in the padded layout).
// A workgroup keeps the digit words of at most 1024 steps, so a launch always cuts j into at least two runs (KsDims::min_slices).
// L2 traffic drops 16x against keyswitch_lvl2_kernel (a workgroup per ciphertext, rows straight from L2).
// ----------------------------------------------------------------------------------
struct KsShapeLvl2 {
    using Desc = LinDesc64;
    static constexpr int kn = k2N, t = k2KsT, row_pad = kKsRowPad, n_out = kLvl0N;
    // iksoffsetgen<lvl20> + roundoffset (include/keyswitch_gpu.cuh:13-23,92-98); only the top t*basebit = 14 bits carry digits
    static constexpr uint64_t koff()
    {
        uint64_t o = 1ull << (64 - (1 + k2KsBasebit * k2KsT));
        for (int i = 1; i <= k2KsT; i++) o += ((1ull << k2KsBasebit) / 2) << (64 - i * k2KsBasebit);
        return o;
    }
    static __device__ __forceinline__ uint32_t digit_word(const Desc& d, int j)
    {
        const uint64_t v = (uint64_t)(int64_t)d.ca * d.in0[j] + (uint64_t)(int64_t)d.cb * d.in1[j];
        return (uint32_t)((v + koff()) >> 48);
    }
    static __device__ __forceinline__ uint32_t bprime(const Desc& d)
    {
        const uint64_t v = (uint64_t)(int64_t)d.ca * d.in0[kn] + (uint64_t)(int64_t)d.cb * d.in1[kn];
        return (uint32_t)((v + d.off + (1ull << 31)) >> 32);           // rounding narrowing, :100-101
    }
};
static_assert(k2KsBasebit == kKsBasebit && k2KsNumBase == kKsNumBase, "keyswitch_kernel decodes digits of two bits");
static_assert(KsDims<KsShapeLvl2>::step_bytes == k2KsStepRows * kKsRowPad * 4 && KsDims<KsShapeLvl2>::min_slices == 2, "lvl20 shape");

}  // namespace cufhe_amd
