// The lvl20 key switch for large launches (kept apart from kernels_lvl2.hip.h, whose hash stamps the profiled
// blind rotation in profiles/kernel_facts.json).
#pragma once
#include "kernels_lvl2.hip.h"

namespace cufhe_amd {

// ----------------------------------------------------------------------------------
// The same key switch for large launches, the table shared through LDS: the design of keyswitch_kernel
// (kernels.hip.h) at lvl20 sizes.  A workgroup of 16 waves handles 16 ciphertexts and walks j in
// lock-step; the 14 candidate rows of one j (t = 7 levels x 2 values, 35 KiB contiguous in the padded
// layout) are copied ONCE into LDS by LDS-DMA two steps ahead (3 buffers) and every wave adds or
// subtracts the rows its own digits select.  L2 traffic drops 16x against keyswitch_lvl2_kernel.
// The digit words of one ciphertext (2048 x u16) are kept for half of j at a time, so that the
// three step buffers still fit: 32 KiB + 3 x 35 KiB.
// ----------------------------------------------------------------------------------
constexpr int k2KsStepBytes = k2KsStepRows * kKsRowPad * 4;          // 35840
constexpr int k2KsStepPieces = k2KsStepBytes / 1024;                 // 35 DMA pieces of 1 KiB: waves 0-2 move 3, the others 2
constexpr int k2KsWaves3 = k2KsStepPieces - 2 * kKsWaves;            // 3
constexpr int k2KsLdsDigits = kKsWaves * kN * 2;                     // 32768: digit words of 1024 values of j per wave
constexpr int k2KsLdsBytes = k2KsLdsDigits + kKsBuffers * k2KsStepBytes;   // 140288
static_assert(k2KsStepBytes % 1024 == 0 && k2KsWaves3 >= 0 && k2KsWaves3 <= kKsWaves, "DMA piece split");
static_assert(k2N == 2 * kN, "two passes of 1024 digit words");

__global__ __launch_bounds__(kKsThreads) void keyswitch_lvl2_shared_kernel(
    const LinDesc64* __restrict__ descs, int count, const uint32_t* __restrict__ ksk_padded)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int g = blockIdx.x * kKsWaves + wave;
    const bool live = g < count;
    if (!live) g = count - 1;
    const LinDesc64 d = descs[g];
    uint16_t* dig = (uint16_t*)smem + wave * kN;
    char* bufs = smem + k2KsLdsDigits;

    auto issue = [&](int j) {
        if (j >= k2N) return;
        const char* src = (const char*)ksk_padded + (size_t)j * k2KsStepBytes + lane * 16;
        char* dst = bufs + (j % kKsBuffers) * k2KsStepBytes;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int piece = wave < k2KsWaves3 ? 3 * wave + c : 3 * k2KsWaves3 + 2 * (wave - k2KsWaves3) + c;
            if (c == 2 && wave >= k2KsWaves3) break;
            lds_dma16(src + piece * 1024, dst + piece * 1024);
        }
    };
    issue(0);
    issue(1);

    // iksoffsetgen<lvl20> + roundoffset (include/keyswitch_gpu.cuh:13-23,92-98); only the top t*basebit = 14 bits carry digits
    uint64_t koff = 1ull << (64 - (1 + k2KsBasebit * k2KsT));
    for (int i = 1; i <= k2KsT; i++) koff += ((1ull << k2KsBasebit) / 2) << (64 - i * k2KsBasebit);
    auto digits = [&](int pass) {       // this wave's digit words of j in [1024 pass, 1024 pass + 1024)
        for (int jj = lane; jj < kN; jj += 64) {
            const int j = pass * kN + jj;
            const uint64_t v = (uint64_t)(int64_t)d.ca * d.in0[j] + (uint64_t)(int64_t)d.cb * d.in1[j];
            dig[jj] = (uint16_t)((v + koff) >> 48);
        }
    };
    digits(0);
    uint32_t bprime = 0;
    if (lane == 0) {
        const uint64_t v = (uint64_t)(int64_t)d.ca * d.in0[k2N] + (uint64_t)(int64_t)d.cb * d.in1[k2N];
        bprime = (uint32_t)((v + d.off + (1ull << 31)) >> 32);          // rounding narrowing, :100-101
    }
    bprime = __builtin_amdgcn_readlane(bprime, 0);

    uint4 res[kKsPieces];
#pragma unroll
    for (int m = 0; m < kKsPieces; m++) res[m] = make_uint4(0, 0, 0, 0);
    if (lane == 29) res[2].z = bprime;           // word 630 = 4 * (29 + 128) + 2 starts from b'
    int off[kKsPieces];
    off[0] = lane * 16; off[1] = (lane + 64) * 16; off[2] = (lane < 32 ? lane + 128 : 159) * 16;
    const char* pbase[kKsBuffers][kKsPieces];
#pragma unroll
    for (int bi = 0; bi < kKsBuffers; bi++)
#pragma unroll
        for (int m = 0; m < kKsPieces; m++) pbase[bi][m] = smem + opaque(k2KsLdsDigits + bi * k2KsStepBytes + off[m]);

    __syncthreads();          // digit words visible; the prologue's plain loads have drained vmcnt
    auto step = [&](int j, int jj, const char* const (&pb)[kKsPieces]) {
        // the pieces of step j+1 (this wave's newest 3 or 2 DMAs) stay in flight across the barrier, only step j must
        // have landed; lgkmcnt(0): this wave has finished reading step j-1, whose buffer step j+2 overwrites
        if (j + 1 >= k2N) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (wave < k2KsWaves3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        issue(j + 2);
        const uint32_t dj = __builtin_amdgcn_readfirstlane((uint32_t)dig[jj]);
#pragma unroll
        for (int k = 0; k < k2KsT; k++) {
            // field f = val + 2: 0 -> +row(v=2), 1 -> +row(v=1), 2 -> nothing, 3 -> -row(v=1)
            const uint32_t f = (dj >> (16 - (k + 1) * k2KsBasebit)) & ((1u << k2KsBasebit) - 1);
            if (f != 2) {
                const int roff = (k * k2KsNumBase + (f == 0 ? 1 : 0)) * (kKsRowPad * 4);
                uint4 r[kKsPieces];
#pragma unroll
                for (int m = 0; m < kKsPieces; m++) r[m] = *(const uint4*)(pb[m] + roff);
                if (f == 3) {
#pragma unroll
                    for (int m = 0; m < kKsPieces; m++) { res[m].x -= r[m].x; res[m].y -= r[m].y; res[m].z -= r[m].z; res[m].w -= r[m].w; }
                } else {
#pragma unroll
                    for (int m = 0; m < kKsPieces; m++) { res[m].x += r[m].x; res[m].y += r[m].y; res[m].z += r[m].z; res[m].w += r[m].w; }
                }
            }
        }
    };
    static_assert(kKsBuffers == 3, "the j loops are unrolled by the number of buffers");
    // pass 0: j = 0 .. 1023 (1024 = 3 * 341 + 1), buffers 0 1 2 0 1 2 ... 0
#pragma unroll 1
    for (int j = 0; j + 2 < kN; j += 3) {
        step(j, j, pbase[0]);
        step(j + 1, j + 1, pbase[1]);
        step(j + 2, j + 2, pbase[2]);
    }
    step(kN - 1, kN - 1, pbase[(kN - 1) % kKsBuffers]);
    digits(1);                // each wave reads only its own digit words: no barrier needed around the refill
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // pass 1: j = 1024 .. 2047, buffers 1 2 0 1 2 0 ... (1024 % 3 = 1), then 2047 % 3 = 1
#pragma unroll 1
    for (int j = kN; j + 2 < k2N; j += 3) {
        step(j, j - kN, pbase[1]);
        step(j + 1, j + 1 - kN, pbase[2]);
        step(j + 2, j + 2 - kN, pbase[0]);
    }
    step(k2N - 1, kN - 1, pbase[(k2N - 1) % kKsBuffers]);
    if (!live) return;
#pragma unroll
    for (int m = 0; m < kKsPieces; m++) {
        if (m == 2 && lane >= 32) break;
        const int i = off[m] / 4;
        if (i + 0 <= kLvl0N) d.out[i + 0] = res[m].x;
        if (i + 1 <= kLvl0N) d.out[i + 1] = res[m].y;
        if (i + 2 <= kLvl0N) d.out[i + 2] = res[m].z;
        if (i + 3 <= kLvl0N) d.out[i + 3] = res[m].w;
    }
}

}  // namespace cufhe_amd
