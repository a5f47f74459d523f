/*
 * tfhe_oracle_lvl2.h -- CPU oracle for the N = 2048 / 64-bit-torus gate path
 * (BASELINE.json configs[4], "lvl2").  TEST INFRASTRUCTURE ONLY, same rules as
 * tfhe_oracle.h.
 *
 * PARITY STATUS: "parity unpinned" -- and here there is nothing to pin against:
 * the reference has NO N = 2048 code path (its NTT dispatch is
 * `if constexpr (N == 1024) ... else if (N == 512)`, its prime 2^60+30721 has
 * 2-adicity 11) and no 64-bit accumulate.  This oracle restates
 * what the reference's gate templates compute when instantiated at
 * brP = lvl02, iksP = lvl20:
 *   __BlindRotatePreAdd__ / Accumulate   include/gatebootstrapping_gpu.cuh:10-52,115-345
 *   __SampleExtractIndex__<P,0>          src/bootstrap_gpu.cu:366-381
 *   KeySwitchFromTLWE<P>                 include/keyswitch_gpu.cuh:83-134
 *                                        (domain 64-bit, target 32-bit: the
 *                                        rounding narrowing of :100-101)
 * with the 64-bit constants the templates would need (the reference types
 * decomp_offset as uint32_t, gatebootstrapping_gpu.cuh:147; here it is 64 bit).
 * The external product is exact integer arithmetic mod 2^64.  It is computed
 * over Goldilocks 2^64 - 2^32 + 1 (the field of the reference's legacy path,
 * include/ntt_gpu/ntt_ffp.cuh, which does have 4096-th roots) on the two
 * 32-bit halves of every key word, |sum| <= 8*2048*256*2^32 = 2^54 < p/2, and
 * pinned by tests/test_oracle_lvl2.py against a schoolbook product mod 2^64 and
 * against decrypt == truth table.  The device code uses a different field and a
 * different split (one FP64 prime, three 22-bit limbs), so agreement between
 * the two is not an artefact of shared code.
 *
 * Parameter set (frozen here):
 *   lvl2 : N = 2048, k = 1, l = 4, Bgbit = 9, T = uint64, mu = 2^61, alpha = 2^-44
 *   lvl02: blind rotate from a lvl0 TLWE (n = 630, uint32) into the lvl2 ring
 *   lvl20: key switch lvl2 -> lvl0, t = 7, basebit = 2
 */
#ifndef TFHE_ORACLE_LVL2_H
#define TFHE_ORACLE_LVL2_H

#include "tfhe_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORC2_NBIT = 11,
    ORC2_N = 2048,
    ORC2_L = 4,
    ORC2_BGBIT = 9,
    ORC2_T = 7,
    ORC2_BASEBIT = 2,
    ORC2_KS_NUMBASE = 1 << (ORC2_BASEBIT - 1),
    ORC2_LVL2_WORDS = ORC2_N + 1,
    ORC2_TRLWE_WORDS = 2 * ORC2_N,
    ORC2_BK_ROWS = 2 * ORC2_L
};
#define ORC2_MU ((uint64_t)1 << 61)
#define ORC2_BK_WORDS ((size_t)ORC_n * ORC2_BK_ROWS * 2 * ORC2_N)                         /* uint64 */
#define ORC2_KSK_WORDS ((size_t)ORC2_N * ORC2_T * ORC2_KS_NUMBASE * ORC_LVL0_WORDS)      /* uint32 */

/* binary lvl2 key s2[ORC2_N] (the lvl0 key comes from orc_keygen) */
void orc2_keygen(uint64_t seed, uint32_t* s2);
/* bk[i][row][comp][N] (uint64): TRGSW_{s2}(s0[i]), row = j*l + d carries s0[i] * 2^(64-(d+1)Bgbit) */
void orc2_bkgen(uint64_t seed, const uint32_t* s0, const uint32_t* s2, uint64_t* bk);
/* ksk[j][kappa][v-1][n+1] (uint32): TLWE_{s0}(v * s2[j] * 2^(32-(kappa+1)basebit)) */
void orc2_kskgen(uint64_t seed, const uint32_t* s0, const uint32_t* s2, uint32_t* ksk);

void orc2_tlwe_encrypt(orc_rng* r, const uint32_t* s2, int bit, uint64_t* ct /*[N+1]*/);
uint64_t orc2_tlwe_phase(const uint32_t* s2, const uint64_t* ct);
int orc2_tlwe_decrypt(const uint32_t* s2, const uint64_t* ct);

/* negacyclic products mod 2^64, a small signed, b torus */
void orc2_polymul_schoolbook(uint64_t* res, const int32_t* a, const uint64_t* b);
void orc2_polymul_ntt(uint64_t* res, const int32_t* a, const uint64_t* b);

typedef struct orc2_evalkey orc2_evalkey;
orc2_evalkey* orc2_evalkey_create(const uint64_t* bk, const uint32_t* ksk);
void orc2_evalkey_destroy(orc2_evalkey* ek);

/* acc[2][N] <- rotated test vector (mu = 2^61), then `steps` CMux steps (< 0: all n) */
void orc2_blind_rotate(const orc2_evalkey* ek, uint64_t* acc, const uint32_t* tlwe0, int steps);
void orc2_sample_extract0(uint64_t* tlwe2 /*[N+1]*/, const uint64_t* acc /*[2N]*/);
void orc2_keyswitch(const orc2_evalkey* ek, uint32_t* tlwe0 /*[n+1]*/, const uint64_t* tlwe2 /*[N+1]*/);

/* gates on lvl0 ciphertexts through the lvl2 ring (blind rotate lvl02 -> key switch lvl20);
 * op codes, operand meaning and the (ca, cb, offset) table are those of orc_gate */
void orc2_gate(const orc2_evalkey* ek, int op, uint32_t* out,
               const uint32_t* in0, const uint32_t* in1, const uint32_t* in2);
void orc2_gate_batch(const orc2_evalkey* ek, const int32_t* ops, int ops_stride, size_t count,
                     uint32_t* out, const uint32_t* in0, const uint32_t* in1, const uint32_t* in2,
                     int threads);

#ifdef __cplusplus
}
#endif
#endif
