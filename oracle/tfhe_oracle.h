/*
 * tfhe_oracle.h -- CPU restatement of the cuFHE gate-bootstrapping path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it, and there only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" at ciphertext-word level.  The reference
 * (virtualsecureplatform/cuFHE) holds no golden vectors, no KATs and no fixed
 * seeds for this path, its CUDA sources cannot be built
 * here (nvcc absent) and its CPU crypto (TFHEpp, an un-vendored submodule with
 * an unrecorded pin: .gitmodules:1-3) is absent from /root/reference.  What IS
 * pinned, by tests/test_oracle.py:
 *   - decrypt(gate(ct...)) == the reference's own truth tables, compiled from
 *     /root/reference/test/plain.h into oracle/_ref/ (test/plain.h:10-69, the
 *     check of test/test_util.h:75-94);
 *   - NTT product == schoolbook negacyclic product mod 2^32, the check of
 *     test/test_polynomial_mult_1024.cu:51-73,209-223 (exactly, not "diff<=2");
 *   - the reference's published NTT constants (prime, psi, Barrett mu,
 *     include/ntt_gpu/ntt_gpuntt.cuh:36-40, src/ntt_gpu/ntt_gpuntt.cu:31-32).
 * Every function below cites the reference file:line it restates.
 *
 * All torus arithmetic is mod 2^32 (uint32_t wrap-around).  Parameter set:
 * SURVEY.md appendix C (n=630, N=1024, k=1, l=3, Bgbit=6, t=8, basebit=2,
 * mu = 2^29, alpha0 = 2^-15, alpha1 = 2^-25).
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Parameter set: compile-time, like the reference's TFHEpp macros (CMakeLists.txt:8-24 ->
 * add_compile_definitions -> TFHEpp params.hpp).  The default is the set BASELINE.json names; the
 * Makefile also builds liboracle_<set>.so for the alternative sets of cufhe_amd/csrc/kernels_ps.hip.h:
 *   -DORC_SET_K2N512   n = 630, N = 512, k = 2, l = 3, Bgbit = 6, t = 8, basebit = 2     (the k = 2 / N = 512
 *                      shape of src/bootstrap_gpu.cu:413-416 and include/ntt_gpu/ntt_gpuntt.cuh:283-329)
 *   -DORC_SET_CGGI16   n = 500, N = 1024, k = 1, l = 2, Bgbit = 10, t = 8, basebit = 2   (the original
 *                      TFHE 80-bit set, what -DUSE_80BIT_SECURITY selects)
 *   -DORC_SET_SMALLMOD the default numbers through the reference's small-modulus NTT (-DUSE_SMALL_NTT_MODULUS)
 * The numeric parameters of TFHEpp's headers are not in the reference tree: these sets are
 * defined HERE, and parity for them means oracle == GPU on the same numbers. */
#if defined(ORC_SET_K2N512)
#define ORC_SET_NAME "k2n512"
#define ORC_n 630
#define ORC_NBIT 9
#define ORC_K 2
#define ORC_L 3
#define ORC_BGBIT 6
#define ORC_T 8
#define ORC_BASEBIT 2
#define ORC_ALPHA0 (1.0 / 32768.0)
#define ORC_ALPHA1 (1.0 / 33554432.0)
#elif defined(ORC_SET_CGGI16)
#define ORC_SET_NAME "cggi16"
#define ORC_n 500
#define ORC_NBIT 10
#define ORC_K 1
#define ORC_L 2
#define ORC_BGBIT 10
#define ORC_T 8
#define ORC_BASEBIT 2
#define ORC_ALPHA0 2.44e-5
#define ORC_ALPHA1 3.73e-9
#elif defined(ORC_SET_SMALLMOD)
/* the BASELINE set computed the way -DUSE_SMALL_NTT_MODULUS builds the reference (CMakeLists.txt:12,26-28): NTT modulus
 * P = 625 * 2^20 + 1, the bootstrapping key and every CMux increment switched between the 2^32 and the P discretisation of the
 * torus (include/ntt_gpu/ntt_small_modulus.cuh).  Approximate by design; oracle == GPU word for word all the same. */
#define ORC_SET_NAME "smallmod"
#define ORC_SMALL_NTT_MODULUS 1
#define ORC_n 630
#define ORC_NBIT 10
#define ORC_K 1
#define ORC_L 3
#define ORC_BGBIT 6
#define ORC_T 8
#define ORC_BASEBIT 2
#define ORC_ALPHA0 (1.0 / 32768.0)
#define ORC_ALPHA1 (1.0 / 33554432.0)
#else
#define ORC_SET_NAME "default"
#define ORC_SET_DEFAULT 1
#define ORC_n 630          /* lvl0 dimension                      */
#define ORC_NBIT 10
#define ORC_K 1            /* TRLWE mask polynomials              */
#define ORC_L 3            /* gadget levels                       */
#define ORC_BGBIT 6
#define ORC_T 8            /* key-switch levels                   */
#define ORC_BASEBIT 2
#define ORC_ALPHA0 (1.0 / 32768.0)      /* 2^-15 */
#define ORC_ALPHA1 (1.0 / 33554432.0)   /* 2^-25 */
#endif
#define ORC_N (1 << ORC_NBIT)           /* lvl1 polynomial degree */
#define ORC_KS_NUMBASE (1 << (ORC_BASEBIT - 1))
#define ORC_LVL0_WORDS (ORC_n + 1)
#define ORC_LVL1_WORDS (ORC_K * ORC_N + 1)
#define ORC_TRLWE_WORDS ((ORC_K + 1) * ORC_N)
#define ORC_BK_ROWS ((ORC_K + 1) * ORC_L)
#define ORC_MU ((uint32_t)1u << 29)
#define ORC_BK_WORDS ((size_t)ORC_n * ORC_BK_ROWS * (ORC_K + 1) * ORC_N)
#define ORC_KSK_WORDS ((size_t)ORC_K * ORC_N * ORC_T * ORC_KS_NUMBASE * ORC_LVL0_WORDS)

/* gate op-codes; (ca, cb, offset) table of src/bootstrap_gpu.cu:424-512 */
enum orc_op {
    ORC_NAND = 0, ORC_NOR, ORC_XNOR, ORC_AND, ORC_OR, ORC_XOR,
    ORC_ANDNY, ORC_ANDYN, ORC_ORNY, ORC_ORYN,
    ORC_MUX, ORC_NMUX, ORC_NOT, ORC_COPY, ORC_NUM_OPS
};

/* ---- deterministic PRNG (xoshiro256**, seeded through splitmix64) ---- */
typedef struct { uint64_t s[4]; } orc_rng;
void orc_rng_seed(orc_rng* r, uint64_t seed);
uint64_t orc_rng_next(orc_rng* r);

/* ---- keys ---- */
/* binary secret keys; s0[ORC_n], s1[ORC_K * ORC_N] hold 0/1 */
void orc_keygen(uint64_t seed, uint32_t* s0, uint32_t* s1);
/* bk[i][row][comp][N]: TRGSW_{s1}(s0[i]); layout of src/bootstrap_gpu.cu:43-49 */
void orc_bkgen(uint64_t seed, const uint32_t* s0, const uint32_t* s1, uint32_t* bk);
/* ksk[j][kappa][v-1][n+1]; layout of include/keyswitch_gpu.cuh:123-126 */
void orc_kskgen(uint64_t seed, const uint32_t* s0, const uint32_t* s1, uint32_t* ksk);

/* ---- TLWE encrypt / decrypt (level 0: dim ORC_n, level 1: dim ORC_N) ---- */
void orc_tlwe_encrypt(orc_rng* r, int level, const uint32_t* key, int bit, uint32_t* ct);
int orc_tlwe_decrypt(int level, const uint32_t* key, const uint32_t* ct);
uint32_t orc_tlwe_phase(int level, const uint32_t* key, const uint32_t* ct);
/* batch helpers (bits[count] in, cts[count][words] out), seeded */
void orc_tlwe_encrypt_batch(uint64_t seed, int level, const uint32_t* key,
                            const uint8_t* bits, size_t count, uint32_t* cts);
void orc_tlwe_decrypt_batch(int level, const uint32_t* key, const uint32_t* cts,
                            size_t count, uint8_t* bits);

/* ---- polynomial products (negacyclic, mod 2^32) ---- */
/* res = a (signed small) * b (torus), schoolbook: test/test_polynomial_mult_1024.cu:51-73 */
void orc_polymul_schoolbook(uint32_t* res, const int32_t* a, const uint32_t* b);
/* same product through the reference's NTT prime 2^60+30721 */
void orc_polymul_ntt(uint32_t* res, const int32_t* a, const uint32_t* b);
/* raw transforms over the reference prime (bit-reversed spectrum) */
void orc_ntt_forward(uint64_t* x);   /* in place, natural -> bit-reversed */
void orc_ntt_inverse(uint64_t* x);   /* in place, bit-reversed -> natural, scaled by N^-1 */
uint64_t orc_ntt_modulus(void);
uint64_t orc_ntt_psi(void);
uint64_t orc_ntt_barrett_mu(void);
uint64_t orc_ntt_n_inverse(void);
uint64_t orc_ntt_mulmod(uint64_t a, uint64_t b);   /* barrett_mult */
/* the two modulus switches of include/ntt_gpu/ntt_small_modulus.cuh:147-177 (every build: plain integer formulas) */
uint32_t orc_smallmod_from_torus(uint32_t torus);   /* torus32_to_ntt_mod: round(a P / 2^32), in [0, P] */
uint32_t orc_smallmod_to_torus(int32_t centred);    /* ntt_mod_to_torus32: round(a 2^32 / P) */
uint32_t orc_smallmod_modulus(void);

/* ---- evaluation key in the NTT domain (opaque) ---- */
typedef struct orc_evalkey orc_evalkey;
orc_evalkey* orc_evalkey_create(const uint32_t* bk, const uint32_t* ksk);
void orc_evalkey_destroy(orc_evalkey* ek);

/* ---- the gate path, piece by piece ---- */
/* acc <- test vector rotated by the (pre-added) lvl0 TLWE, then n CMux steps.
 * steps < 0 means all ORC_n steps (used for partial-progress parity checks). */
void orc_blind_rotate(const orc_evalkey* ek, uint32_t* acc /*[2N]*/,
                      const uint32_t* tlwe0 /*[n+1]*/, int steps);
void orc_sample_extract0(uint32_t* tlwe1 /*[N+1]*/, const uint32_t* acc /*[2N]*/);
void orc_keyswitch(const orc_evalkey* ek, uint32_t* tlwe0 /*[n+1]*/, const uint32_t* tlwe1 /*[N+1]*/);

/* ---- TRLWE-level primitives (src/cufhe_gates_gpu.cu:69-146) ---- */
/* CMUXNTT (src/bootstrap_gpu.cu:162-285): res = c0 + trgsw [x] (c1 - c0); trgsw is a torus-domain
 * TRGSW [(k+1)l][k+1][N], trlwe operands are [(k+1)N] words */
void orc_cmux(uint32_t* res, const uint32_t* trgsw, const uint32_t* c1, const uint32_t* c0);
/* SEIandKS (src/keyswitch_gpu.cu:26-40): sample extract at 0, then key switch */
void orc_sample_extract_keyswitch(const orc_evalkey* ek, uint32_t* tlwe0, const uint32_t* trlwe);
/* Refresh / SEIandBootstrap2TRLWE (src/bootstrap_gpu.cu:325-364) with the test vector taken
 * from the key-switched ciphertext (the reference reads an uninitialised buffer there) */
void orc_refresh(const orc_evalkey* ek, uint32_t* trlwe_out, const uint32_t* trlwe_in);

/* whole gates.  level 0: ctxts are lvl0 TLWEs (blind rotate -> key switch);
 * level 1: ctxts are lvl1 TLWEs (key switch -> blind rotate).
 * in2 is only read by MUX/NMUX (out = in0 ? in1 : in2), in1 unused for NOT/COPY. */
void orc_gate(const orc_evalkey* ek, int op, int level, uint32_t* out,
              const uint32_t* in0, const uint32_t* in1, const uint32_t* in2);
/* count gates; ops[count] (or a single op if ops_stride==0); OpenMP over gates */
void orc_gate_batch(const orc_evalkey* ek, const int32_t* ops, int ops_stride, int level,
                    size_t count, uint32_t* out, const uint32_t* in0,
                    const uint32_t* in1, const uint32_t* in2, int threads);
int orc_max_threads(void);
/* the compiled parameter set: n, Nbit, k, l, Bgbit, t, basebit (7 ints); returns its name */
const char* orc_get_params(int* out7);

/* plaintext truth table of op (own restatement; cross-checked against
 * oracle/_ref built from test/plain.h) */
int orc_truth(int op, int a, int b, int c);
/* gate linear part: ca, cb, offset (in units of mu) */
void orc_gate_coeffs(int op, int* ca, int* cb, int* off_mu);

#ifdef __cplusplus
}
#endif
#endif
