/*
 * cpu_fast.c -- an OPTIMISED CPU implementation of the level-0 gate (blind rotate -> sample extract ->
 * key switch), timed by bench.py as `cpu_baseline`.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY, like everything under oracle/: the product never loads it.
 * tfhe_oracle.c is the checker (a line-by-line restatement of the reference: radix-2 NTT over
 * 2^60+30721 with 128-bit Barrett products); as a *baseline* it is a strawman (126 ms per gate per
 * core).  This file computes the same words -- tests/test_oracle.py compares them -- the way one would
 * write the path for a CPU:
 *   - the same exact FP64 field as the GPU kernels (p = 5440^4 + 1, lazy balanced residues, six
 *     FMA-class operations per modular product, cufhe_amd/csrc/fpfield.h) -- on AVX2 / AVX-512
 *     every operation is one vector instruction;
 *   - constant-geometry (Pease) radix-2 transforms: every stage reads the pairs (x[j], x[j+N/2]) and
 *     writes (y[2j], y[2j+1]), so all ten stages are unit-stride vector loops (the in-place
 *     Cooley-Tukey of the reference, include/ntt_gpu/ntt_gpuntt.cuh:232-276, has strides 4, 2, 1 in
 *     its last stages); twiddles are pre-expanded per stage;
 *   - the bootstrapping key kept in the transform domain, scaled by N^-1;
 *   - OpenMP over gates, four gates per thread walking the key together so that a 96 KiB key step
 *     is read from L2 by four rotations;
 *   - function multi-versioning (AVX-512, AVX2+FMA, baseline) resolved at load time: the library is
 *     built in one container and timed on another host.
 * The reference's own CPU context figure is 10 ms per gate for the TFHE library (README.md:29-31).
 *
 * Algorithm (reference lines): pre-add + modswitch include/gatebootstrapping_gpu.cuh:316-345, test
 * vector :29-52, CMux :115-285, sample extract src/bootstrap_gpu.cu:366-381, key switch
 * include/keyswitch_gpu.cuh:83-134.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "tfhe_oracle.h"

#define N ORC_N
#define HALF (N / 2)
#define LOGN 10
#define ROWS ORC_BK_ROWS              /* (k+1) l = 6 */

static const double P = 875781160960001.0;
static const uint64_t P_U64 = 875781160960001ull;
static const uint64_t PSI_2048 = 423584205157050ull;
static const double PINV = 0x1.491cc17c934a8p-50;
static const double MAGIC0 = 6755399441055744.0;     /* 1.5 * 2^52 */
static const double MAGIC1 = 13510798882111488.0;    /* 1.5 * 2^53 */

/* ---- tables ---- */
static double g_fwd[LOGN][HALF], g_inv[LOGN][HALF];   /* per stage, per pair position */
static int g_tables_ready = 0;

static uint64_t mulmod_u(uint64_t a, uint64_t b) { return (uint64_t)((unsigned __int128)a * b % P_U64); }
static uint64_t powmod_u(uint64_t a, uint64_t e)
{
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mulmod_u(r, a);
        a = mulmod_u(a, a);
        e >>= 1;
    }
    return r;
}
static double balanced(uint64_t v) { return v > P_U64 / 2 ? -(double)(P_U64 - v) : (double)v; }
static uint32_t bitrev10(uint32_t x)
{
    uint32_t r = 0;
    for (int i = 0; i < LOGN; i++) r |= ((x >> i) & 1u) << (LOGN - 1 - i);
    return r;
}
static void build_tables(void)
{
    if (g_tables_ready) return;
    const uint64_t psi_inv = powmod_u(PSI_2048, P_U64 - 2);
    for (int s = 0; s < LOGN; s++) {
        const uint32_t m = 1u << s;                      /* groups at stage s; root[m + i] = psi^bitrev(m + i) */
        for (uint32_t q = 0; q < HALF; q++) {
            const uint32_t idx = m + (q & (m - 1));
            g_fwd[s][q] = balanced(powmod_u(PSI_2048, bitrev10(idx)));
            g_inv[s][q] = balanced(powmod_u(psi_inv, bitrev10(idx)));
        }
    }
    g_tables_ready = 1;
}

/* ---- the vector loops, once per instruction set; picked at load time ---- */
#define FN_(name, suf) name##suf
#define FN__(name, suf) FN_(name, suf)
#define FN(name) FN__(name, SUFFIX)

#define SUFFIX _base
#include "cpu_fast_kernels.inc"
#undef SUFFIX
#pragma GCC push_options
#pragma GCC target("avx2,fma")
#define SUFFIX _avx2
#include "cpu_fast_kernels.inc"
#undef SUFFIX
#pragma GCC pop_options
#pragma GCC push_options
#pragma GCC target("avx512f,avx512dq,avx512vl,avx512bw,fma")
#define SUFFIX _avx512
#include "cpu_fast_kernels.inc"
#undef SUFFIX
#pragma GCC pop_options

static void (*ntt_forward)(double* restrict, double* restrict) = ntt_forward_base;
static void (*ntt_inverse)(double* restrict, double* restrict) = ntt_inverse_base;
static void (*pointwise_acc)(double* restrict, double* restrict, const double* restrict, const double* restrict,
                             const double* restrict) = pointwise_acc_base;
static void (*rotate_sub)(uint32_t* restrict, const uint32_t* restrict, uint32_t) = rotate_sub_base;
static void (*digits)(double* restrict, const uint32_t* restrict, int) = digits_base;
static void (*lift_add)(uint32_t* restrict, const double* restrict) = lift_add_base;
static void (*reduce_all)(double* restrict) = reduce_all_base;
static void (*row_add)(uint32_t* restrict, const uint32_t* restrict, int) = row_add_base;
static void (*row_sub)(uint32_t* restrict, const uint32_t* restrict, int) = row_sub_base;
#define mulmod_wide mulmod_wide_base
#define reduce reduce_base

int fast_isa_level(void);
static void pick_isa(void)
{
    const int isa = fast_isa_level();
#define PICK(suf) do { ntt_forward = ntt_forward##suf; ntt_inverse = ntt_inverse##suf; pointwise_acc = pointwise_acc##suf; \
        rotate_sub = rotate_sub##suf; digits = digits##suf; lift_add = lift_add##suf; reduce_all = reduce_all##suf; row_add = row_add##suf; row_sub = row_sub##suf; } while (0)
    if (isa == 4) PICK(_avx512);
    else if (isa == 3) PICK(_avx2);
#undef PICK
}

struct fast_evalkey {
    double* bk;           /* [n][row][out][N], transform domain, scaled by N^-1 */
    const uint32_t* ksk;  /* caller's array, must outlive the key */
};
typedef struct fast_evalkey fast_evalkey;

fast_evalkey* fast_evalkey_create(const uint32_t* bk, const uint32_t* ksk)
{
    build_tables();
    pick_isa();
    fast_evalkey* ek = (fast_evalkey*)malloc(sizeof(*ek));
    const size_t polys = (size_t)ORC_n * ROWS * 2;
    ek->bk = (double*)aligned_alloc(64, polys * N * sizeof(double));
    ek->ksk = ksk;
    const double ninv = balanced(powmod_u(N, P_U64 - 2));
#pragma omp parallel
    {
        double* x = (double*)aligned_alloc(64, N * 8);
        double* y = (double*)aligned_alloc(64, N * 8);
#pragma omp for schedule(static)
        for (size_t p = 0; p < polys; p++) {
            /* 32-bit words in: reduce every stage (the digit schedule assumes |x| <= 32) */
            for (int e = 0; e < N; e++) x[e] = (double)(int32_t)bk[p * N + e];
            double *a = x, *b = y;
            for (int s = 0; s < LOGN; s++) {
                for (int j = 0; j < HALF; j++) {
                    const double u = a[j], t = mulmod_wide(reduce(a[j + HALF]), g_fwd[s][j]);
                    b[2 * j] = reduce(u + t);
                    b[2 * j + 1] = reduce(u - t);
                }
                double* tmp = a; a = b; b = tmp;
            }
            for (int e = 0; e < N; e++) ek->bk[p * N + e] = reduce(mulmod_wide(a[e], ninv));
        }
        free(x);
        free(y);
    }
    return ek;
}
void fast_evalkey_destroy(fast_evalkey* ek)
{
    if (!ek) return;
    free(ek->bk);
    free(ek);
}

#define BLOCK 16    /* gates per thread walking the keys together: a 96 KiB bootstrapping-key step and the 40 KiB
                       of key-switching rows of one j are fetched once per 16 gates */

typedef struct {
    uint32_t acc[BLOCK][2][N];
    uint32_t abar[BLOCK][ORC_n];
    uint32_t t1[BLOCK][N + 1];
    uint32_t res[BLOCK][ORC_LVL0_WORDS + 9];
    uint32_t tmp[N];
    double A0[N], A1[N], X[N], Y[N];
} block_ws;

static void gate_block(const fast_evalkey* ek, int nb, const int32_t* ca, const int32_t* cb, const uint32_t* off,
                       uint32_t* const* out, const uint32_t* const* in0, const uint32_t* const* in1, block_ws* w)
{
    for (int g = 0; g < nb; g++) {
        uint32_t b = off[g];
        for (int i = 0; i < ORC_n; i++) {
            const uint32_t c = (uint32_t)ca[g] * in0[g][i] + (uint32_t)cb[g] * in1[g][i];
            w->abar[g][i] = (c + (1u << (32 - 2 - LOGN))) >> (32 - 1 - LOGN);
        }
        b += (uint32_t)ca[g] * in0[g][ORC_n] + (uint32_t)cb[g] * in1[g][ORC_n];
        const uint32_t bbar = 2 * N - (b >> (32 - 1 - LOGN));
        for (uint32_t e = 0; e < N; e++) {
            w->acc[g][0][e] = 0;
            const int neg = (bbar != 2 * N) && ((e < (bbar & (N - 1))) != ((bbar >> LOGN) != 0));
            w->acc[g][1][e] = neg ? 0u - ORC_MU : ORC_MU;
        }
    }
    for (int i = 0; i < ORC_n; i++) {
        const double* key = ek->bk + (size_t)i * ROWS * 2 * N;
        for (int g = 0; g < nb; g++) {
            if (w->abar[g][i] == 0) continue;        /* (X^0 - 1) acc = 0: nothing to add */
            memset(w->A0, 0, sizeof(w->A0));
            memset(w->A1, 0, sizeof(w->A1));
            for (int j = 0; j < 2; j++) {
                rotate_sub(w->tmp, w->acc[g][j], w->abar[g][i]);
                for (int d = 0; d < ORC_L; d++) {
                    digits(w->X, w->tmp, d);
                    ntt_forward(w->X, w->Y);
                    const double* row = key + (size_t)(j * ORC_L + d) * 2 * N;
                    pointwise_acc(w->A0, w->A1, w->X, row, row + N);
                }
            }
            for (int o = 0; o < 2; o++) {
                double* A = o ? w->A1 : w->A0;
                reduce_all(A);
                ntt_inverse(A, w->Y);
                lift_add(w->acc[g][o], A);
            }
        }
    }
    /* sample extract at 0, then the key switch with j outermost: the 16 candidate rows of one j serve the whole block */
    uint32_t koff = 1u << (32 - (1 + ORC_BASEBIT * ORC_T));
    for (int k = 1; k <= ORC_T; k++) koff += (1u << (ORC_BASEBIT - 1)) << (32 - k * ORC_BASEBIT);
    for (int g = 0; g < nb; g++) {
        uint32_t* t1 = w->t1[g];
        t1[0] = w->acc[g][0][0];
        for (int m = 1; m < N; m++) t1[m] = 0u - w->acc[g][0][N - m];
        t1[N] = w->acc[g][1][0];
        memset(w->res[g], 0, sizeof(w->res[g]));
        w->res[g][ORC_n] = t1[N];
    }
    for (int j = 0; j < N; j++) {
        const uint32_t* rows = ek->ksk + (size_t)j * ORC_T * ORC_KS_NUMBASE * ORC_LVL0_WORDS;
        for (int g = 0; g < nb; g++) {
            const uint32_t a = w->t1[g][j] + koff;
            for (int k = 0; k < ORC_T; k++) {
                const int val = (int)((a >> (32 - (k + 1) * ORC_BASEBIT)) & ((1u << ORC_BASEBIT) - 1)) - (1 << (ORC_BASEBIT - 1));
                if (val == 0) continue;
                const int v = val > 0 ? val : -val;
                const uint32_t* row = rows + ((size_t)k * ORC_KS_NUMBASE + (v - 1)) * ORC_LVL0_WORDS;
                if (val > 0) row_sub(w->res[g], row, ORC_LVL0_WORDS);
                else row_add(w->res[g], row, ORC_LVL0_WORDS);
            }
        }
    }
    for (int g = 0; g < nb; g++) memcpy(out[g], w->res[g], ORC_LVL0_WORDS * sizeof(uint32_t));
}

/* `count` two-input gates on lvl0 ciphertexts (NAND ... ORYN); same contract as orc_gate_batch at level 0 */
int fast_gate_batch(const fast_evalkey* ek, const int32_t* ops, int ops_stride, size_t count, uint32_t* out,
                    const uint32_t* in0, const uint32_t* in1, int threads)
{
    for (size_t g = 0; g < count; g++)
        if (ops[g * (size_t)ops_stride] < 0 || ops[g * (size_t)ops_stride] >= ORC_MUX) return -1;
    const size_t blocks = (count + BLOCK - 1) / BLOCK;
#pragma omp parallel num_threads(threads > 0 ? threads : omp_get_max_threads())
    {
        block_ws* ws = (block_ws*)aligned_alloc(64, (sizeof(block_ws) + 63) & ~(size_t)63);
#pragma omp for schedule(dynamic, 1)
        for (size_t b = 0; b < blocks; b++) {
            int32_t ca[BLOCK], cb[BLOCK];
            uint32_t off[BLOCK];
            uint32_t* o[BLOCK];
            const uint32_t *a[BLOCK], *c[BLOCK];
            int nb = 0;
            for (size_t g = b * BLOCK; g < count && nb < BLOCK; g++, nb++) {
                int x, y, m;
                orc_gate_coeffs(ops[g * (size_t)ops_stride], &x, &y, &m);
                ca[nb] = x; cb[nb] = y; off[nb] = (uint32_t)m * ORC_MU;
                o[nb] = out + g * ORC_LVL0_WORDS;
                a[nb] = in0 + g * ORC_LVL0_WORDS;
                c[nb] = in1 + g * ORC_LVL0_WORDS;
            }
            gate_block(ek, nb, ca, cb, off, o, a, c, ws);
        }
        free(ws);
    }
    return 0;
}

/* which code path the loader picked on this host: 4 = AVX-512, 3 = AVX2 + FMA, 0 = baseline */
int fast_isa_level(void)
{
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") &&
        __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512cd")) return 4;
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return 3;
    return 0;
}
