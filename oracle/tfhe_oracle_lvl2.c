/*
 * tfhe_oracle_lvl2.c -- CPU oracle for the N = 2048 / 64-bit-torus gate path.
 * TEST INFRASTRUCTURE ONLY; status and citations in tfhe_oracle_lvl2.h.
 */
#include "tfhe_oracle_lvl2.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

#define N2 ORC2_N

/* ------------------------------------------------------------------ */
/* Goldilocks field 2^64 - 2^32 + 1                                   */
/* ------------------------------------------------------------------ */
#define GL_P 0xFFFFFFFF00000001ull
#define GL_EPS 0xFFFFFFFFull              /* 2^64 mod p */

static inline uint64_t gl_add(uint64_t a, uint64_t b)
{
    u128 s = (u128)a + b;
    return (uint64_t)(s >= GL_P ? s - GL_P : s);
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }
/* 2^64 = 2^32 - 1 and 2^96 = -1 (mod p) */
static inline uint64_t gl_mul(uint64_t a, uint64_t b)
{
    const u128 x = (u128)a * b;
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    const uint64_t hi_lo = hi & GL_EPS, hi_hi = hi >> 32;
    u128 s = (u128)lo + (u128)hi_lo * GL_EPS + (GL_P - hi_hi);     /* < 3 * 2^64 */
    s = (u128)(uint64_t)s + (u128)(uint64_t)(s >> 64) * GL_EPS;    /* < 2^64 + 2^34 */
    s = (u128)(uint64_t)s + (u128)(uint64_t)(s >> 64) * GL_EPS;    /* < 2^64 */
    uint64_t r = (uint64_t)s;
    return r >= GL_P ? r - GL_P : r;
}
static uint64_t gl_pow(uint64_t a, uint64_t e)
{
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, a);
        a = gl_mul(a, a);
        e >>= 1;
    }
    return r;
}
static inline uint64_t gl_from_i64(int64_t a) { return a < 0 ? GL_P - (uint64_t)(-a) : (uint64_t)a; }
/* centred lift */
static inline int64_t gl_lift(uint64_t v) { return v > GL_P / 2 ? -(int64_t)(GL_P - v) : (int64_t)v; }

static uint32_t bitrev(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

static uint64_t g_fwd[N2], g_inv[N2], g_ninv;
static int g_ready = 0;

/* same table shape as GenerateRootTables, src/ntt_gpu/ntt_gpuntt.cu:66-112 */
static void tables(void)
{
    int ready;
#pragma omp atomic read
    ready = g_ready;
    if (ready) return;
#pragma omp critical(orc2_tables)
    {
        if (!g_ready) {
            const uint64_t psi = gl_pow(7, (GL_P - 1) / (2 * N2));   /* 7 generates the unit group */
            const uint64_t psi_inv = gl_pow(psi, GL_P - 2);
            static uint64_t f[N2], v[N2];
            f[0] = 1; v[0] = 1;
            for (int i = 1; i < N2; i++) {
                f[i] = gl_mul(f[i - 1], psi);
                v[i] = gl_mul(v[i - 1], psi_inv);
            }
            for (int i = 0; i < N2; i++) {
                const uint32_t b = bitrev((uint32_t)i, ORC2_NBIT);
                g_fwd[i] = f[b];
                g_inv[i] = v[b];
            }
            g_ninv = gl_pow(N2, GL_P - 2);
#pragma omp atomic write
            g_ready = 1;
        }
    }
}

/* merged-psi Cooley-Tukey / Gentleman-Sande, the structure of
 * include/ntt_gpu/ntt_gpuntt.cuh:232-276,342-392 at N = 2048 */
static void ntt_forward(uint64_t* x)
{
    int t = N2 >> 1;
    for (int m = 1; m < N2; m <<= 1, t >>= 1)
        for (int g = 0; g < m; g++) {
            const uint64_t w = g_fwd[m + g];
            uint64_t* a = x + 2 * g * t;
            for (int j = 0; j < t; j++) {
                const uint64_t u = a[j], v = gl_mul(a[j + t], w);
                a[j] = gl_add(u, v);
                a[j + t] = gl_sub(u, v);
            }
        }
}
static void ntt_inverse(uint64_t* x)
{
    int t = 1;
    for (int m = N2 >> 1; m >= 1; m >>= 1, t <<= 1)
        for (int g = 0; g < m; g++) {
            const uint64_t w = g_inv[m + g];
            uint64_t* a = x + 2 * g * t;
            for (int j = 0; j < t; j++) {
                const uint64_t u = a[j], v = a[j + t];
                a[j] = gl_add(u, v);
                a[j + t] = gl_mul(gl_sub(u, v), w);
            }
        }
    for (int i = 0; i < N2; i++) x[i] = gl_mul(x[i], g_ninv);
}

void orc2_polymul_schoolbook(uint64_t* res, const int32_t* a, const uint64_t* b)
{
    for (int i = 0; i < N2; i++) res[i] = 0;
    for (int i = 0; i < N2; i++)
        for (int j = 0; j < N2; j++) {
            const uint64_t prod = (uint64_t)(int64_t)a[i] * b[j];
            const int k = i + j;
            if (k < N2) res[k] += prod;
            else res[k - N2] -= prod;
        }
}
/* b = lo + 2^32 hi: two exact products, recombined mod 2^64 */
void orc2_polymul_ntt(uint64_t* res, const int32_t* a, const uint64_t* b)
{
    tables();
    uint64_t *fa = malloc(3 * N2 * sizeof(uint64_t)), *lo = fa + N2, *hi = lo + N2;
    for (int i = 0; i < N2; i++) {
        fa[i] = gl_from_i64(a[i]);
        lo[i] = b[i] & 0xffffffffu;
        hi[i] = b[i] >> 32;
    }
    ntt_forward(fa); ntt_forward(lo); ntt_forward(hi);
    for (int i = 0; i < N2; i++) {
        lo[i] = gl_mul(fa[i], lo[i]);
        hi[i] = gl_mul(fa[i], hi[i]);
    }
    ntt_inverse(lo); ntt_inverse(hi);
    for (int i = 0; i < N2; i++) res[i] = (uint64_t)gl_lift(lo[i]) + ((uint64_t)gl_lift(hi[i]) << 32);
    free(fa);
}

/* ------------------------------------------------------------------ */
/* Keys, encryption                                                   */
/* ------------------------------------------------------------------ */
static const double ALPHA0 = 1.0 / 32768.0;                 /* 2^-15 */
static const double ALPHA2 = 1.0 / 17592186044416.0;        /* 2^-44 */

static inline double rng_unit(orc_rng* r) { return ((double)(orc_rng_next(r) >> 11) + 1.0) * (1.0 / 9007199254740992.0); }
static double rng_normal(orc_rng* r)
{
    const double u1 = rng_unit(r), u2 = rng_unit(r);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}
static inline uint32_t gauss32(orc_rng* r, double alpha) { return (uint32_t)(int64_t)llround(rng_normal(r) * alpha * 4294967296.0); }
static inline uint64_t gauss64(orc_rng* r, double alpha) { return (uint64_t)(int64_t)llround(rng_normal(r) * alpha * 18446744073709551616.0); }

void orc2_keygen(uint64_t seed, uint32_t* s2)
{
    orc_rng r;
    orc_rng_seed(&r, seed ^ 0x6c766c32ull);
    for (int i = 0; i < N2; i++) s2[i] = (uint32_t)(orc_rng_next(&r) >> 63);
}

void orc2_tlwe_encrypt(orc_rng* r, const uint32_t* s2, int bit, uint64_t* ct)
{
    uint64_t b = (bit ? ORC2_MU : (uint64_t)0 - ORC2_MU) + gauss64(r, ALPHA2);
    for (int i = 0; i < N2; i++) {
        ct[i] = orc_rng_next(r);
        b += ct[i] * s2[i];
    }
    ct[N2] = b;
}
uint64_t orc2_tlwe_phase(const uint32_t* s2, const uint64_t* ct)
{
    uint64_t ph = ct[N2];
    for (int i = 0; i < N2; i++) ph -= ct[i] * s2[i];
    return ph;
}
int orc2_tlwe_decrypt(const uint32_t* s2, const uint64_t* ct) { return (int64_t)orc2_tlwe_phase(s2, ct) > 0; }

static void trlwe_zero64(orc_rng* r, const uint32_t* s2, uint64_t* a, uint64_t* b)
{
    for (int i = 0; i < N2; i++) {
        a[i] = orc_rng_next(r);
        b[i] = gauss64(r, ALPHA2);
    }
    for (int j = 0; j < N2; j++) {
        if (!s2[j]) continue;
        for (int m = 0; m < j; m++) b[m] -= a[N2 + m - j];
        for (int m = j; m < N2; m++) b[m] += a[m - j];
    }
}

void orc2_bkgen(uint64_t seed, const uint32_t* s0, const uint32_t* s2, uint64_t* bk)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < ORC_n; i++) {
        orc_rng r;
        orc_rng_seed(&r, seed * 0x100000001b3ull + 0x2000000ull + (uint64_t)i);
        for (int row = 0; row < ORC2_BK_ROWS; row++) {
            uint64_t* a = bk + (((size_t)i * ORC2_BK_ROWS + row) * 2 + 0) * N2;
            uint64_t* b = a + N2;
            trlwe_zero64(&r, s2, a, b);
            const int j = row / ORC2_L, d = row % ORC2_L;
            const uint64_t h = (uint64_t)1 << (64 - (d + 1) * ORC2_BGBIT);
            (j == 0 ? a : b)[0] += (uint64_t)s0[i] * h;
        }
    }
}

void orc2_kskgen(uint64_t seed, const uint32_t* s0, const uint32_t* s2, uint32_t* ksk)
{
#pragma omp parallel for schedule(static)
    for (int j = 0; j < N2; j++) {
        orc_rng r;
        orc_rng_seed(&r, seed * 0x100000001b3ull + 0x5eed2000ull + (uint64_t)j);
        for (int kap = 0; kap < ORC2_T; kap++)
            for (int v = 1; v <= ORC2_KS_NUMBASE; v++) {
                uint32_t* ct = ksk + ((((size_t)j * ORC2_T + kap) * ORC2_KS_NUMBASE) + (v - 1)) * ORC_LVL0_WORDS;
                const uint32_t msg = (uint32_t)v * s2[j] * ((uint32_t)1u << (32 - (kap + 1) * ORC2_BASEBIT));
                uint32_t b = msg + gauss32(&r, ALPHA0);
                for (int i = 0; i < ORC_n; i++) {
                    ct[i] = (uint32_t)(orc_rng_next(&r) >> 32);
                    b += ct[i] * s0[i];
                }
                ct[ORC_n] = b;
            }
    }
}

/* ------------------------------------------------------------------ */
/* Evaluation key: every key polynomial as two 32-bit halves, NTT'd   */
/* ------------------------------------------------------------------ */
struct orc2_evalkey {
    uint64_t* bkntt;      /* [n][rows][out][half][N] */
    const uint32_t* ksk;  /* borrowed */
};

orc2_evalkey* orc2_evalkey_create(const uint64_t* bk, const uint32_t* ksk)
{
    tables();
    orc2_evalkey* ek = (orc2_evalkey*)malloc(sizeof(*ek));
    ek->bkntt = (uint64_t*)malloc(ORC2_BK_WORDS * 2 * sizeof(uint64_t));
    ek->ksk = ksk;
    const long polys = (long)(ORC2_BK_WORDS / N2);
#pragma omp parallel for schedule(static)
    for (long p = 0; p < polys; p++) {
        uint64_t* lo = ek->bkntt + (size_t)p * 2 * N2;
        uint64_t* hi = lo + N2;
        const uint64_t* src = bk + (size_t)p * N2;
        for (int i = 0; i < N2; i++) {
            lo[i] = src[i] & 0xffffffffu;
            hi[i] = src[i] >> 32;
        }
        ntt_forward(lo);
        ntt_forward(hi);
    }
    return ek;
}
void orc2_evalkey_destroy(orc2_evalkey* ek)
{
    if (!ek) return;
    free(ek->bkntt);
    free(ek);
}

/* ------------------------------------------------------------------ */
/* Blind rotate lvl02                                                 */
/* ------------------------------------------------------------------ */
/* modSwitchFromTorus<lvl02>, include/gatebootstrapping_gpu.cuh:10-16 */
static inline uint32_t mod_switch(uint32_t phase) { return phase >> (32 - 1 - ORC2_NBIT); }

/* RotatedTestVector<lvl2param>, :29-52 */
static void rotated_test_vector(uint64_t* acc, uint32_t bar, uint64_t mu)
{
    for (int i = 0; i < N2; i++) {
        acc[i] = 0;
        if (bar == 2 * N2) acc[i + N2] = mu;
        else acc[i + N2] = (((uint32_t)i < (bar & (N2 - 1))) ^ (bar >> ORC2_NBIT)) ? (uint64_t)0 - mu : mu;
    }
}

/* Accumulate<lvl02>, :115-285, with 64-bit decomposition constants */
static void accumulate(uint64_t* acc, uint32_t a_bar, const uint64_t* tgsw_ntt, uint64_t* scratch)
{
    const uint64_t decomp_mask = ((uint64_t)1 << ORC2_BGBIT) - 1;
    const int64_t decomp_half = (int64_t)1 << (ORC2_BGBIT - 1);
    uint64_t decomp_offset = 0;                                 /* offsetgen<lvl2param> :18-27 */
    for (int i = 1; i <= ORC2_L; i++) decomp_offset += ((uint64_t)1 << (ORC2_BGBIT - 1)) << (64 - i * ORC2_BGBIT);
    const uint64_t roundoffset = (uint64_t)1 << (64 - ORC2_L * ORC2_BGBIT - 1);

    uint64_t* accum = scratch;                /* [out][half][N] */
    uint64_t* work = scratch + 4 * N2;
    memset(accum, 0, 4 * N2 * sizeof(uint64_t));

    for (int j = 0; j < 2; j++)
        for (int digit = 0; digit < ORC2_L; digit++) {
            for (int i = 0; i < N2; i++) {                       /* :157-181 */
                uint64_t temp = acc[j * N2 + (((uint32_t)i - a_bar) & (N2 - 1))];
                temp = (((uint32_t)i < (a_bar & (N2 - 1))) ^ (a_bar >> ORC2_NBIT)) ? (uint64_t)0 - temp : temp;
                temp -= acc[j * N2 + i];
                temp += decomp_offset + roundoffset;
                const int64_t dv = (int64_t)((temp >> (64 - (digit + 1) * ORC2_BGBIT)) & decomp_mask) - decomp_half;
                work[i] = gl_from_i64(dv);
            }
            ntt_forward(work);
            const int row = j * ORC2_L + digit;                  /* :206-221 */
            for (int out = 0; out < 2; out++)
                for (int half = 0; half < 2; half++) {
                    const uint64_t* key = tgsw_ntt + (((size_t)row * 2 + out) * 2 + half) * N2;
                    uint64_t* ac = accum + (out * 2 + half) * N2;
                    for (int i = 0; i < N2; i++) ac[i] = gl_add(ac[i], gl_mul(work[i], key[i]));
                }
        }
    for (int out = 0; out < 2; out++) {                          /* :227-284 */
        uint64_t* lo = accum + (out * 2) * N2;
        uint64_t* hi = lo + N2;
        ntt_inverse(lo);
        ntt_inverse(hi);
        for (int i = 0; i < N2; i++)
            acc[out * N2 + i] += (uint64_t)gl_lift(lo[i]) + ((uint64_t)gl_lift(hi[i]) << 32);
    }
}

/* __BlindRotatePreAdd__ / __BlindRotate__ at lvl02, :287-345 (pre-add done by the caller) */
void orc2_blind_rotate(const orc2_evalkey* ek, uint64_t* acc, const uint32_t* tlwe0, int steps)
{
    tables();
    const uint32_t bar = 2 * N2 - mod_switch(tlwe0[ORC_n]);
    rotated_test_vector(acc, bar, ORC2_MU);
    const uint32_t roundoffset = 1u << (32 - 2 - ORC2_NBIT);
    if (steps < 0 || steps > ORC_n) steps = ORC_n;
    uint64_t* scratch = (uint64_t*)malloc(5 * N2 * sizeof(uint64_t));
    for (int i = 0; i < steps; i++) {
        const uint32_t a_bar = mod_switch(tlwe0[i] + roundoffset);
        accumulate(acc, a_bar, ek->bkntt + (size_t)i * ORC2_BK_ROWS * 2 * 2 * N2, scratch);
    }
    free(scratch);
}

/* __SampleExtractIndex__<lvl2param,0>, src/bootstrap_gpu.cu:366-381 */
void orc2_sample_extract0(uint64_t* res, const uint64_t* in)
{
    res[0] = in[0];
    for (int i = 1; i < N2; i++) res[i] = (uint64_t)0 - in[N2 - i];
    res[N2] = in[N2];
}

/* KeySwitchFromTLWE<lvl20>, include/keyswitch_gpu.cuh:83-134 (iksoffsetgen :13-23) */
void orc2_keyswitch(const orc2_evalkey* ek, uint32_t* lwe, const uint64_t* tlwe)
{
    const uint64_t roundoffset = (uint64_t)1 << (64 - (1 + ORC2_BASEBIT * ORC2_T));
    uint64_t decompoffset = 0;
    for (int i = 1; i <= ORC2_T; i++) decompoffset += (((uint64_t)1 << ORC2_BASEBIT) / 2) << (64 - i * ORC2_BASEBIT);
    const uint64_t mask = ((uint64_t)1 << ORC2_BASEBIT) - 1;
    const int32_t halfbase = 1 << (ORC2_BASEBIT - 1);
    const uint32_t* ksk = ek->ksk;

    for (int i = 0; i < ORC_n; i++) lwe[i] = 0;
    lwe[ORC_n] = (uint32_t)((tlwe[N2] + ((uint64_t)1 << 31)) >> 32);          /* :100-101 */
    for (int j = 0; j < N2; j++) {
        const uint64_t tmp = tlwe[j] + decompoffset + roundoffset;
        for (int k = 0; k < ORC2_T; k++) {
            const int32_t val = (int32_t)((tmp >> (64 - (k + 1) * ORC2_BASEBIT)) & mask) - halfbase;
            if (val == 0) continue;
            const uint32_t* row = ksk + (((size_t)j * ORC2_T + k) * ORC2_KS_NUMBASE + (size_t)(abs(val) - 1)) * ORC_LVL0_WORDS;
            if (val > 0) for (int i = 0; i <= ORC_n; i++) lwe[i] -= row[i];
            else for (int i = 0; i <= ORC_n; i++) lwe[i] += row[i];
        }
    }
}

/* ------------------------------------------------------------------ */
/* Gates on lvl0 ciphertexts through the lvl2 ring                    */
/* ------------------------------------------------------------------ */
static void bootstrap_to_tlwe2(const orc2_evalkey* ek, uint64_t* tlwe2, int ca, const uint32_t* in0,
                               int cb, const uint32_t* in1, uint32_t off)
{
    uint32_t c[ORC_LVL0_WORDS];
    for (int i = 0; i < ORC_LVL0_WORDS; i++) c[i] = (uint32_t)ca * in0[i] + (uint32_t)cb * in1[i];
    c[ORC_n] += off;
    uint64_t* acc = (uint64_t*)malloc(ORC2_TRLWE_WORDS * sizeof(uint64_t));
    orc2_blind_rotate(ek, acc, c, -1);
    orc2_sample_extract0(tlwe2, acc);
    free(acc);
}

/* __HomGate__ (br -> iks), src/bootstrap_gpu.cu:402-421, and the Mux of :515-588,
 * instantiated at brP = lvl02, iksP = lvl20 */
void orc2_gate(const orc2_evalkey* ek, int op, uint32_t* out,
               const uint32_t* in0, const uint32_t* in1, const uint32_t* in2)
{
    if (op == ORC_NOT || op == ORC_COPY) {
        for (int i = 0; i < ORC_LVL0_WORDS; i++) out[i] = (op == ORC_NOT) ? (uint32_t)(0u - in0[i]) : in0[i];
        return;
    }
    uint64_t* t = (uint64_t*)malloc(2 * ORC2_LVL2_WORDS * sizeof(uint64_t));
    if (op == ORC_MUX || op == ORC_NMUX) {
        uint64_t* t0 = t + ORC2_LVL2_WORDS;
        const uint32_t negmu = (uint32_t)(0u - ORC_MU);
        bootstrap_to_tlwe2(ek, t, 1, in0, 1, in1, negmu);
        bootstrap_to_tlwe2(ek, t0, -1, in0, 1, in2, negmu);
        for (int i = 0; i < ORC2_LVL2_WORDS; i++) t[i] = (op == ORC_MUX) ? t[i] + t0[i] : (uint64_t)0 - t[i] - t0[i];
        t[N2] += (op == ORC_MUX) ? ORC2_MU : (uint64_t)0 - ORC2_MU;
    } else {
        int ca, cb, om;
        orc_gate_coeffs(op, &ca, &cb, &om);
        bootstrap_to_tlwe2(ek, t, ca, in0, cb, in1, (uint32_t)om * ORC_MU);
    }
    orc2_keyswitch(ek, out, t);
    free(t);
}

void orc2_gate_batch(const orc2_evalkey* ek, const int32_t* ops, int ops_stride, size_t count,
                     uint32_t* out, const uint32_t* in0, const uint32_t* in1, const uint32_t* in2,
                     int threads)
{
    const size_t w = ORC_LVL0_WORDS;
    tables();
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (long g = 0; g < (long)count; g++)
        orc2_gate(ek, ops[(size_t)g * ops_stride], out + g * w, in0 + g * w,
                  in1 ? in1 + g * w : NULL, in2 ? in2 + g * w : NULL);
}
