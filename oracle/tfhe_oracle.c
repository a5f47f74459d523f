/*
 * tfhe_oracle.c -- CPU restatement of the cuFHE gate-bootstrapping path.
 * TEST INFRASTRUCTURE ONLY (see tfhe_oracle.h for the rules and the parity
 * status: "parity unpinned" at ciphertext-word level, pinned at decrypt level
 * and at NTT-product level).
 *
 * The polynomial products use the reference's live NTT: radix-2, merged-psi
 * Cooley-Tukey forward / Gentleman-Sande inverse over p = 2^60 + 30721 with
 * Barrett reduction (include/ntt_gpu/ntt_gpuntt.cuh:36-40,170-210,232-276,
 * 342-392; tables src/ntt_gpu/ntt_gpuntt.cu:66-112).  Because the arithmetic
 * is exact the resulting ciphertext words do not depend on the
 * prime or on the butterfly order, so this restatement is a bit-exact oracle
 * for any exact device implementation.
 */
#include "tfhe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ */
/* PRNG                                                               */
/* ------------------------------------------------------------------ */
static uint64_t splitmix64(uint64_t* x)
{
    uint64_t z = (*x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
void orc_rng_seed(orc_rng* r, uint64_t seed)
{
    uint64_t x = seed;
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&x);
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t orc_rng_next(orc_rng* r)
{
    uint64_t* s = r->s;
    const uint64_t result = rotl64(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}
static inline uint32_t rng_u32(orc_rng* r) { return (uint32_t)(orc_rng_next(r) >> 32); }
static inline double rng_unit(orc_rng* r) /* (0,1] */
{
    return ((double)(orc_rng_next(r) >> 11) + 1.0) * (1.0 / 9007199254740992.0);
}
/* modular Gaussian torus noise of standard deviation alpha (App. B) */
static uint32_t rng_gauss_torus(orc_rng* r, double alpha)
{
    double u1 = rng_unit(r), u2 = rng_unit(r);
    double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
    return (uint32_t)(int64_t)llround(z * alpha * 4294967296.0);
}

static const double ALPHA0 = ORC_ALPHA0;
static const double ALPHA1 = ORC_ALPHA1;

/* ------------------------------------------------------------------ */
/* Reference NTT prime arithmetic                                     */
/* ------------------------------------------------------------------ */
/* the small-modulus constants and switches, include/ntt_gpu/ntt_small_modulus.cuh:36-68,147-177 */
#define SMALL_P 655360001u                                  /* :44  K * 2^SHIFTAMOUNT + 1, K = 625, SHIFTAMOUNT = 20 */
#define SMALL_INV_MODSWITCH_MUL (((uint64_t)1 << 63) / SMALL_P)   /* :67 */
uint32_t orc_smallmod_modulus(void) { return SMALL_P; }
uint32_t orc_smallmod_from_torus(uint32_t torus_val)        /* torus32_to_ntt_mod :147-154 */
{
    uint64_t temp = (uint64_t)torus_val * SMALL_P + ((uint64_t)1 << 31);
    return (uint32_t)(temp >> 32);
}
uint32_t orc_smallmod_to_torus(int32_t ntt_val)             /* ntt_mod_to_torus32 :166-177 */
{
    uint32_t a = (ntt_val < 0) ? (uint32_t)(ntt_val + (int32_t)SMALL_P) : (uint32_t)ntt_val;
    uint64_t temp = (uint64_t)a * SMALL_INV_MODSWITCH_MUL;
    temp = (temp + ((uint64_t)1 << 30)) >> 31;
    return (uint32_t)temp;
}

#ifdef ORC_SMALL_NTT_MODULUS
/* -DUSE_SMALL_NTT_MODULUS: the transforms of include/ntt_gpu/ntt_small_modulus.cuh:201-300 have the stage structure and
 * table indexing of the 64-bit ones (the code below), over P with small_mod_mult / small_mod_add / small_mod_sub (:117-140);
 * psi is found as src/ntt_gpu/ntt_small_modulus.cu:71-110 finds it */
#define NTT_P ((uint64_t)SMALL_P)
static uint64_t small_psi(void);
#define NTT_PSI small_psi()
uint64_t orc_ntt_modulus(void) { return NTT_P; }
uint64_t orc_ntt_psi(void) { return NTT_PSI; }
uint64_t orc_ntt_barrett_mu(void) { return ((uint64_t)1 << 60) / SMALL_P; }     /* BARRETT_MU :52 (unused by the live butterflies) */
static inline uint64_t barrett_mult(uint64_t a, uint64_t b) { return (a * b) % NTT_P; }   /* small_mod_mult :137-140; a, b <= P */
uint64_t orc_ntt_mulmod(uint64_t a, uint64_t b) { return barrett_mult(a, b); }
#else
#define NTT_P 1152921504606877697ull     /* include/ntt_gpu/ntt_gpuntt.cuh:36 */
#define NTT_MU 9223372036854530040ull    /* :39  = floor(2^123 / p)           */
#define NTT_BIT 61                       /* :40                               */
#define NTT_PSI 1689264667710614ull      /* src/ntt_gpu/ntt_gpuntt.cu:32      */

uint64_t orc_ntt_modulus(void) { return NTT_P; }
uint64_t orc_ntt_psi(void) { return NTT_PSI; }
uint64_t orc_ntt_barrett_mu(void) { return NTT_MU; }

/* barrett_mult, include/ntt_gpu/ntt_gpuntt.cuh:170-182 */
static inline uint64_t barrett_mult(uint64_t a, uint64_t b)
{
    u128 z = (u128)a * b;
    uint64_t w = (uint64_t)(z >> (NTT_BIT - 2));
    w = (uint64_t)(((u128)w * NTT_MU) >> (NTT_BIT + 3));
    z -= (u128)w * NTT_P;
    uint64_t r = (uint64_t)z;
    return (r >= NTT_P) ? r - NTT_P : r;   /* r < 2p: checked against % in tests/test_oracle.py */
}
uint64_t orc_ntt_mulmod(uint64_t a, uint64_t b) { return barrett_mult(a, b); }
#endif
static inline uint64_t mod_add(uint64_t a, uint64_t b) /* :184-188 */
{
    uint64_t s = a + b;
    return (s >= NTT_P) ? s - NTT_P : s;
}
static inline uint64_t mod_sub(uint64_t a, uint64_t b) /* :190-194 */
{
    uint64_t d = a + NTT_P - b;
    return (d >= NTT_P) ? d - NTT_P : d;
}
static uint64_t mod_pow(uint64_t a, uint64_t e)
{
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = barrett_mult(r, a);
        a = barrett_mult(a, a);
        e >>= 1;
    }
    return r;
}
#ifdef ORC_SMALL_NTT_MODULUS
/* find_primitive_root, src/ntt_gpu/ntt_small_modulus.cu:71-110: the smallest g >= 3 that is neither a square nor a fifth power
 * residue generates Z_P^* (P - 1 = 2^20 5^4); psi = g^((P-1)/2048), a primitive 2048-th root (psi^1024 = -1) */
static uint64_t small_psi(void)
{
    uint64_t g = 3;
    while (mod_pow(g, (NTT_P - 1) / 2) == 1 || mod_pow(g, (NTT_P - 1) / 5) == 1) g++;
    return mod_pow(g, (NTT_P - 1) / 2048);
}
#endif
static uint32_t bitrev(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

static uint64_t g_fwd[ORC_N], g_inv[ORC_N], g_ninv;
static int g_tables_ready = 0;

/* GenerateRootTables, src/ntt_gpu/ntt_gpuntt.cu:66-112 */
static void ntt_tables(void)
{
    int ready;
#pragma omp atomic read
    ready = g_tables_ready;
    if (ready) return;
#pragma omp critical(orc_tables)
    {
        if (!g_tables_ready) {
            /* the reference's psi is a 2048-th root (N = 1024); for N = 512 its square
             * (ComputePsi, src/ntt_gpu/ntt_gpuntt.cu:40-64) */
            uint64_t psi = mod_pow(NTT_PSI, 1024 / ORC_N), psi_inv = mod_pow(psi, NTT_P - 2);
            uint64_t f[ORC_N], v[ORC_N];
            f[0] = 1; v[0] = 1;
            for (int i = 1; i < ORC_N; i++) {
                f[i] = barrett_mult(f[i - 1], psi);
                v[i] = barrett_mult(v[i - 1], psi_inv);
            }
            for (int i = 0; i < ORC_N; i++) {
                uint32_t b = bitrev((uint32_t)i, ORC_NBIT);
                g_fwd[i] = f[b];
                g_inv[i] = v[b];
            }
            g_ninv = mod_pow(ORC_N, NTT_P - 2);
#pragma omp atomic write
            g_tables_ready = 1;
        }
    }
}
uint64_t orc_ntt_n_inverse(void) { ntt_tables(); return g_ninv; }

/* SmallForwardNTT_1024, include/ntt_gpu/ntt_gpuntt.cuh:232-276: stage s has
 * m = 2^s groups, stride t = N/(2m); butterfly (a, a+t) of group g uses
 * root_table[m + g]; CooleyTukeyUnit :197-202 */
void orc_ntt_forward(uint64_t* x)
{
    ntt_tables();
    int t = ORC_N >> 1;
    for (int m = 1; m < ORC_N; m <<= 1, t >>= 1) {
        for (int g = 0; g < m; g++) {
            const uint64_t w = g_fwd[m + g];
            uint64_t* a = x + 2 * g * t;
            for (int j = 0; j < t; j++) {
                uint64_t u = a[j], v = barrett_mult(a[j + t], w);
                a[j] = mod_add(u, v);
                a[j + t] = mod_sub(u, v);
            }
        }
    }
}
/* SmallInverseNTT_1024, :342-392; GentlemanSandeUnit :205-210; N^-1 :389-390 */
void orc_ntt_inverse(uint64_t* x)
{
    ntt_tables();
    int t = 1;
    for (int m = ORC_N >> 1; m >= 1; m >>= 1, t <<= 1) {
        for (int g = 0; g < m; g++) {
            const uint64_t w = g_inv[m + g];
            uint64_t* a = x + 2 * g * t;
            for (int j = 0; j < t; j++) {
                uint64_t u = a[j], v = a[j + t];
                a[j] = mod_add(u, v);
                a[j + t] = barrett_mult(mod_sub(u, v), w);
            }
        }
    }
    for (int i = 0; i < ORC_N; i++) x[i] = barrett_mult(x[i], g_ninv);
}

/* FFP(int32): include/ntt_gpu/ntt_gpuntt.cuh:68-70 */
static inline uint64_t ffp_from_i32(int32_t a)
{
    return (a < 0) ? NTT_P - (uint64_t)(-(int64_t)a) : (uint64_t)a;
}
/* centred lift truncated to the torus: include/gatebootstrapping_gpu.cuh:258-281 */
static inline uint32_t ffp_lift_u32(uint64_t v)
{
#ifdef ORC_SMALL_NTT_MODULUS
    /* include/gatebootstrapping_gpu.cuh:236-248: centred, then switched back to the 2^32 discretisation */
    const int32_t signed_val = (v > NTT_P / 2) ? (int32_t)((uint32_t)v - SMALL_P) : (int32_t)v;
    return orc_smallmod_to_torus(signed_val);
#else
    return (v > NTT_P / 2) ? (uint32_t)((int64_t)v - (int64_t)NTT_P) : (uint32_t)v;
#endif
}

void orc_polymul_schoolbook(uint32_t* res, const int32_t* a, const uint32_t* b)
{
    for (int i = 0; i < ORC_N; i++) res[i] = 0;
    for (int i = 0; i < ORC_N; i++)
        for (int j = 0; j < ORC_N; j++) {
            uint32_t prod = (uint32_t)a[i] * b[j];
            int k = i + j;
            if (k < ORC_N) res[k] += prod;
            else res[k - ORC_N] -= prod;
        }
}
void orc_polymul_ntt(uint32_t* res, const int32_t* a, const uint32_t* b)
{
    uint64_t fa[ORC_N], fb[ORC_N];
    for (int i = 0; i < ORC_N; i++) {
        fa[i] = ffp_from_i32(a[i]);
        fb[i] = (uint64_t)b[i];          /* unsigned torus: ntt_gpuntt.cuh:495-496 */
    }
    orc_ntt_forward(fa);
    orc_ntt_forward(fb);
    for (int i = 0; i < ORC_N; i++) fa[i] = barrett_mult(fa[i], fb[i]);
    orc_ntt_inverse(fa);
    for (int i = 0; i < ORC_N; i++) res[i] = ffp_lift_u32(fa[i]);
}

/* ------------------------------------------------------------------ */
/* Keys, encryption (App. B: b = <a,s> + m + e, binary keys)          */
/* ------------------------------------------------------------------ */
void orc_keygen(uint64_t seed, uint32_t* s0, uint32_t* s1)
{
    orc_rng r;
    orc_rng_seed(&r, seed);
    for (int i = 0; i < ORC_n; i++) s0[i] = (uint32_t)(orc_rng_next(&r) >> 63);
    for (int i = 0; i < ORC_K * ORC_N; i++) s1[i] = (uint32_t)(orc_rng_next(&r) >> 63);
}

static inline int lvl_dim(int level) { return level ? ORC_K * ORC_N : ORC_n; }

void orc_tlwe_encrypt(orc_rng* r, int level, const uint32_t* key, int bit, uint32_t* ct)
{
    const int n = lvl_dim(level);
    uint32_t b = (bit ? ORC_MU : (uint32_t)(0u - ORC_MU)) + rng_gauss_torus(r, level ? ALPHA1 : ALPHA0);
    for (int i = 0; i < n; i++) {
        ct[i] = rng_u32(r);
        b += ct[i] * key[i];
    }
    ct[n] = b;
}
uint32_t orc_tlwe_phase(int level, const uint32_t* key, const uint32_t* ct)
{
    const int n = lvl_dim(level);
    uint32_t ph = ct[n];
    for (int i = 0; i < n; i++) ph -= ct[i] * key[i];
    return ph;
}
int orc_tlwe_decrypt(int level, const uint32_t* key, const uint32_t* ct)
{
    return (int32_t)orc_tlwe_phase(level, key, ct) > 0;
}
void orc_tlwe_encrypt_batch(uint64_t seed, int level, const uint32_t* key,
                            const uint8_t* bits, size_t count, uint32_t* cts)
{
    orc_rng r;
    orc_rng_seed(&r, seed);
    const size_t w = (size_t)lvl_dim(level) + 1;
    for (size_t g = 0; g < count; g++) orc_tlwe_encrypt(&r, level, key, bits[g], cts + g * w);
}
void orc_tlwe_decrypt_batch(int level, const uint32_t* key, const uint32_t* cts,
                            size_t count, uint8_t* bits)
{
    const size_t w = (size_t)lvl_dim(level) + 1;
    for (size_t g = 0; g < count; g++) bits[g] = (uint8_t)orc_tlwe_decrypt(level, key, cts + g * w);
}

/* TRLWE encryption of zero under s1 (k mask polynomials): b = sum_c a_c * s1_c + e (negacyclic, binary key);
 * t points at the k+1 polynomials a_0 .. a_{k-1}, b */
static void trlwe_zero(orc_rng* r, const uint32_t* s1, uint32_t* t)
{
    uint32_t* b = t + ORC_K * ORC_N;
#if ORC_K == 1
    for (int i = 0; i < ORC_N; i++) {       /* the draw order the committed golden vectors were made with */
        t[i] = rng_u32(r);
        b[i] = rng_gauss_torus(r, ALPHA1);
    }
#else
    for (int c = 0; c < ORC_K; c++)
        for (int i = 0; i < ORC_N; i++) t[c * ORC_N + i] = rng_u32(r);
    for (int i = 0; i < ORC_N; i++) b[i] = rng_gauss_torus(r, ALPHA1);
#endif
    for (int c = 0; c < ORC_K; c++) {
        const uint32_t* a = t + c * ORC_N;
        const uint32_t* sc = s1 + c * ORC_N;
        for (int j = 0; j < ORC_N; j++) {
            if (!sc[j]) continue;
            for (int m = 0; m < j; m++) b[m] -= a[ORC_N + m - j];
            for (int m = j; m < ORC_N; m++) b[m] += a[m - j];
        }
    }
}

void orc_bkgen(uint64_t seed, const uint32_t* s0, const uint32_t* s1, uint32_t* bk)
{
    /* one independent stream per TRGSW so generation parallelises deterministically */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < ORC_n; i++) {
        orc_rng r;
        orc_rng_seed(&r, seed * 0x100000001b3ull + (uint64_t)i + 1);
        for (int row = 0; row < ORC_BK_ROWS; row++) {
            uint32_t* t = bk + (((size_t)i * ORC_BK_ROWS + row) * (ORC_K + 1) + 0) * ORC_N;
            trlwe_zero(&r, s1, t);
            const int j = row / ORC_L, d = row % ORC_L;
            const uint32_t h = (uint32_t)1u << (32 - (d + 1) * ORC_BGBIT);
            t[j * ORC_N] += s0[i] * h;          /* App. B item 4: component j (j < k: a_j, j = k: b) */
        }
    }
}

void orc_kskgen(uint64_t seed, const uint32_t* s0, const uint32_t* s1, uint32_t* ksk)
{
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ORC_K * ORC_N; j++) {
        orc_rng r;
        orc_rng_seed(&r, seed * 0x100000001b3ull + 0x5eed0000ull + (uint64_t)j);
        for (int kap = 0; kap < ORC_T; kap++)
            for (int v = 1; v <= ORC_KS_NUMBASE; v++) {
                uint32_t* ct = ksk + ((((size_t)j * ORC_T + kap) * ORC_KS_NUMBASE) + (v - 1)) * ORC_LVL0_WORDS;
                uint32_t msg = (uint32_t)v * s1[j] * ((uint32_t)1u << (32 - (kap + 1) * ORC_BASEBIT));
                uint32_t b = msg + rng_gauss_torus(&r, ALPHA0);
                for (int i = 0; i < ORC_n; i++) {
                    ct[i] = rng_u32(&r);
                    b += ct[i] * s0[i];
                }
                ct[ORC_n] = b;
            }
    }
}

/* ------------------------------------------------------------------ */
/* Evaluation key in the NTT domain                                   */
/* ------------------------------------------------------------------ */
struct orc_evalkey {
    uint64_t* bkntt;      /* [n][rows][k+1][N], BootstrappingKeyToNTT src/bootstrap_gpu.cu:111-138 */
    const uint32_t* ksk;  /* borrowed */
};

orc_evalkey* orc_evalkey_create(const uint32_t* bk, const uint32_t* ksk)
{
    ntt_tables();
    orc_evalkey* ek = (orc_evalkey*)malloc(sizeof(*ek));
    ek->bkntt = (uint64_t*)malloc(ORC_BK_WORDS * sizeof(uint64_t));
    ek->ksk = ksk;
    const long polys = (long)(ORC_BK_WORDS / ORC_N);
#pragma omp parallel for schedule(static)
    for (long p = 0; p < polys; p++) {
        uint64_t* dst = ek->bkntt + (size_t)p * ORC_N;
        const uint32_t* src = bk + (size_t)p * ORC_N;
#ifdef ORC_SMALL_NTT_MODULUS
        for (int i = 0; i < ORC_N; i++) dst[i] = orc_smallmod_from_torus(src[i]);    /* __TRGSW2NTT__, src/bootstrap_gpu.cu:50-66 */
#else
        for (int i = 0; i < ORC_N; i++) dst[i] = (uint64_t)src[i];
#endif
        orc_ntt_forward(dst);
    }
    return ek;
}
void orc_evalkey_destroy(orc_evalkey* ek)
{
    if (!ek) return;
    free(ek->bkntt);
    free(ek);
}

/* ------------------------------------------------------------------ */
/* Blind rotate                                                       */
/* ------------------------------------------------------------------ */
/* modSwitchFromTorus, include/gatebootstrapping_gpu.cuh:10-16 */
static inline uint32_t mod_switch(uint32_t phase) { return phase >> (32 - 1 - ORC_NBIT); }

/* RotatedTestVector, include/gatebootstrapping_gpu.cuh:29-52 */
static void rotated_test_vector(uint32_t* acc, uint32_t bar, uint32_t mu)
{
    for (int i = 0; i < ORC_N; i++) {
        for (int k = 0; k < ORC_K; k++) acc[i + k * ORC_N] = 0;
        if (bar == 2 * ORC_N) acc[i + ORC_K * ORC_N] = mu;
        else
            acc[i + ORC_K * ORC_N] =
                (((uint32_t)i < (bar & (ORC_N - 1))) ^ (bar >> ORC_NBIT)) ? (uint32_t)(0u - mu) : mu;
    }
}

/* Accumulate, include/gatebootstrapping_gpu.cuh:115-285 */
static void accumulate(uint32_t* acc, uint32_t a_bar, const uint64_t* tgsw_ntt)
{
    const uint32_t decomp_mask = (1u << ORC_BGBIT) - 1;
    const int32_t decomp_half = 1 << (ORC_BGBIT - 1);
    uint32_t decomp_offset = 0;                        /* offsetgen :18-27 */
    for (int i = 1; i <= ORC_L; i++) decomp_offset += (uint32_t)(1u << (ORC_BGBIT - 1)) << (32 - i * ORC_BGBIT);
    const uint32_t roundoffset = 1u << (32 - ORC_L * ORC_BGBIT - 1);

    uint64_t accum[(ORC_K + 1) * ORC_N];
    uint64_t work[ORC_N];
    memset(accum, 0, sizeof(accum));

    for (int j = 0; j <= ORC_K; j++) {
        for (int digit = 0; digit < ORC_L; digit++) {
            for (int i = 0; i < ORC_N; i++) {          /* :157-181 */
                uint32_t temp = acc[j * ORC_N + (((uint32_t)i - a_bar) & (ORC_N - 1))];
                temp = (((uint32_t)i < (a_bar & (ORC_N - 1))) ^ (a_bar >> ORC_NBIT)) ? (uint32_t)(0u - temp) : temp;
                temp -= acc[j * ORC_N + i];
                temp += decomp_offset + roundoffset;
                int32_t digit_val = (int32_t)((temp >> (32 - (digit + 1) * ORC_BGBIT)) & decomp_mask) - decomp_half;
                work[i] = ffp_from_i32(digit_val);
            }
            orc_ntt_forward(work);
            const int digit_linear = j * ORC_L + digit;    /* :206-221 */
            for (int out_k = 0; out_k <= ORC_K; out_k++) {
                const uint64_t* bkrow = tgsw_ntt + (((size_t)(ORC_K + 1) * digit_linear + out_k) << ORC_NBIT);
                uint64_t* ac = accum + out_k * ORC_N;
                for (int i = 0; i < ORC_N; i++) ac[i] = mod_add(ac[i], barrett_mult(work[i], bkrow[i]));
            }
        }
    }
    for (int k_idx = 0; k_idx <= ORC_K; k_idx++) {          /* :227-284 */
        uint64_t* ac = accum + k_idx * ORC_N;
        orc_ntt_inverse(ac);
        for (int i = 0; i < ORC_N; i++) acc[k_idx * ORC_N + i] += ffp_lift_u32(ac[i]);
    }
}

/* __BlindRotate__ / __BlindRotatePreAdd__, include/gatebootstrapping_gpu.cuh:287-345
 * (the pre-add itself is done by the caller; it is linear mod 2^32) */
void orc_blind_rotate(const orc_evalkey* ek, uint32_t* acc, const uint32_t* tlwe0, int steps)
{
    const uint32_t bar = 2 * ORC_N - mod_switch(tlwe0[ORC_n]);
    rotated_test_vector(acc, bar, ORC_MU);
    const uint32_t roundoffset = 1u << (32 - 2 - ORC_NBIT);
    if (steps < 0 || steps > ORC_n) steps = ORC_n;
    for (int i = 0; i < steps; i++) {
        const uint32_t a_bar = mod_switch(tlwe0[i] + roundoffset);
        accumulate(acc, a_bar, ek->bkntt + (size_t)i * ORC_BK_ROWS * (ORC_K + 1) * ORC_N);
    }
}

/* __SampleExtractIndex__<P,0>, src/bootstrap_gpu.cu:366-381 */
void orc_sample_extract0(uint32_t* res, const uint32_t* in)
{
    const uint32_t index = 0;
    for (uint32_t i = 0; i <= ORC_K * ORC_N; i++) {
        if (i == ORC_K * ORC_N) res[i] = in[ORC_K * ORC_N + index];
        else {
            const uint32_t k = i >> ORC_NBIT, n = i & (ORC_N - 1);
            if (n <= index) res[i] = in[k * ORC_N + index - n];
            else res[i] = (uint32_t)(0u - in[k * ORC_N + ORC_N + index - n]);
        }
    }
}

/* KeySwitchFromTLWE, include/keyswitch_gpu.cuh:83-134 (iksoffsetgen :13-23) */
void orc_keyswitch(const orc_evalkey* ek, uint32_t* lwe, const uint32_t* tlwe)
{
    const uint32_t roundoffset = (ORC_BASEBIT * ORC_T) < 32 ? 1u << (32 - (1 + ORC_BASEBIT * ORC_T)) : 0;
    uint32_t decompoffset = 0;
    for (int i = 1; i <= ORC_T; i++) decompoffset += ((1u << ORC_BASEBIT) / 2) << (32 - i * ORC_BASEBIT);
    const uint32_t mask = (1u << ORC_BASEBIT) - 1;
    const int32_t halfbase = 1 << (ORC_BASEBIT - 1);
    const uint32_t* ksk = ek->ksk;

    for (int i = 0; i < ORC_n; i++) lwe[i] = 0;
    lwe[ORC_n] = tlwe[ORC_K * ORC_N];
    for (int j = 0; j < ORC_K * ORC_N; j++) {
        const uint32_t tmp = tlwe[j] + decompoffset + roundoffset;
        for (int k = 0; k < ORC_T; k++) {
            const int32_t val = (int32_t)((tmp >> (32 - (k + 1) * ORC_BASEBIT)) & mask) - halfbase;
            if (val == 0) continue;
            const uint32_t* row = ksk + (((size_t)j * ORC_T + k) * ORC_KS_NUMBASE + (size_t)(abs(val) - 1)) * ORC_LVL0_WORDS;
            if (val > 0) for (int i = 0; i <= ORC_n; i++) lwe[i] -= row[i];
            else for (int i = 0; i <= ORC_n; i++) lwe[i] += row[i];
        }
    }
}

/* ------------------------------------------------------------------ */
/* TRLWE-level primitives                                             */
/* ------------------------------------------------------------------ */
/* __CMUXNTT__ with TRLWESubAndDecomposition, src/bootstrap_gpu.cu:162-285 */
void orc_cmux(uint32_t* res, const uint32_t* trgsw, const uint32_t* c1, const uint32_t* c0)
{
    const uint32_t decomp_mask = (1u << ORC_BGBIT) - 1;
    const int32_t decomp_half = 1 << (ORC_BGBIT - 1);
    uint32_t decomp_offset = 0;
    for (int i = 1; i <= ORC_L; i++) decomp_offset += (uint32_t)(1u << (ORC_BGBIT - 1)) << (32 - i * ORC_BGBIT);
    const uint32_t roundoffset = 1u << (32 - ORC_L * ORC_BGBIT - 1);
    ntt_tables();
    uint64_t accum[(ORC_K + 1) * ORC_N], work[ORC_N], key[ORC_N];
    memset(accum, 0, sizeof(accum));
    for (int j = 0; j <= ORC_K; j++)
        for (int digit = 0; digit < ORC_L; digit++) {
            for (int i = 0; i < ORC_N; i++) {
                const uint32_t temp = c1[j * ORC_N + i] - c0[j * ORC_N + i] + decomp_offset + roundoffset;
                work[i] = ffp_from_i32((int32_t)((temp >> (32 - (digit + 1) * ORC_BGBIT)) & decomp_mask) - decomp_half);
            }
            orc_ntt_forward(work);
            for (int out_k = 0; out_k <= ORC_K; out_k++) {
                const uint32_t* row = trgsw + ((size_t)(j * ORC_L + digit) * (ORC_K + 1) + out_k) * ORC_N;
                for (int i = 0; i < ORC_N; i++) key[i] = row[i];
                orc_ntt_forward(key);                    /* TRGSW2NTT, src/bootstrap_gpu.cu:75-94 */
                uint64_t* ac = accum + out_k * ORC_N;
                for (int i = 0; i < ORC_N; i++) ac[i] = mod_add(ac[i], barrett_mult(work[i], key[i]));
            }
        }
    for (int k = 0; k <= ORC_K; k++) {
        uint64_t* ac = accum + k * ORC_N;
        orc_ntt_inverse(ac);
        for (int i = 0; i < ORC_N; i++) res[k * ORC_N + i] = c0[k * ORC_N + i] + ffp_lift_u32(ac[i]);
    }
}

void orc_sample_extract_keyswitch(const orc_evalkey* ek, uint32_t* tlwe0, const uint32_t* trlwe)
{
    uint32_t t1[ORC_LVL1_WORDS];
    orc_sample_extract0(t1, trlwe);
    orc_keyswitch(ek, tlwe0, t1);
}

void orc_refresh(const orc_evalkey* ek, uint32_t* trlwe_out, const uint32_t* trlwe_in)
{
    uint32_t t0[ORC_LVL0_WORDS];
    orc_sample_extract_keyswitch(ek, t0, trlwe_in);
    orc_blind_rotate(ek, trlwe_out, t0, -1);
}

/* ------------------------------------------------------------------ */
/* Gates                                                              */
/* ------------------------------------------------------------------ */
/* (ca, cb, offset/mu): src/bootstrap_gpu.cu:424-512 (br->iks) and :591-679 (iks->br) */
void orc_gate_coeffs(int op, int* ca, int* cb, int* off_mu)
{
    static const int tab[10][3] = {
        {-1, -1, 1},   /* NAND  :430 */
        {-1, -1, -1},  /* NOR   :439 */
        {-2, -2, -2},  /* XNOR  :448 */
        {1, 1, -1},    /* AND   :457 */
        {1, 1, 1},     /* OR    :466 */
        {2, 2, 2},     /* XOR   :475 */
        {-1, 1, -1},   /* ANDNY :484 */
        {1, -1, -1},   /* ANDYN :493 */
        {-1, 1, 1},    /* ORNY  :502 */
        {1, -1, 1},    /* ORYN  :511 */
    };
    *ca = tab[op][0]; *cb = tab[op][1]; *off_mu = tab[op][2];
}

int orc_truth(int op, int a, int b, int c)
{
    switch (op) {
    case ORC_NAND: return !(a && b);
    case ORC_NOR: return !(a || b);
    case ORC_XNOR: return a == b;
    case ORC_AND: return a && b;
    case ORC_OR: return a || b;
    case ORC_XOR: return a != b;
    case ORC_ANDNY: return !a && b;
    case ORC_ANDYN: return a && !b;
    case ORC_ORNY: return !a || b;
    case ORC_ORYN: return a || !b;
    case ORC_MUX: return a ? b : c;
    case ORC_NMUX: return !(a ? b : c);
    case ORC_NOT: return !a;
    case ORC_COPY: return a;
    }
    return -1;
}

static void lincomb(uint32_t* out, int words, int ca, const uint32_t* a, int cb, const uint32_t* b, uint32_t off)
{
    for (int i = 0; i < words; i++) out[i] = (uint32_t)ca * a[i] + (uint32_t)cb * b[i];
    out[words - 1] += off;
}

/* lvl0 ctxts: __HomGate__ br->iks, src/bootstrap_gpu.cu:402-421 */
static void bootstrap_lvl0_to_tlwe1(const orc_evalkey* ek, uint32_t* tlwe1, int ca, const uint32_t* in0,
                                    int cb, const uint32_t* in1, uint32_t off)
{
    uint32_t c[ORC_LVL0_WORDS], acc[ORC_TRLWE_WORDS];
    lincomb(c, ORC_LVL0_WORDS, ca, in0, cb, in1, off);
    orc_blind_rotate(ek, acc, c, -1);
    orc_sample_extract0(tlwe1, acc);
}
/* lvl1 ctxts: __HomGate__ iks->br, src/bootstrap_gpu.cu:383-400
 * (IdentityKeySwitchPreAdd include/keyswitch_gpu.cuh:136-188 == pre-add then key switch) */
static void bootstrap_lvl1_to_tlwe1(const orc_evalkey* ek, uint32_t* tlwe1, int ca, const uint32_t* in0,
                                    int cb, const uint32_t* in1, uint32_t off)
{
    uint32_t c[ORC_LVL1_WORDS], t0[ORC_LVL0_WORDS], acc[ORC_TRLWE_WORDS];
    lincomb(c, ORC_LVL1_WORDS, ca, in0, cb, in1, off);
    orc_keyswitch(ek, t0, c);
    orc_blind_rotate(ek, acc, t0, -1);
    orc_sample_extract0(tlwe1, acc);
}

void orc_gate(const orc_evalkey* ek, int op, int level, uint32_t* out,
              const uint32_t* in0, const uint32_t* in1, const uint32_t* in2)
{
    const int words = level ? ORC_LVL1_WORDS : ORC_LVL0_WORDS;
    if (op == ORC_NOT || op == ORC_COPY) {          /* src/bootstrap_gpu.cu:681-703 */
        for (int i = 0; i < words; i++) out[i] = (op == ORC_NOT) ? (uint32_t)(0u - in0[i]) : in0[i];
        return;
    }
    if (op == ORC_MUX || op == ORC_NMUX) {
        uint32_t t1[ORC_LVL1_WORDS], t0[ORC_LVL1_WORDS], s[ORC_LVL1_WORDS];
        const uint32_t negmu = (uint32_t)(0u - ORC_MU);
        if (level == 0) {                           /* :515-588 */
            bootstrap_lvl0_to_tlwe1(ek, t1, 1, in0, 1, in1, negmu);
            bootstrap_lvl0_to_tlwe1(ek, t0, -1, in0, 1, in2, negmu);
        } else {                                    /* :706-780 */
            bootstrap_lvl1_to_tlwe1(ek, t1, 1, in0, 1, in1, negmu);
            bootstrap_lvl1_to_tlwe1(ek, t0, -1, in0, 1, in2, negmu);
        }
        if (op == ORC_MUX) lincomb(s, ORC_LVL1_WORDS, 1, t1, 1, t0, ORC_MU);
        else lincomb(s, ORC_LVL1_WORDS, -1, t1, -1, t0, negmu);
        if (level == 0) orc_keyswitch(ek, out, s);
        else memcpy(out, s, sizeof(s));
        return;
    }
    int ca, cb, om;
    orc_gate_coeffs(op, &ca, &cb, &om);
    const uint32_t off = (uint32_t)om * ORC_MU;
    uint32_t t[ORC_LVL1_WORDS];
    if (level == 0) {
        bootstrap_lvl0_to_tlwe1(ek, t, ca, in0, cb, in1, off);
        orc_keyswitch(ek, out, t);
    } else {
        bootstrap_lvl1_to_tlwe1(ek, t, ca, in0, cb, in1, off);
        memcpy(out, t, sizeof(t));
    }
}

const char* orc_get_params(int* out7)
{
    const int v[7] = {ORC_n, ORC_NBIT, ORC_K, ORC_L, ORC_BGBIT, ORC_T, ORC_BASEBIT};
    if (out7) memcpy(out7, v, sizeof(v));
    return ORC_SET_NAME;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_gate_batch(const orc_evalkey* ek, const int32_t* ops, int ops_stride, int level,
                    size_t count, uint32_t* out, const uint32_t* in0,
                    const uint32_t* in1, const uint32_t* in2, int threads)
{
    const size_t w = level ? ORC_LVL1_WORDS : ORC_LVL0_WORDS;
    ntt_tables();
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (long g = 0; g < (long)count; g++) {
        orc_gate(ek, ops[(size_t)g * ops_stride], level, out + g * w, in0 + g * w,
                 in1 ? in1 + g * w : NULL, in2 ? in2 + g * w : NULL);
    }
}
