/*
 * CPU oracle (test infrastructure only) — see tfhe_oracle.h for scope and parity status.
 *
 * The arithmetic below restates concrete-core 1.x (not vendored by the reference):
 *   - SignedDecomposer::closest_representable / decompose_one_level
 *   - pbs_modulus_switch
 *   - LweKeyswitchKey::keyswitch_ciphertext
 *   - FourierLweBootstrapKey::blind_rotate / cmux / add_external_product_assign
 *   - Fft::forward_as_{torus,integer} / add_backward_as_torus and from_torus
 *   - GlweCiphertext::fill_lwe_with_sample_extraction (n_th = 0)
 * The reference reaches them at concrete-shortint/src/engine/server_side/mod.rs:104-116,
 * :141-153 and concrete-boolean/src/engine/bootstrapping/cpu.rs:180-185,206-210,231-254.
 */
/* Floating-point contract: the file is compiled with -ffp-contract=off and every fused
 * multiply-add is written as an explicit fma(), so that the sequence of IEEE operations is
 * fixed by the source. The workgroup-per-ciphertext GPU kernel performs the same sequence,
 * which is what makes its ciphertexts comparable bit for bit. */
#include "tfhe_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ rng -- */
static uint64_t splitmix64(uint64_t* x) {
  uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void orc_rng_seed(orc_rng* r, uint64_t seed) {
  uint64_t x = seed;
  for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&x);
}
uint64_t orc_rng_next(orc_rng* r) {
  uint64_t* s = r->s;
  const uint64_t result = rotl64(s[1] * 5, 7) * 9;
  const uint64_t t = s[1] << 17;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl64(s[3], 45);
  return result;
}
/* Box-Muller, one normal deviate per call (two 53-bit uniforms consumed). */
double orc_rng_gauss(orc_rng* r) {
  const double u1 = ((double)(orc_rng_next(r) >> 11) + 1.0) * 0x1.0p-53; /* (0,1] */
  const double u2 = (double)(orc_rng_next(r) >> 11) * 0x1.0p-53;         /* [0,1) */
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

/* --------------------------------------------------------------- scalars -- */
uint64_t orc_closest_representable(uint64_t x, uint32_t base_log, uint32_t level, uint32_t bits) {
  const uint32_t non_rep = bits - base_log * level;
  if (bits < 64) x &= ((1ULL << bits) - 1);
  if (non_rep == 0) return x;
  const uint64_t msb = (x >> (non_rep - 1)) & 1ULL;
  uint64_t res = (x >> non_rep) + msb;
  res <<= non_rep;
  if (bits < 64) res &= ((1ULL << bits) - 1);
  return res;
}

/* one pass of the signed decomposition state machine; returns the digit as a
 * two's-complement value in the `bits`-wide word (sign-extended to 64 here). */
static inline int64_t decompose_one_level(uint32_t base_log, uint64_t* state, uint64_t mod_b_mask) {
  uint64_t res = *state & mod_b_mask;
  *state >>= base_log;
  uint64_t carry = ((res - 1ULL) | *state) & res;
  carry >>= (base_log - 1);
  *state += carry;
  return (int64_t)(res - (carry << base_log));
}

void orc_decompose(uint64_t x, uint32_t base_log, uint32_t level, uint32_t bits, int64_t* digits) {
  const uint32_t non_rep = bits - base_log * level;
  uint64_t state = orc_closest_representable(x, base_log, level, bits) >> non_rep;
  const uint64_t mask = (1ULL << base_log) - 1ULL;
  for (uint32_t l = level; l >= 1; l--) digits[l - 1] = decompose_one_level(base_log, &state, mask);
}

uint32_t orc_modswitch(uint64_t x, uint32_t log2_poly_size, uint32_t bits) {
  if (bits < 64) x &= ((1ULL << bits) - 1);
  uint64_t out = x >> (bits - log2_poly_size - 2);
  out += out & 1ULL;
  out >>= 1;
  return (uint32_t)out;
}

/* ----------------------------------------------------------- accumulators -- */
void orc_generate_accumulator_u64(const orc_params* p, const uint64_t* f_values, uint64_t* glwe_out) {
  const size_t N = p->polynomial_size, k = p->glwe_dimension;
  const size_t modulus_sup = (size_t)p->message_modulus * p->carry_modulus;
  const size_t box = N / modulus_sup;
  const uint64_t delta = (1ULL << 63) / (uint64_t)modulus_sup;
  uint64_t* tmp = (uint64_t*)calloc(N, sizeof(uint64_t));
  for (size_t i = 0; i < modulus_sup; i++)
    for (size_t j = 0; j < box; j++) tmp[i * box + j] = f_values[i] * delta;
  const size_t half = box / 2;
  for (size_t j = 0; j < half; j++) tmp[j] = (uint64_t)0 - tmp[j];
  memset(glwe_out, 0, k * N * sizeof(uint64_t));
  uint64_t* body = glwe_out + k * N;
  for (size_t j = 0; j < N; j++) body[j] = tmp[(j + half) % N]; /* rotate_left(half) */
  free(tmp);
}

void orc_boolean_accumulator_u32(const orc_params* p, uint32_t* glwe_out) {
  const size_t N = p->polynomial_size, k = p->glwe_dimension;
  memset(glwe_out, 0, k * N * sizeof(uint32_t));
  for (size_t j = 0; j < N; j++) glwe_out[k * N + j] = 1u << 29;
}

/* -------------------------------------------------------------------- fft -- */
/* cos and sin are called through volatile pointers so that no compiler merges the pair into a
 * sincos() call: the twiddle values are then the same libm results in every build of this file
 * and of the GPU library, which is part of the bit-for-bit contract stated at the top. */
static double (*volatile orc_cos)(double) = cos;
static double (*volatile orc_sin)(double) = sin;

struct orc_fft_plan {
  uint32_t N, M, logM;
  double *tw_re, *tw_im; /* twist e^{i pi j / N}, j < M */
  double *st_re, *st_im; /* stage twiddles e^{-2 pi i j / len}, stages len = M, M/2, ..., 2 */
};

orc_fft_plan* orc_fft_plan_create(uint32_t N) {
  orc_fft_plan* pl = (orc_fft_plan*)calloc(1, sizeof(*pl));
  pl->N = N;
  pl->M = N / 2;
  uint32_t lg = 0;
  while ((1u << lg) < pl->M) lg++;
  pl->logM = lg;
  const uint32_t M = pl->M;
  pl->tw_re = (double*)aligned_alloc(64, sizeof(double) * M);
  pl->tw_im = (double*)aligned_alloc(64, sizeof(double) * M);
  for (uint32_t j = 0; j < M; j++) {
    const double a = M_PI * (double)j / (double)N;
    pl->tw_re[j] = orc_cos(a);
    pl->tw_im[j] = orc_sin(a);
  }
  const size_t st = (M > 1) ? M : 1;
  pl->st_re = (double*)aligned_alloc(64, sizeof(double) * ((st + 7) / 8 * 8));
  pl->st_im = (double*)aligned_alloc(64, sizeof(double) * ((st + 7) / 8 * 8));
  size_t off = 0;
  for (uint32_t len = M; len >= 2; len >>= 1) {
    const uint32_t half = len / 2;
    for (uint32_t j = 0; j < half; j++) {
      const double a = -2.0 * M_PI * (double)j / (double)len;
      pl->st_re[off + j] = orc_cos(a);
      pl->st_im[off + j] = orc_sin(a);
    }
    off += half;
  }
  return pl;
}
void orc_fft_plan_destroy(orc_fft_plan* pl) {
  if (!pl) return;
  free(pl->tw_re); free(pl->tw_im); free(pl->st_re); free(pl->st_im);
  free(pl);
}

/* in-place decimation-in-frequency, natural order in, bit-reversed order out */
static void fft_dif(const orc_fft_plan* pl, double* restrict re, double* restrict im) {
  const uint32_t M = pl->M;
  size_t off = 0;
  for (uint32_t len = M; len >= 2; len >>= 1) {
    const uint32_t half = len / 2;
    const double* restrict wr = pl->st_re + off;
    const double* restrict wi = pl->st_im + off;
    for (uint32_t blk = 0; blk < M; blk += len) {
      double* restrict ar = re + blk;
      double* restrict ai = im + blk;
      double* restrict br = re + blk + half;
      double* restrict bi = im + blk + half;
      for (uint32_t j = 0; j < half; j++) {
        const double xr = ar[j], xi = ai[j], yr = br[j], yi = bi[j];
        const double dr = xr - yr, di = xi - yi;
        ar[j] = xr + yr;
        ai[j] = xi + yi;
        br[j] = fma(dr, wr[j], -(di * wi[j]));
        bi[j] = fma(dr, wi[j], di * wr[j]);
      }
    }
    off += half;
  }
}
/* in-place decimation-in-time inverse (unnormalised), bit-reversed in, natural out */
static void fft_dit_inv(const orc_fft_plan* pl, double* restrict re, double* restrict im) {
  const uint32_t M = pl->M;
  size_t off = (M > 1) ? M - 1 : 0;
  for (uint32_t len = 2; len <= M; len <<= 1) {
    const uint32_t half = len / 2;
    off -= half;
    const double* restrict wr = pl->st_re + off;
    const double* restrict wi = pl->st_im + off;
    for (uint32_t blk = 0; blk < M; blk += len) {
      double* restrict ar = re + blk;
      double* restrict ai = im + blk;
      double* restrict br = re + blk + half;
      double* restrict bi = im + blk + half;
      for (uint32_t j = 0; j < half; j++) {
        /* multiply by the conjugate twiddle */
        const double yr = fma(br[j], wr[j], bi[j] * wi[j]);
        const double yi = fma(bi[j], wr[j], -(br[j] * wi[j]));
        const double xr = ar[j], xi = ai[j];
        ar[j] = xr + yr;
        ai[j] = xi + yi;
        br[j] = xr - yr;
        bi[j] = xi - yi;
      }
    }
  }
}

void orc_negacyclic_schoolbook_u64(uint32_t n, const uint64_t* a, const int64_t* b, uint64_t* out) {
  for (uint32_t i = 0; i < n; i++) out[i] = 0;
  for (uint32_t i = 0; i < n; i++)
    for (uint32_t j = 0; j < n; j++) {
      const uint64_t prod = a[i] * (uint64_t)b[j];
      const uint32_t d = i + j;
      if (d < n) out[d] += prod; else out[d - n] -= prod;
    }
}

/* f64::round of Rust / round() of C: nearest, ties away from zero. Written with trunc so
 * that the compiler can inline it (vroundsd) instead of calling libm per coefficient. */
static inline double orc_round_half_away(double x) {
  const double t = trunc(x);
  const double d = x - t; /* exact */
  return (fabs(d) >= 0.5) ? t + copysign(1.0, x) : t;
}

/* ------------------------------------------------ generic body, two widths -- */
#define T uint64_t
#define ST int64_t
#define BITS 64
#define TWO_POW_BITS 0x1.0p64
#define FN(name) name##_u64
#include "tfhe_oracle_body.inc"
#undef T
#undef ST
#undef BITS
#undef TWO_POW_BITS
#undef FN

#define T uint32_t
#define ST int32_t
#define BITS 32
#define TWO_POW_BITS 0x1.0p32
#define FN(name) name##_u32
#include "tfhe_oracle_body.inc"
#undef T
#undef ST
#undef BITS
#undef TWO_POW_BITS
#undef FN
