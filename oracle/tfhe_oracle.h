/*
 * CPU oracle for the TFHE keyswitch + programmable-bootstrap path.
 *
 * THIS IS TEST INFRASTRUCTURE. It is a plain-C restatement of the algorithm the
 * reference reaches through concrete-core 1.x (an un-vendored crates.io
 * dependency, requirement "1.0.0", absent from the reference tree). Only tests,
 * the smoke check and the CPU-baseline leg of the benchmark may link or load
 * it; the product library never does.
 *
 * Parity status: "parity unpinned" at ciphertext level. The reference holds no
 * golden vectors or known-answer tests for this path (its tests are randomized
 * decrypt-equality checks), and its arithmetic lives in a crate that cannot be
 * built here. The oracle is therefore pinned by (1) the integer specifications
 * that ARE in the reference tree (accumulator construction, encodings, decrypt
 * rounding, gate linear combinations), (2) algebraic known-answer tests that do
 * not depend on any implementation (trivial-ciphertext bootstrap, noise-free
 * keys, FFT vs schoolbook product, decomposer identities) and (3) the
 * reference's own decrypt-level test properties re-run on it.
 *
 * Reference call sites restated here:
 *   concrete-shortint/src/engine/server_side/mod.rs:31-72   server key generation
 *   concrete-shortint/src/engine/server_side/mod.rs:95-155  keyswitch then bootstrap
 *   concrete-shortint/src/engine/mod.rs:130-178             accumulator (LUT) construction
 *   concrete-shortint/src/engine/client_side.rs:9-144       client key, encrypt, decrypt
 *   concrete-boolean/src/engine/bootstrapping/cpu.rs:24-64,100-142,159-260
 *   concrete-boolean/src/engine/mod.rs:107-147,364-405      boolean encode / NAND
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- deterministic generator (xoshiro256** seeded with splitmix64) ---- */
typedef struct { uint64_t s[4]; } orc_rng;
void     orc_rng_seed(orc_rng* r, uint64_t seed);
uint64_t orc_rng_next(orc_rng* r);
double   orc_rng_gauss(orc_rng* r);

/* ---- parameter set (shortint Parameters / BooleanParameters) ---- */
typedef struct {
  uint32_t lwe_dimension;      /* n  */
  uint32_t glwe_dimension;     /* k  */
  uint32_t polynomial_size;    /* N  */
  uint32_t pbs_base_log;
  uint32_t pbs_level;
  uint32_t ks_base_log;
  uint32_t ks_level;
  uint32_t message_modulus;    /* 0 for boolean */
  uint32_t carry_modulus;      /* 0 for boolean */
  double   lwe_modular_std_dev;
  double   glwe_modular_std_dev;
} orc_params;

/* ---- scalar pieces (bit-width independent forms take/return u64) ---- */
/* closest_representable and one signed-decomposition pass; digits[level-1]. */
uint64_t orc_closest_representable(uint64_t x, uint32_t base_log, uint32_t level, uint32_t bits);
void     orc_decompose(uint64_t x, uint32_t base_log, uint32_t level, uint32_t bits, int64_t* digits);
/* pbs modulus switch: value in [0, 2N]. */
uint32_t orc_modswitch(uint64_t x, uint32_t log2_poly_size, uint32_t bits);

/* ---- accumulator (look-up table) construction: engine/mod.rs:130-178 ---- */
/* f_values[i] = f(i) for i in [0, message_modulus*carry_modulus). Output: trivial
 * GLWE, (k+1)*N words, mask polynomials zero, body polynomial last. */
void orc_generate_accumulator_u64(const orc_params* p, const uint64_t* f_values, uint64_t* glwe_out);
/* boolean accumulator: body filled with 2^29 (cpu.rs:49-53). */
void orc_boolean_accumulator_u32(const orc_params* p, uint32_t* glwe_out);

/* ---- FFT plan for the negacyclic transform of size N ---- */
typedef struct orc_fft_plan orc_fft_plan;
orc_fft_plan* orc_fft_plan_create(uint32_t polynomial_size);
void          orc_fft_plan_destroy(orc_fft_plan* plan);
/* Fourier polynomial = N/2 complex numbers stored re[0..N/2) then im[0..N/2). */
void orc_fft_forward_as_integer_u64(const orc_fft_plan*, const uint64_t* poly, double* fourier);
void orc_fft_forward_as_torus_u64(const orc_fft_plan*, const uint64_t* poly, double* fourier);
void orc_fft_add_backward_as_torus_u64(const orc_fft_plan*, const double* fourier, uint64_t* poly);
void orc_fft_forward_as_integer_u32(const orc_fft_plan*, const uint32_t* poly, double* fourier);
void orc_fft_forward_as_torus_u32(const orc_fft_plan*, const uint32_t* poly, double* fourier);
void orc_fft_add_backward_as_torus_u32(const orc_fft_plan*, const double* fourier, uint32_t* poly);
/* exact negacyclic product out = a * b mod (X^N + 1), wrapping; b small signed. */
void orc_negacyclic_schoolbook_u64(uint32_t n, const uint64_t* a, const int64_t* b, uint64_t* out);

/* ---- the generic API, instantiated for u64 and u32 torus words ---- */
#define ORC_DECLARE(T, S)                                                                         \
  void orc_gen_binary_key_##S(orc_rng* rng, size_t len, T* key);                                  \
  void orc_lwe_encrypt_##S(orc_rng* rng, const T* key, size_t dim, T plaintext, double std_dev,   \
                           T* ct_out);                                                            \
  T    orc_lwe_phase_##S(const T* key, size_t dim, const T* ct);                                  \
  /* KSK[in_dim][level][out_dim+1], level 1 (coarsest) first. */                                  \
  void orc_gen_ksk_##S(orc_rng* rng, const T* in_key, size_t in_dim, const T* out_key,            \
                       size_t out_dim, uint32_t base_log, uint32_t level, double std_dev,         \
                       T* ksk_out);                                                               \
  /* BSK[n][level][k+1 rows][k+1 cols][N], level 1 first; glwe_key = k*N bits. */                 \
  void orc_gen_bsk_##S(orc_rng* rng, const orc_params* p, const T* lwe_key, const T* glwe_key,    \
                       T* bsk_out);                                                               \
  void orc_keyswitch_##S(const T* ksk, size_t in_dim, size_t out_dim, uint32_t base_log,          \
                         uint32_t level, const T* ct_in, T* ct_out);                              \
  /* fourier BSK: [n][level][row][col][N] doubles (re half then im half per polynomial). */       \
  void orc_bsk_to_fourier_##S(const orc_fft_plan* plan, const orc_params* p, const T* bsk,        \
                              double* fourier_out);                                               \
  /* programmable bootstrap: in n+1 words, out kN+1 words. */                                     \
  void orc_bootstrap_##S(const orc_fft_plan* plan, const orc_params* p, const double* fourier_bsk,\
                         const T* accumulator, const T* ct_in, T* ct_out);                        \
  /* blind rotation only: out is the (k+1)*N-word GLWE before sample extraction. */               \
  void orc_blind_rotate_##S(const orc_fft_plan* plan, const orc_params* p,                        \
                            const double* fourier_bsk, const T* accumulator, const T* ct_in,      \
                            T* glwe_out);                                                         \
  void orc_sample_extract_##S(const orc_params* p, const T* glwe, T* lwe_out);                    \
  /* batches over `count` ciphertexts with `threads` worker threads.                              \
   * lut_index[i] selects accumulators + lut_index[i]*(k+1)*N (NULL = all zero). */               \
  void orc_ks_pbs_batch_##S(const orc_fft_plan* plan, const orc_params* p, const T* ksk,          \
                            const double* fourier_bsk, const T* accumulators,                     \
                            const uint32_t* lut_index, const T* ct_in, T* ct_out, size_t count,   \
                            int threads);                                                         \
  void orc_pbs_ks_batch_##S(const orc_fft_plan* plan, const orc_params* p, const T* ksk,          \
                            const double* fourier_bsk, const T* accumulators,                     \
                            const uint32_t* lut_index, const T* ct_in, T* ct_out, size_t count,   \
                            int threads);

ORC_DECLARE(uint64_t, u64)
ORC_DECLARE(uint32_t, u32)

#ifdef __cplusplus
}
#endif
#endif
