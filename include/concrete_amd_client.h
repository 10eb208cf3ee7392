/*
 * concrete_amd — host-side client operations (no GPU involved): secret keys, server-key
 * material in the STANDARD domain, LWE encryption and decryption, look-up-table construction.
 *
 * These mirror what the reference does on the client through concrete-core's DefaultEngine;
 * they exist so that a caller (tests, the benchmark, the host-side mirror of the reference
 * API) can produce the inputs of the GPU path. The generator has two kinds:
 *   - camd_rng_seed_secure: ChaCha20 keyed with 256 bits from the operating system (getrandom);
 *     secret keys, masks and noise of real keys come from this one, as the reference draws them
 *     from concrete-csprng seeded by the OS (concrete-shortint/src/engine/mod.rs:15-31);
 *   - camd_rng_seed: a seedable xoshiro256** sequence, deterministic and NOT cryptographically
 *     secure: for tests and benchmarks that need reproducible keys, never for keys that protect data.
 *
 * Every function returns 0 on success, a negative CAMD_E* code otherwise.
 */
#ifndef CONCRETE_AMD_CLIENT_H
#define CONCRETE_AMD_CLIENT_H

#include "concrete_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct camd_rng {
  uint64_t s[4];      /* xoshiro256** state, or the ChaCha20 key of the secure kind */
  uint64_t counter;   /* secure kind: number of the next 64-byte block */
  uint32_t secure;    /* 0 = seeded deterministic sequence, 1 = ChaCha20 keyed from the OS */
  uint32_t have;      /* secure kind: unread 64-bit words left in `block` */
  uint64_t block[8];
} camd_rng;

/* deterministic sequence from a 64-bit seed: tests and benchmarks only */
void camd_rng_seed(camd_rng* rng, uint64_t seed);
/* ChaCha20 keyed with 256 bits of operating-system entropy */
int camd_rng_seed_secure(camd_rng* rng);

/* Uniform binary secret key of `len` words (each 0 or 1) in the torus width `torus_bits`.
 * Replaces generate_new_lwe_secret_key / generate_new_glwe_secret_key
 * (concrete-shortint/src/engine/client_side.rs:11-18, concrete-boolean/src/engine/mod.rs:86-95). */
int camd_client_gen_binary_key(camd_rng* rng, uint32_t torus_bits, size_t len, void* key);

/* Standard-domain bootstrap key [n][level][k+1][k+1][N], encrypting each bit of `lwe_key`
 * (n words) under `glwe_key` (k*N words). Replaces generate_new_lwe_bootstrap_key
 * (concrete-shortint/src/engine/server_side/mod.rs:39-45). `threads` worker threads. */
int camd_client_gen_bootstrap_key(camd_rng* rng, const camd_params* params, double glwe_std_dev,
                                  const void* lwe_key, const void* glwe_key, void* bsk_out, int threads);

/* Keyswitch key [in_dim][level][out_dim+1] from `in_key` to `out_key`. Replaces
 * generate_new_lwe_keyswitch_key (server_side/mod.rs:56-62). */
int camd_client_gen_keyswitch_key(camd_rng* rng, uint32_t torus_bits, const void* in_key, size_t in_dim,
                                  const void* out_key, size_t out_dim, uint32_t base_log, uint32_t level,
                                  double lwe_std_dev, void* ksk_out);

/* LWE encryption of `count` plaintext words under `key` (dim words): mask uniform,
 * body = <mask, key> + plaintext + gaussian(std_dev). ct_out: [count][dim+1]. Replaces
 * encrypt_lwe_ciphertext (client_side.rs:73-75; boolean engine/mod.rs:119-122). */
int camd_client_lwe_encrypt(camd_rng* rng, uint32_t torus_bits, const void* key, size_t dim,
                            const void* plaintexts, size_t count, double std_dev, void* ct_out);

/* phase = body - <mask, key> for `count` ciphertexts. Replaces decrypt_lwe_ciphertext
 * (client_side.rs:120-126); the caller applies the decoding of its scheme. */
int camd_client_lwe_phase(uint32_t torus_bits, const void* key, size_t dim, const void* ct, size_t count,
                          void* phases_out);

/* shortint look-up table: f_values[i] = f(i), i < message_modulus*carry_modulus. Output is a
 * trivial GLWE of (k+1)*N 64-bit words. Replaces ShortintEngine::generate_accumulator_with_engine
 * (concrete-shortint/src/engine/mod.rs:130-178). */
int camd_client_shortint_accumulator(uint32_t glwe_dimension, uint32_t polynomial_size,
                                     uint32_t message_modulus, uint32_t carry_modulus,
                                     const uint64_t* f_values, uint64_t* glwe_out);

/* boolean look-up table: body filled with 1/8 (2^29), masks zero. Replaces Memory::as_buffers
 * (concrete-boolean/src/engine/bootstrapping/cpu.rs:49-53). */
int camd_client_boolean_accumulator(uint32_t glwe_dimension, uint32_t polynomial_size, uint32_t* glwe_out);

#ifdef __cplusplus
}
#endif
#endif
