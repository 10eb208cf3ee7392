/* Plain-C caller of the C ABI: the shape a Rust/C host would have. Reads a parameter set, keys,
 * one look-up table and a batch of ciphertexts from a flat binary file, runs keyswitch + bootstrap
 * through camd_ks_pbs_batch, writes the output ciphertexts. No Python, no torch in the process.
 *
 * file layout (little endian): 8 x uint32 camd_params, uint64 count, then bsk words, ksk words,
 * (k+1)N look-up-table words, count x (kN+1) ciphertext words (64-bit torus). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "concrete_amd.h"

static void* read_block(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "short read (%zu bytes)\n", bytes);
    exit(2);
  }
  return p;
}

#define CHECK(call)                                                      \
  do {                                                                   \
    int rc_ = (call);                                                    \
    if (rc_ != CAMD_OK) {                                                \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, camd_last_error()); \
      return 1;                                                          \
    }                                                                    \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s <input file> <output file>\n", argv[0]);
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  camd_params p;
  uint64_t count;
  if (fread(&p, sizeof p, 1, f) != 1 || fread(&count, sizeof count, 1, f) != 1) return 2;
  const size_t n = p.lwe_dimension, k = p.glwe_dimension, N = p.polynomial_size;
  const size_t bsk_words = n * p.pbs_level * (k + 1) * (k + 1) * N;
  const size_t ksk_words = k * N * p.ks_level * (n + 1);
  const size_t big = k * N + 1;
  uint64_t* bsk = read_block(f, bsk_words * 8);
  uint64_t* ksk = read_block(f, ksk_words * 8);
  uint64_t* lut = read_block(f, (k + 1) * N * 8);
  uint64_t* cts = read_block(f, count * big * 8);
  fclose(f);

  camd_context* ctx = NULL;
  camd_server_key* key = NULL;
  CHECK(camd_context_create(0, &ctx));
  CHECK(camd_server_key_create(ctx, &p, bsk, ksk, CAMD_PBS_AUTO, &key));
  uint32_t lut_id = 99;
  CHECK(camd_lut_upload(key, lut, &lut_id));
  uint64_t* out = malloc(count * big * 8);
  CHECK(camd_ks_pbs_batch(key, cts, NULL, out, (size_t)count));
  /* error path: a look-up table id that does not exist must be refused, not crash */
  uint32_t* bad = malloc(count * sizeof *bad);
  for (uint64_t i = 0; i < count; i++) bad[i] = 7;
  if (camd_ks_pbs_batch(key, cts, bad, out + 0, (size_t)count) == CAMD_OK) {
    fprintf(stderr, "an unknown look-up table id was accepted\n");
    return 1;
  }
  CHECK(camd_ks_pbs_batch(key, cts, NULL, out, (size_t)count));
  int variant = -1;
  CHECK(camd_server_key_variant(key, &variant));
  CHECK(camd_server_key_destroy(key));
  CHECK(camd_context_destroy(ctx));

  f = fopen(argv[2], "wb");
  if (!f || fwrite(out, 8, count * big, f) != count * big) return 2;
  fclose(f);
  printf("%s: %llu ciphertexts, look-up table %u, kernel variant %d\n", camd_version(), (unsigned long long)count, lut_id,
         variant);
  return 0;
}
