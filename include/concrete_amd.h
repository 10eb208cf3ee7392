/*
 * concrete_amd — C ABI of the MI355X (gfx950) keyswitch + programmable-bootstrap backend.
 *
 * The reference (ryan-levatel/concrete) has no FFI for this path: its boundary is the Rust
 * interface of concrete-core engines called from concrete-shortint and concrete-boolean.
 * Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - every function returns 0 on success and a negative CAMD_E* code on failure; it never
 *     throws, aborts or prints. camd_last_error() returns the calling thread's last message.
 *   - "host" entry points take host pointers (pageable or pinned), block until the result is
 *     in the output buffer, and may be called concurrently on one key: calls that arrive
 *     while the key is busy wait and are then run together, as one batch per operation (the
 *     reference calls this path one ciphertext at a time from many threads; here such calls
 *     share a launch). Each caller gets its own status and error message. A large host batch
 *     is cut into slices of 2048 ciphertexts (environment variable CAMD_HOST_CHUNK, 0 = one
 *     slice) so that the PCIe copies of one slice run beside the kernels of its neighbours;
 *     results do not depend on the slicing.
 *   - "device" entry points take device pointers on the key's GPU plus a hipStream_t (as
 *     void*, NULL = the default stream), enqueue the kernels and return without synchronising.
 *   - LWE ciphertexts are dim+1 torus words, body last. GLWE ciphertexts (look-up tables) are
 *     (k+1)*N words, mask polynomials first, body polynomial last.
 *   - standard bootstrap key layout: [n][level][k+1 rows][k+1 cols][N] words, level 1
 *     (coarsest, q/B) first. keyswitch key layout: [kN][level][n+1] words, level 1 first.
 *   - the torus word is uint64_t for shortint parameter sets and uint32_t for boolean ones.
 */
#ifndef CONCRETE_AMD_H
#define CONCRETE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAMD_OK 0
#define CAMD_EINVAL (-1)   /* bad argument (null pointer, unsupported parameter set, bad id) */
#define CAMD_EHIP (-2)     /* a HIP runtime call failed; message holds hipGetErrorString */
#define CAMD_ENOMEM (-3)   /* host or device allocation failed */
#define CAMD_ENODEV (-4)   /* no usable gfx950 device */

typedef struct camd_context camd_context;
typedef struct camd_server_key camd_server_key;

/* The cryptographic shape of a server key.
 * shortint: concrete-shortint/src/parameters/mod.rs:28-45; boolean:
 * concrete-boolean/src/parameters/mod.rs:30-41. torus_bits is 64 or 32. */
typedef struct camd_params {
  uint32_t lwe_dimension;   /* n */
  uint32_t glwe_dimension;  /* k */
  uint32_t polynomial_size; /* N, power of two, 256..16384 */
  uint32_t pbs_base_log;
  uint32_t pbs_level;
  uint32_t ks_base_log;
  uint32_t ks_level;
  uint32_t torus_bits;
} camd_params;

/* Bootstrap kernel family. GENERIC and BLOCK produce ciphertexts that are bit for bit those of
 * the CPU restatement of the reference algorithm; WAVE (the fastest where it has an instance)
 * performs the same algorithm with its own order of f64 operations: decryptions and the noise
 * bound are the same, the last bits of a ciphertext may differ. AUTO picks WAVE wherever it has an
 * instance (shortint: N = 1024 with k = 2, N = 512 with k = 3, N = 4096 ... 16384 with k = 1 and
 * two levels of base 2^15, N = 2048 ... 8192 with k = 1 and one level of base 2^23; the two boolean parameter
 * sets, 32-bit torus: N = 512, k = 2, two levels; N = 1024, k = 1, three levels) and BLOCK otherwise.
 * (N = 16384 with three levels of base 2^11 has a WAVE instance that AUTO does not select: the 32-bit
 * accumulator would be most of that set's small output noise.)
 * The shortint instances keep 32 bits per accumulator coefficient (N = 1024 and N = 512: in large
 * batches; N >= 2048: always): the ciphertexts they return have zero low dwords, and each
 * blind-rotation step is rounded at 2^-32 of the torus (far below the output noise; part of the
 * tested bound).
 * A WAVE batch of at most one ciphertext per CU (two for N = 512) runs on a variant that gives
 * each ciphertext k+1 wavefronts: less than half the latency, same decryptions; for the
 * shortint N = 1024 and N = 512 sets the words differ from those the same ciphertext gets inside a
 * large batch. The environment variable CAMD_WAVE_LATENCY=0 disables it (a positive value sets
 * the number of ciphertexts per CU up to which it is used). */
#define CAMD_PBS_AUTO 0
#define CAMD_PBS_GENERIC 1
#define CAMD_PBS_WAVE 2
#define CAMD_PBS_BLOCK 3

const char* camd_last_error(void);
/* library version string */
const char* camd_version(void);

/* One context per (process, GPU). Replaces the per-thread engine objects
 * (concrete-shortint/src/engine/mod.rs:88-128; concrete-boolean/src/engine/bootstrapping/cuda.rs:102-121). */
int camd_context_create(int device_id, camd_context** out);
int camd_context_destroy(camd_context* ctx);
int camd_context_device(const camd_context* ctx, int* device_id);
int camd_context_synchronize(camd_context* ctx);

/* Server key creation from the STANDARD-domain bootstrap key and the keyswitch key (host
 * pointers, copied; the caller keeps ownership). The forward negacyclic FFT of every
 * bootstrap-key polynomial runs on the GPU. Replaces
 * concrete-shortint/src/engine/server_side/mod.rs:47-50 (convert_lwe_bootstrap_key) and
 * concrete-boolean/src/engine/bootstrapping/cuda.rs:123-137 (device key from the CPU key). */
int camd_server_key_create(camd_context* ctx, const camd_params* params, const void* bsk_standard,
                           const void* ksk, int pbs_variant, camd_server_key** out);
int camd_server_key_destroy(camd_server_key* key);
int camd_server_key_params(const camd_server_key* key, camd_params* out);
/* which bootstrap kernel family the key uses (CAMD_PBS_GENERIC, CAMD_PBS_WAVE or CAMD_PBS_BLOCK) */
int camd_server_key_variant(const camd_server_key* key, int* variant);

/* Device-resident key material as one blob (Fourier bootstrap key, keyswitch key), so that a
 * key built on one GPU can be replicated to others with one broadcast and no second FFT.
 * export copies the blob into `device_dst` (>= blob size) on `stream`; import builds a key around
 * a copy of `device_src`. Both pointers may also be host memory (the copy direction is taken
 * from the pointer), which is how a key is written to and read back from a file: synchronise
 * `stream` before reading an exported host buffer. The layout of the Fourier part depends on
 * the kernel family, so the variant and params given to import must be those of the exporting
 * key (camd_server_key_variant, camd_server_key_params). It is also private to the library
 * build (the order of the spectrum follows the kernels' thread layout): import only blobs whose
 * layout id, taken with camd_fourier_layout_id when they were exported, equals the importing
 * library's for the same params and variant (a raw blob carries no header; whoever stores one
 * stores the id beside it and compares before importing). */
int camd_server_key_blob_size(const camd_server_key* key, size_t* bytes);
/* Identifies the order in which this build of the library lays out the Fourier bootstrap key of
 * `params` under kernel family `pbs_variant` (AUTO resolves as in camd_server_key_create): a hash
 * of the family, the shape and every compile-time choice that moves a spectrum value inside the
 * blob. No counterpart in the reference, whose Fourier key layout is concrete-fft's own. */
int camd_fourier_layout_id(const camd_params* params, int pbs_variant, uint64_t* id);
int camd_server_key_export_blob(const camd_server_key* key, void* device_dst, void* stream);
int camd_server_key_import_blob(camd_context* ctx, const camd_params* params, int pbs_variant,
                                const void* device_src, size_t bytes, void* stream,
                                camd_server_key** out);

/* Look-up tables ("accumulators"): trivial GLWE ciphertexts of (k+1)*N words built by the
 * caller (concrete-shortint/src/engine/mod.rs:130-178; boolean cpu.rs:49-53). Uploaded once
 * and addressed by id afterwards; ids are dense, starting at 0. */
int camd_lut_upload(camd_server_key* key, const void* glwe_host, uint32_t* lut_id);
int camd_lut_count(const camd_server_key* key, uint32_t* count);

/* ---- host-pointer batch entry points (blocking) ---- */

/* shortint order, keyswitch then bootstrap: replaces
 * ShortintEngine::programmable_bootstrap_keyswitch_assign and keyswitch_bootstrap_assign
 * (concrete-shortint/src/engine/server_side/mod.rs:131-155, :95-118), batched.
 * ct_in/ct_out: [count][kN+1] words, may alias. lut_ids: [count] or NULL (all use lut 0). */
int camd_ks_pbs_batch(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids, void* ct_out,
                      size_t count);
/* boolean order, bootstrap then keyswitch: replaces Bootstrapper::bootstrap_keyswitch
 * (concrete-boolean/src/engine/bootstrapping/mod.rs:41-47, cpu.rs:215-259), batched.
 * ct_in/ct_out: [count][n+1] words, may alias. */
int camd_pbs_ks_batch(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids, void* ct_out,
                      size_t count);
/* bootstrap only: replaces FftEngine::discard_bootstrap_lwe_ciphertext as called at
 * server_side/mod.rs:148-153 and Bootstrapper::bootstrap (bootstrapping/mod.rs:27-31).
 * ct_in: [count][n+1]; ct_out: [count][kN+1]. */
int camd_pbs_batch(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids, void* ct_out,
                   size_t count);
/* keyswitch only: replaces DefaultEngine::discard_keyswitch_lwe_ciphertext as called at
 * server_side/mod.rs:141-145 and Bootstrapper::keyswitch (bootstrapping/mod.rs:33-39).
 * ct_in: [count][kN+1]; ct_out: [count][n+1]. The input dimension kN must be a multiple of 32
 * (it is for every polynomial size the library accepts). */
int camd_ks_batch(camd_server_key* key, const void* ct_in, void* ct_out, size_t count);

/* ---- device-pointer batch entry points (asynchronous on `stream`) ----
 * Same semantics; pointers are device memory on the key's GPU. lut_ids is a device pointer or
 * NULL. Scratch is owned by the key and grown on demand (growth synchronises the device);
 * camd_reserve sizes it ahead of time for `count` ciphertexts. Calls on one key are serialised:
 * a call on another stream than the previous one (the host-pointer entry points use a stream of
 * the context) first makes its stream wait for the previous call's kernels, because all calls
 * on a key share its scratch buffers; use one key per stream for concurrent work.
 * A look-up table id that is out of range cannot be rejected up front here (the ids are in device
 * memory): the kernel bootstraps that ciphertext with table 0 and raises a flag in a host-visible
 * status word, and the NEXT call on the key (any batch entry point, or camd_server_key_check)
 * fails with CAMD_EINVAL and clears the flag. */
int camd_reserve(camd_server_key* key, size_t count);
/* Synchronises `stream` and reports whether any device-pointer call on the key since the last
 * report had a look-up table id out of range (CAMD_EINVAL) - the counterpart, for asynchronous
 * calls, of the id check the host-pointer entry points make before launching. */
int camd_server_key_check(camd_server_key* key, void* stream);
int camd_ks_pbs_batch_device(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids,
                             void* ct_out, size_t count, void* stream);
/* camd_ks_pbs_batch_device for callers that cannot allocate (tight serving loops, harnesses that
 * forbid hipMalloc). It has no counterpart in the reference. It is not meant to be captured into a
 * HIP graph: like every call on a key it reads the key's host status word and records the key's
 * completion event on `stream`. It
 * calls no hipMalloc, hipFree or other allocating routine: the keyswitch output and the kernels'
 * counters live in a fixed 16 MiB + 4 KiB device array that is part of the library's code object
 * (each GPU holds its own copy for the life of the process). Limits: keys of the wave kernel
 * family only, and count * (n + 1) * word bytes <= 16 MiB (2723 ciphertexts for
 * PARAM_MESSAGE_2_CARRY_2); anything beyond fails with CAMD_EINVAL and a message. Results are
 * those of camd_ks_pbs_batch_device. All calls on one GPU share the array, whichever key or
 * context they use: calls that overlap in time on one GPU race on it and must be serialised by
 * the caller (calls on one key are serialised by the library, as above). Kernel timing
 * (camd_timing_*) does not cover this entry point. */
int camd_ks_pbs_batch_device_noalloc(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids,
                                     void* ct_out, size_t count, void* stream);
int camd_pbs_ks_batch_device(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids,
                             void* ct_out, size_t count, void* stream);
int camd_pbs_batch_device(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids,
                          void* ct_out, size_t count, void* stream);
int camd_ks_batch_device(camd_server_key* key, const void* ct_in, void* ct_out, size_t count,
                         void* stream);

/* ---- leveled operations on device ciphertexts (asynchronous on `stream`) ----
 * out[c][j] = alpha * x[c][j] + beta * y[c][j] (wrapping, j < words), then body_add is added to the
 * last word of every row; y may be NULL (then beta is ignored). Rows c < count start
 * `*_stride` words apart, so blocks of a batch of radix integers can be addressed in place; out
 * may alias x or y. Replaces the concrete-core calls the reference makes around a bootstrap:
 * fuse_add_lwe_ciphertext (concrete-shortint/src/engine/server_side/add.rs:22),
 * fuse_opp_lwe_ciphertext + fuse_add_lwe_ciphertext_plaintext (neg.rs:59-66),
 * fuse_mul_lwe_ciphertext_cleartext (scalar_mul.rs:26-27), the shift-and-add ahead of a
 * bivariate bootstrap (mod.rs:182-188) and the boolean gate combinations
 * (concrete-boolean/src/engine/mod.rs:347-353). alpha, beta, body_add are truncated to
 * torus_bits. */
int camd_lwe_axpby_device(camd_context* ctx, uint32_t torus_bits, void* out, size_t out_stride,
                          const void* x, size_t x_stride, uint64_t alpha, const void* y,
                          size_t y_stride, uint64_t beta, uint64_t body_add, size_t words,
                          size_t count, void* stream);

/* ---- a leveled operation and the bootstrap that follows it, as one call ----
 * camd_ks_pbs_batch_device / camd_pbs_ks_batch_device on the rows alpha * x[c] + beta * y[c] with
 * body_add added to the body, as camd_lwe_axpby_device would have written them (same argument
 * meaning; the row length is that of the operation's input ciphertexts). The combination is not
 * written to memory where the first kernel of the operation can form it while it reads its input:
 * every keyswitch kernel (shortint order), and the bootstrap kernels of the block and generic
 * families. For the wave family's bootstrap kernels (boolean order) the library writes it to
 * scratch of the key first, inside the same call; results are the same either way. ct_out may
 * alias x or y.
 * Replaces, as one launch sequence, the shift-and-add followed by the bootstrap of a bivariate
 * operation (concrete-shortint/src/engine/server_side/mod.rs:172-196) and a gate's linear
 * combination followed by bootstrap_keyswitch (concrete-boolean/src/engine/mod.rs:364-405). */
int camd_axpby_ks_pbs_batch_device(camd_server_key* key, const void* x, size_t x_stride,
                                   uint64_t alpha, const void* y, size_t y_stride, uint64_t beta,
                                   uint64_t body_add, const uint32_t* lut_ids, void* ct_out,
                                   size_t count, void* stream);
int camd_axpby_pbs_ks_batch_device(camd_server_key* key, const void* x, size_t x_stride,
                                   uint64_t alpha, const void* y, size_t y_stride, uint64_t beta,
                                   uint64_t body_add, const uint32_t* lut_ids, void* ct_out,
                                   size_t count, void* stream);

/* ---- measurement helpers ----
 * Average device time in milliseconds of the bootstrap kernel (and of the keyswitch kernel)
 * over the launches made since the last reset, taken with HIP events on the stream the
 * kernels were launched on. Timing is off by default (enable != 0 turns it on). */
int camd_timing_enable(camd_server_key* key, int enable);
/* The FP64 fused-multiply-add rate the GPU of `ctx` sustains, in TFLOP/s (independent v_fma_f64
 * chains on every SIMD for about 5 ms): a measured ceiling to set beside the datasheet's. */
int camd_measure_fp64_fma_rate(camd_context* ctx, double* tflops);
int camd_timing_read(camd_server_key* key, double* pbs_ms_avg, uint64_t* pbs_launches,
                     double* ks_ms_avg, uint64_t* ks_launches);

#ifdef __cplusplus
}
#endif
#endif
