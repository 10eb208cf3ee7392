// LWE keyswitch, batched: out = (0,...,0,b_in) - sum_i sum_l digit(a_i, l) * KSK[i][l][:]
//
// Replaces DefaultEngine::discard_keyswitch_lwe_ciphertext as the reference calls it at
// concrete-shortint/src/engine/server_side/mod.rs:141-145 and
// concrete-boolean/src/engine/bootstrapping/cpu.rs:250-254 (one ciphertext per call there).
//
// Exact integer arithmetic on the f64 pipe. gfx950 has no 64-bit integer multiply: a wrapping
// u64 * small-digit multiply-add costs about nine integer instructions. Instead every key word
// is split into 32-bit limbs, and limb * digit products are accumulated with v_fma_f64:
// |digit| <= B/2 and limb < 2^32, so a sum of up to 2^(52 - 32 - log2(B/2)) terms is an exact
// integer in a double. The partial sums are folded into the wrapping torus accumulator before
// that bound is reached (for every parameter set of the reference: once, at the end), so the
// result is bit for bit the integer one. One multiply-add is then one FMA per limb.
//
// Tiling: one workgroup owns CT_TILE ciphertexts and 256 output columns; each key word fetched
// (from L2 / Infinity Cache) is used CT_TILE times. Mask elements are decomposed once per chunk
// into LDS, already converted to f64, and read back as broadcasts.
#include "kernels.hpp"

namespace camd {

namespace {

constexpr int KS_THREADS = 256;
constexpr int KS_CT_TILE = 16;    // ciphertexts per workgroup
constexpr int KS_COLS = 1;        // output columns per thread
constexpr int KS_CHUNK = 32;      // mask elements decomposed per LDS refill
constexpr int KS_MAX_LEVEL = 8;

// exact integer-valued double (|v| < 2^52) -> two's-complement 64-bit word
__device__ __forceinline__ uint64_t exact_to_u64(double v) {
  const double hi = floor(v * 0x1.0p-32);
  const double lo = fma(hi, -0x1.0p32, v);  // [0, 2^32)
  return ((uint64_t)(uint32_t)(int32_t)hi << 32) | (uint64_t)(uint32_t)lo;
}

template <typename T>
__global__ __launch_bounds__(KS_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void keyswitch_kernel(
    const T* __restrict__ ct_in, size_t in_stride, T* __restrict__ ct_out, size_t out_stride,
    const T* __restrict__ ksk, uint32_t in_dim, uint32_t out_dim, uint32_t base_log, uint32_t level,
    size_t count, uint32_t flush_chunks) {
  constexpr int LIMBS = sizeof(T) / 4;
  extern __shared__ double digits[];  // [KS_CHUNK][level][KS_CT_TILE]

  const uint32_t tid = threadIdx.x;
  const size_t ct0 = (size_t)blockIdx.x * KS_CT_TILE;
  const uint32_t ncols = out_dim + 1;
  const uint32_t col_base = blockIdx.y * (KS_THREADS * KS_COLS);

  T total[KS_CT_TILE][KS_COLS];            // wrapping accumulator
  double part[KS_CT_TILE][KS_COLS][LIMBS]; // exact partial sums, one per limb
#pragma unroll
  for (int c = 0; c < KS_CT_TILE; c++)
#pragma unroll
    for (int q = 0; q < KS_COLS; q++) {
      total[c][q] = 0;
#pragma unroll
      for (int m = 0; m < LIMBS; m++) part[c][q][m] = 0.0;
    }

  auto flush = [&]() {
#pragma unroll
    for (int c = 0; c < KS_CT_TILE; c++)
#pragma unroll
      for (int q = 0; q < KS_COLS; q++) {
        uint64_t s = exact_to_u64(part[c][q][0]);
        if (LIMBS == 2) s += exact_to_u64(part[c][q][LIMBS - 1]) << 32;
        total[c][q] -= (T)s;
#pragma unroll
        for (int m = 0; m < LIMBS; m++) part[c][q][m] = 0.0;
      }
  };

  uint32_t chunk_no = 0;
  for (uint32_t i0 = 0; i0 < in_dim; i0 += KS_CHUNK, chunk_no++) {
    // decompose KS_CHUNK mask elements of each ciphertext of the tile
    for (uint32_t e = tid; e < KS_CHUNK * KS_CT_TILE; e += KS_THREADS) {
      const uint32_t c = e % KS_CT_TILE, ii = e / KS_CT_TILE;
      const uint32_t i = i0 + ii;
      T a = 0;
      if (i < in_dim && ct0 + c < count) a = ct_in[(ct0 + c) * in_stride + i];
      T state = decomp_init_state<T>(a, base_log, level);
      for (uint32_t l = level; l >= 1; l--)
        digits[(ii * level + (l - 1)) * KS_CT_TILE + c] = (double)(int32_t)decomp_next<T>(state, base_log);
    }
    __syncthreads();
    const uint32_t iend = min((uint32_t)KS_CHUNK, in_dim - i0);
    const uint32_t rows = iend * level;               // key rows of this chunk, (i, l) flattened
    const T* row0 = ksk + (size_t)i0 * level * ncols;
    // key words are fetched one row ahead of the products that use them
    T kv[KS_COLS];
#pragma unroll
    for (int q = 0; q < KS_COLS; q++) {
      const uint32_t col = col_base + q * KS_THREADS + tid;
      kv[q] = (col < ncols) ? row0[col] : T(0);
    }
    // a wave whose columns all lie past the end (the last column block is mostly empty: n+1 is
    // rarely a multiple of the block) skips the products; it still takes part in the barriers
    const bool wave_active = col_base + (tid & ~63u) < ncols;
    for (uint32_t rr = 0; wave_active && rr < rows; rr++) {
      double kf[KS_COLS][LIMBS];
#pragma unroll
      for (int q = 0; q < KS_COLS; q++) {
        kf[q][0] = (double)(uint32_t)kv[q];
        if (LIMBS == 2) kf[q][LIMBS - 1] = (double)(uint32_t)((uint64_t)kv[q] >> 32);
      }
      if (rr + 1 < rows) {
        const T* nrow = row0 + (size_t)(rr + 1) * ncols;
#pragma unroll
        for (int q = 0; q < KS_COLS; q++) {
          const uint32_t col = col_base + q * KS_THREADS + tid;
          kv[q] = (col < ncols) ? nrow[col] : T(0);
        }
      }
      const double2* dg = (const double2*)(digits + (size_t)rr * KS_CT_TILE);
#pragma unroll
      for (int c = 0; c < KS_CT_TILE; c += 2) {
        const double2 d = dg[c >> 1];  // two ciphertexts' digits per LDS read
#pragma unroll
        for (int q = 0; q < KS_COLS; q++)
#pragma unroll
          for (int m = 0; m < LIMBS; m++) {
            part[c][q][m] = fma(d.x, kf[q][m], part[c][q][m]);
            part[c + 1][q][m] = fma(d.y, kf[q][m], part[c + 1][q][m]);
          }
      }
    }
    __syncthreads();
    if ((chunk_no + 1) % flush_chunks == 0) flush();
  }
  flush();

#pragma unroll
  for (int c = 0; c < KS_CT_TILE; c++) {
    if (ct0 + c >= count) break;
#pragma unroll
    for (int q = 0; q < KS_COLS; q++) {
      const uint32_t col = col_base + q * KS_THREADS + tid;
      if (col < ncols) {
        T v = total[c][q];
        if (col == out_dim) v += ct_in[(ct0 + c) * in_stride + in_dim];
        ct_out[(ct0 + c) * out_stride + col] = v;
      }
    }
  }
}

}  // namespace

template <typename T>
int launch_keyswitch(const T* ct_in, size_t in_stride, T* ct_out, size_t out_stride, const T* ksk,
                     uint32_t in_dim, uint32_t out_dim, uint32_t base_log, uint32_t level, size_t count,
                     hipStream_t stream) {
  if (count == 0) return CAMD_OK;
  if (level > KS_MAX_LEVEL) return fail(CAMD_EINVAL, "keyswitch: level count above 8 is not supported");
  // number of products an f64 partial sum can take exactly: 2^(52 - 32 - (base_log - 1))
  const int budget_log = 52 - 32 - ((int)base_log - 1);
  if (budget_log < 0) return fail(CAMD_EINVAL, "keyswitch: base_log above 21 is not supported");
  const uint64_t budget = 1ull << budget_log;
  const uint64_t per_chunk = (uint64_t)KS_CHUNK * level;
  uint32_t flush_chunks = (uint32_t)(budget / per_chunk);
  if (flush_chunks == 0) return fail(CAMD_EINVAL, "keyswitch: decomposition too wide for exact f64 accumulation");
  const uint32_t ncols = out_dim + 1;
  const uint32_t per_block = KS_THREADS * KS_COLS;
  dim3 grid((unsigned)((count + KS_CT_TILE - 1) / KS_CT_TILE), (ncols + per_block - 1) / per_block);
  const size_t lds = (size_t)KS_CHUNK * level * KS_CT_TILE * sizeof(double);
  hipLaunchKernelGGL(keyswitch_kernel<T>, grid, dim3(KS_THREADS), lds, stream, ct_in, in_stride, ct_out,
                     out_stride, ksk, in_dim, out_dim, base_log, level, count, flush_chunks);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

template int launch_keyswitch<uint64_t>(const uint64_t*, size_t, uint64_t*, size_t, const uint64_t*, uint32_t,
                                        uint32_t, uint32_t, uint32_t, size_t, hipStream_t);
template int launch_keyswitch<uint32_t>(const uint32_t*, size_t, uint32_t*, size_t, const uint32_t*, uint32_t,
                                        uint32_t, uint32_t, uint32_t, size_t, hipStream_t);

}  // namespace camd
