// LWE keyswitch, batched: out = (0,...,0,b_in) - sum_i sum_l digit(a_i, l) * KSK[i][l][:]
//
// Replaces DefaultEngine::discard_keyswitch_lwe_ciphertext as the reference calls it at
// concrete-shortint/src/engine/server_side/mod.rs:141-145 and
// concrete-boolean/src/engine/bootstrapping/cpu.rs:250-254 (one ciphertext per call there).
//
// Integer work, bound by the stream of key rows. One workgroup owns CT_TILE ciphertexts and
// every output column, so that each key word fetched (from L2 / Infinity Cache) is used
// CT_TILE times; mask elements are decomposed once per chunk into LDS and broadcast.
#include "kernels.hpp"

namespace camd {

namespace {

constexpr int KS_THREADS = 256;
constexpr int KS_CT_TILE = 8;     // ciphertexts per workgroup
constexpr int KS_COLS = 4;        // output columns per thread (covers n+1 <= 1024)
constexpr int KS_CHUNK = 32;      // mask elements decomposed per LDS refill
constexpr int KS_MAX_LEVEL = 8;

template <typename T>
__global__ __launch_bounds__(KS_THREADS) void keyswitch_kernel(
    const T* __restrict__ ct_in, size_t in_stride, T* __restrict__ ct_out, size_t out_stride,
    const T* __restrict__ ksk, uint32_t in_dim, uint32_t out_dim, uint32_t base_log, uint32_t level,
    size_t count) {
  using S = typename Torus<T>::Signed;
  __shared__ int32_t digits[KS_CHUNK][KS_MAX_LEVEL][KS_CT_TILE];

  const uint32_t tid = threadIdx.x;
  const size_t ct0 = (size_t)blockIdx.x * KS_CT_TILE;
  const uint32_t ncols = out_dim + 1;
  const uint32_t col_base = blockIdx.y * (KS_THREADS * KS_COLS);

  T acc[KS_CT_TILE][KS_COLS];
#pragma unroll
  for (int c = 0; c < KS_CT_TILE; c++)
#pragma unroll
    for (int q = 0; q < KS_COLS; q++) acc[c][q] = 0;

  for (uint32_t i0 = 0; i0 < in_dim; i0 += KS_CHUNK) {
    // decompose KS_CHUNK mask elements of each ciphertext of the tile
    for (uint32_t e = tid; e < KS_CHUNK * KS_CT_TILE; e += KS_THREADS) {
      const uint32_t c = e % KS_CT_TILE, ii = e / KS_CT_TILE;
      const uint32_t i = i0 + ii;
      T a = 0;
      if (i < in_dim && ct0 + c < count) a = ct_in[(ct0 + c) * in_stride + i];
      T state = decomp_init_state<T>(a, base_log, level);
      for (uint32_t l = level; l >= 1; l--) digits[ii][l - 1][c] = (int32_t)decomp_next<T>(state, base_log);
    }
    __syncthreads();
    const uint32_t iend = min((uint32_t)KS_CHUNK, in_dim - i0);
    for (uint32_t ii = 0; ii < iend; ii++) {
      for (uint32_t l = 0; l < level; l++) {
        const T* row = ksk + ((size_t)(i0 + ii) * level + l) * ncols;
        T kv[KS_COLS];
#pragma unroll
        for (int q = 0; q < KS_COLS; q++) {
          const uint32_t col = col_base + q * KS_THREADS + tid;
          kv[q] = (col < ncols) ? row[col] : T(0);
        }
#pragma unroll
        for (int c = 0; c < KS_CT_TILE; c++) {
          const T d = (T)(S)digits[ii][l][c];
#pragma unroll
          for (int q = 0; q < KS_COLS; q++) acc[c][q] -= d * kv[q];
        }
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int c = 0; c < KS_CT_TILE; c++) {
    if (ct0 + c >= count) break;
#pragma unroll
    for (int q = 0; q < KS_COLS; q++) {
      const uint32_t col = col_base + q * KS_THREADS + tid;
      if (col < ncols) {
        T v = acc[c][q];
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
  const uint32_t ncols = out_dim + 1;
  const uint32_t per_block = KS_THREADS * KS_COLS;
  dim3 grid((unsigned)((count + KS_CT_TILE - 1) / KS_CT_TILE), (ncols + per_block - 1) / per_block);
  hipLaunchKernelGGL(keyswitch_kernel<T>, grid, dim3(KS_THREADS), 0, stream, ct_in, in_stride, ct_out,
                     out_stride, ksk, in_dim, out_dim, base_log, level, count);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

template int launch_keyswitch<uint64_t>(const uint64_t*, size_t, uint64_t*, size_t, const uint64_t*, uint32_t,
                                        uint32_t, uint32_t, uint32_t, size_t, hipStream_t);
template int launch_keyswitch<uint32_t>(const uint32_t*, size_t, uint32_t*, size_t, const uint32_t*, uint32_t,
                                        uint32_t, uint32_t, uint32_t, size_t, hipStream_t);

}  // namespace camd
