// LWE keyswitch of a batch as the dense integer contraction it is, on the matrix cores.
//
//   out[b][j] = (j == n ? body_b : 0) - sum_{kappa < kN*l} digit[b][kappa] * KSK[kappa][j]   (mod 2^BITS)
//
// with kappa = (mask element i, level) in the key's own row order. Same operation as keyswitch.hip
// (DefaultEngine::discard_keyswitch_lwe_ciphertext, called by the reference at
// concrete-shortint/src/engine/server_side/mod.rs:141-145 and concrete-boolean/src/engine/
// bootstrapping/cpu.rs:250-254), bit for bit: everything here is integer arithmetic.
//
// [count x K] signed digits (|d| <= B/2 <= 32) times [K x (n+1)] torus words is a GEMM whose left
// operand fits 8 bits. The key words are split once, at key creation, into BALANCED byte limbs
//   w = sum_t s_t 2^(8t)  (mod 2^BITS),   s_t in [-128, 127]
// (the carry of a byte >= 128 goes into the next limb; the carry out of the top limb drops out
// modulo 2^BITS). Then   sum_kappa d * w = sum_t 2^(8t) * (sum_kappa d * s_t),   and each inner sum
// is an exact i32 accumulation of v_mfma_i32_32x32x32_i8 as long as K * (B/2) * 128 < 2^31, which
// holds for every parameter set of the reference (largest: 98304 * 4 * 128 = 2^25.6). The limb sums
// are recombined with shifts in the epilogue, wrapping.
//
// Data layout
//   digits  [count][K] int8      row-major, written by ks_digits_kernel from the input ciphertexts
//   limbs   [NL][Npad][K] int8   limb-planar and TRANSPOSED (kappa contiguous), Npad = columns
//                                rounded up to 32, padding rows zero; written once per key
// so both MFMA operands are "row with kappa contiguous" and take the same path into LDS.
//
// Tiling: a workgroup of four waves owns 256 ciphertexts x 32 columns x all limbs; wave w owns
// rows 64w..64w+63 (two 32x32 tiles) of every limb: 2 x NL accumulators of 16 registers. K is
// walked in stages of 128 bytes per row (one cache line), double-buffered through LDS with rows
// padded to 144 bytes (conflict-free for the 16-lane groups of ds_read_b128). When the grid
// would not fill the chip (small batches, or the N = 16384 sets where K is 98304) K is split over
// blockIdx.z as well and the partial sums are added into the preset output with 64/32-bit atomics:
// wrapping sums commute, so the result is still exact.
#include <algorithm>

#include "kernels.hpp"

namespace camd {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int KSM_THREADS = 256;
constexpr int KSM_M = 256;       // ciphertexts per workgroup
constexpr int KSM_N = 32;        // output columns per workgroup
constexpr int KSM_KT = 128;      // bytes of kappa per row and stage
constexpr int KSM_ROW = 144;     // padded LDS row
constexpr int KSM_CHUNKS = KSM_KT / 16;

// ---- key preparation: balanced byte limbs, transposed ----
template <typename T>
__global__ __launch_bounds__(256) void ks_limbs_kernel(const T* __restrict__ ksk, int8_t* __restrict__ limbs, uint32_t K,
                                                       uint32_t cols, uint32_t npad) {
  constexpr int NL = sizeof(T);
  // a 32 x 32 tile of (kappa, column) per workgroup, transposed through LDS so that both the reads
  // (columns contiguous) and the writes (kappa contiguous) are coalesced
  __shared__ T tile[32][33];
  const uint32_t k0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (uint32_t r = ty; r < 32; r += 8) {
    const uint32_t kappa = k0 + r, j = j0 + tx;
    tile[r][tx] = (kappa < K && j < cols) ? ksk[(size_t)kappa * cols + j] : T(0);
  }
  __syncthreads();
  for (uint32_t r = ty; r < 32; r += 8) {
    const uint32_t j = j0 + r, kappa = k0 + tx;
    if (j >= npad || kappa >= K) continue;
    T w = tile[tx][r];
#pragma unroll
    for (int t = 0; t < NL; t++) {
      const uint32_t byte = (uint32_t)(w & T(0xFF));
      w >>= 8;
      if (byte >= 128u) w += T(1);   // balanced digit byte - 256, carry into the next limb
      limbs[((size_t)t * npad + j) * K + kappa] = (int8_t)byte;
    }
  }
}

// ---- digits of the mask elements of a batch ----
// thread = one mask element of one ciphertext: `level` bytes at kappa = i*level + (l-1), the row
// order of the key (level 1 first); the decomposer yields the finest level first.
template <typename T>
__global__ __launch_bounds__(256) void ks_digits_kernel(const LweInput<T> in,
                                                        int8_t* __restrict__ digits, uint32_t in_dim, uint32_t base_log,
                                                        uint32_t level, size_t count) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const size_t b = blockIdx.y;
  if (i >= in_dim || b >= count) return;
  T state = decomp_init_state<T>(in.at(b, i), base_log, level);
  int8_t* dst = digits + (b * in_dim + i) * level;
  for (uint32_t l = level; l >= 1; l--) dst[l - 1] = (int8_t)decomp_next<T>(state, base_log);
}

// out = (0, ..., 0, b_in): the start value when the partial sums arrive by atomics
template <typename T>
__global__ __launch_bounds__(256) void ksm_preset_kernel(const LweInput<T> in,
                                                         T* __restrict__ ct_out, size_t out_stride, uint32_t in_dim,
                                                         uint32_t out_dim, size_t count) {
  const uint32_t col = blockIdx.x * 256 + threadIdx.x;
  const size_t c = blockIdx.y;
  if (col > out_dim || c >= count) return;
  ct_out[c * out_stride + col] = (col == out_dim) ? in.at(c, in_dim) : T(0);
}

// ---- the contraction ----
template <typename T>
__global__ __launch_bounds__(KSM_THREADS, 1) void ks_mfma_kernel(const int8_t* __restrict__ digits,
                                                                 const int8_t* __restrict__ limbs,
                                                                 const LweInput<T> in,
                                                                 T* __restrict__ ct_out, size_t out_stride,
                                                                 uint32_t in_dim, uint32_t out_dim, uint32_t K, uint32_t npad,
                                                                 size_t count, uint32_t stages_per_split, int atomic) {
  constexpr int NL = sizeof(T);
  constexpr int BROWS = NL * KSM_N;                       // rows of the key tile: (limb, column)
  constexpr int A_BYTES = KSM_M * KSM_ROW, B_BYTES = BROWS * KSM_ROW;
  constexpr int A_LOADS = KSM_M * KSM_CHUNKS / KSM_THREADS;    // 16-byte chunks per thread and stage
  constexpr int B_LOADS = BROWS * KSM_CHUNKS / KSM_THREADS;
  extern __shared__ __align__(16) char smem[];             // [2][A tile | B tile]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t m0 = (size_t)blockIdx.x * KSM_M;
  const uint32_t n0 = blockIdx.y * KSM_N;
  const uint32_t total_stages = K / KSM_KT;
  const uint32_t s_begin = blockIdx.z * stages_per_split;
  const uint32_t s_end = min(s_begin + stages_per_split, total_stages);

  v4i stage_a[A_LOADS], stage_b[B_LOADS];
  auto load_global = [&](uint32_t s) {
    const size_t k0 = (size_t)s * KSM_KT;
#pragma unroll
    for (int q = 0; q < A_LOADS; q++) {
      const int id = tid + q * KSM_THREADS, row = id / KSM_CHUNKS, c = id % KSM_CHUNKS;
      const size_t m = m0 + row;
      stage_a[q] = (m < count) ? *(const v4i*)(digits + m * K + k0 + 16 * c) : v4i{0, 0, 0, 0};
    }
#pragma unroll
    for (int q = 0; q < B_LOADS; q++) {
      const int id = tid + q * KSM_THREADS, row = id / KSM_CHUNKS, c = id % KSM_CHUNKS;
      const int t = row / KSM_N, col = row % KSM_N;
      stage_b[q] = *(const v4i*)(limbs + ((size_t)t * npad + n0 + col) * K + k0 + 16 * c);
    }
  };
  auto store_lds = [&](int buf) {
    char* a = smem + buf * (A_BYTES + B_BYTES);
    char* b = a + A_BYTES;
#pragma unroll
    for (int q = 0; q < A_LOADS; q++) {
      const int id = tid + q * KSM_THREADS, row = id / KSM_CHUNKS, c = id % KSM_CHUNKS;
      *(v4i*)(a + row * KSM_ROW + 16 * c) = stage_a[q];
    }
#pragma unroll
    for (int q = 0; q < B_LOADS; q++) {
      const int id = tid + q * KSM_THREADS, row = id / KSM_CHUNKS, c = id % KSM_CHUNKS;
      *(v4i*)(b + row * KSM_ROW + 16 * c) = stage_b[q];
    }
  };

  v16i acc[2][NL];
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int t = 0; t < NL; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[mi][t][r] = 0;

  if (s_begin < s_end) {
    load_global(s_begin);
    store_lds(0);
  }
  __syncthreads();
  const int frow = lane & 31, fhalf = lane >> 5;   // operand fragment: row (or column) and which 16 of the 32 kappa
  for (uint32_t s = s_begin; s < s_end; s++) {
    const int buf = (int)((s - s_begin) & 1u);
    if (s + 1 < s_end) load_global(s + 1);   // in flight behind the products of this stage
    const char* a = smem + buf * (A_BYTES + B_BYTES);
    const char* b = a + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < KSM_KT / 32; ks++) {
      const int off = (2 * ks + fhalf) * 16;
      v4i fa[2], fb[NL];
#pragma unroll
      for (int mi = 0; mi < 2; mi++) fa[mi] = *(const v4i*)(a + (wave * 64 + mi * 32 + frow) * KSM_ROW + off);
#pragma unroll
      for (int t = 0; t < NL; t++) fb[t] = *(const v4i*)(b + (t * KSM_N + frow) * KSM_ROW + off);
#pragma unroll
      for (int t = 0; t < NL; t++)
#pragma unroll
        for (int mi = 0; mi < 2; mi++) acc[mi][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[mi], fb[t], acc[mi][t], 0, 0, 0);
    }
    if (s + 1 < s_end) store_lds(buf ^ 1);
    __syncthreads();
  }

  // epilogue: recombine the limb sums (wrapping) and subtract from (0, ..., 0, body).
  // accumulator register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31 of its tile
  const uint32_t col = n0 + (uint32_t)(lane & 31);
  if (col > out_dim) return;
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const size_t m = m0 + (size_t)(wave * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
      if (m >= count) continue;
      T sum = 0;
#pragma unroll
      for (int t = 0; t < NL; t++) sum += (T)(typename Torus<T>::Signed)acc[mi][t][r] << (8 * t);
      T* dst = ct_out + m * out_stride + col;
      if (atomic) {
        if constexpr (sizeof(T) == 8) atomicAdd((unsigned long long*)dst, (unsigned long long)(T(0) - sum));
        else atomicAdd((unsigned int*)dst, (unsigned int)(T(0) - sum));
      } else {
        const T base = (col == out_dim) ? in.at(m, in_dim) : T(0);
        *dst = base - sum;
      }
    }
}

}  // namespace

bool keyswitch_mfma_supported(uint32_t in_dim, uint32_t base_log, uint32_t level) {
  const uint64_t K = (uint64_t)in_dim * level;
  if (K % KSM_KT != 0 || base_log == 0 || base_log > 6) return false;
  // exact i32 accumulation: K * (B/2) * 128 < 2^31
  return K * (1ull << (base_log - 1)) * 128ull < (1ull << 31);
}

size_t keyswitch_mfma_limb_bytes(uint32_t in_dim, uint32_t out_dim, uint32_t level, int torus_bytes) {
  const size_t npad = ((size_t)out_dim + 1 + KSM_N - 1) / KSM_N * KSM_N;
  return (size_t)torus_bytes * npad * in_dim * level;
}

size_t keyswitch_mfma_digit_bytes(uint32_t in_dim, uint32_t level, size_t count) { return count * (size_t)in_dim * level; }

template <typename T>
int keyswitch_mfma_prepare(const T* ksk, int8_t* limbs, uint32_t in_dim, uint32_t out_dim, uint32_t level, hipStream_t stream) {
  const uint32_t K = in_dim * level, cols = out_dim + 1;
  const uint32_t npad = (cols + KSM_N - 1) / KSM_N * KSM_N;
  hipLaunchKernelGGL(ks_limbs_kernel<T>, dim3((K + 31) / 32, npad / 32), dim3(256), 0, stream, ksk, limbs, K, cols, npad);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

template <typename T>
int launch_keyswitch_mfma(const LweInput<T>& in, T* ct_out, size_t out_stride, const int8_t* limbs,
                          int8_t* digits, uint32_t in_dim, uint32_t out_dim, uint32_t base_log, uint32_t level, size_t count,
                          hipStream_t stream) {
  if (count == 0) return CAMD_OK;
  if (in.body != in_dim) return fail(CAMD_EINVAL, "keyswitch: the input's body index is not the input dimension");
  if (!keyswitch_mfma_supported(in_dim, base_log, level)) return fail(CAMD_EINVAL, "keyswitch: no matrix-core form for this decomposition");
  if (count > 65535u * 16u) return fail(CAMD_EINVAL, "keyswitch: batch too large for one launch");
  const uint32_t K = in_dim * level, cols = out_dim + 1;
  const uint32_t npad = (cols + KSM_N - 1) / KSM_N * KSM_N;
  // digits of the batch: count x K bytes
  {
    // blockIdx.y is limited to 65535: batches beyond that go in slabs
    for (size_t first = 0; first < count; first += 65535) {
      const size_t nb = std::min<size_t>(65535, count - first);
      hipLaunchKernelGGL(ks_digits_kernel<T>, dim3((in_dim + 255) / 256, (unsigned)nb), dim3(256), 0, stream,
                         in.rows_from(first), digits + first * (size_t)K, in_dim, base_log, level, nb);
    }
    CAMD_HIP(hipGetLastError());
  }
  const unsigned mt = (unsigned)((count + KSM_M - 1) / KSM_M), nt = npad / KSM_N;
  const uint32_t stages = K / KSM_KT;
  // split K until the grid covers the chip about twice, in ranges of at least 4 stages
  int dev = 0, cus = 0;
  CAMD_HIP(hipGetDevice(&dev));
  CAMD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const uint64_t blocks = (uint64_t)mt * nt;
  uint32_t splits = 1;
  const uint64_t want = 2ull * (uint64_t)(cus > 0 ? cus : 256);
  if (blocks < want) splits = (uint32_t)std::min<uint64_t>((want + blocks - 1) / blocks, std::max<uint32_t>(1, stages / 4));
  const uint32_t per_split = (stages + splits - 1) / splits;
  splits = (stages + per_split - 1) / per_split;
  if (splits > 1) {
    for (size_t first = 0; first < count; first += 65535) {
      const size_t nb = std::min<size_t>(65535, count - first);
      hipLaunchKernelGGL(ksm_preset_kernel<T>, dim3((cols + 255) / 256, (unsigned)nb), dim3(256), 0, stream,
                         in.rows_from(first), ct_out + first * out_stride, out_stride, in_dim, out_dim, nb);
    }
    CAMD_HIP(hipGetLastError());
  }
  constexpr size_t lds = 2 * ((size_t)KSM_M * KSM_ROW + sizeof(T) * KSM_N * KSM_ROW);
  static_assert(lds <= 160 * 1024, "LDS budget exceeded");
  auto kern = ks_mfma_kernel<T>;
  CAMD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(mt, nt, splits), dim3(KSM_THREADS), lds, stream, (const int8_t*)digits, limbs, in,
                     ct_out, out_stride, in_dim, out_dim, K, npad, count, per_split, splits > 1 ? 1 : 0);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

template int keyswitch_mfma_prepare<uint64_t>(const uint64_t*, int8_t*, uint32_t, uint32_t, uint32_t, hipStream_t);
template int keyswitch_mfma_prepare<uint32_t>(const uint32_t*, int8_t*, uint32_t, uint32_t, uint32_t, hipStream_t);
template int launch_keyswitch_mfma<uint64_t>(const LweInput<uint64_t>&, uint64_t*, size_t, const int8_t*, int8_t*, uint32_t,
                                             uint32_t, uint32_t, uint32_t, size_t, hipStream_t);
template int launch_keyswitch_mfma<uint32_t>(const LweInput<uint32_t>&, uint32_t*, size_t, const int8_t*, int8_t*, uint32_t,
                                             uint32_t, uint32_t, uint32_t, size_t, hipStream_t);

}  // namespace camd
