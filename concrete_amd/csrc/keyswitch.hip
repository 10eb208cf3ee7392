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
// into LDS, already converted to f64.
//
// The digit operand of a product is the same for every lane (it belongs to the ciphertext, the
// lanes are columns). Reading it from LDS as a broadcast costs a 16-byte LDS read per four
// FMAs, and with sixteen waves per CU the LDS pipe, not the vector ALU, sets the pace (measured:
// twice the FMA time). Instead one 8-byte read per key row puts the sixteen digits of the tile
// into one register pair, lane l holding the digit of ciphertext l mod 16, and each FMA takes its
// digit from there through the DPP row broadcast of the f64 pipe (row_newbcast:c hands lane c of
// every 16-lane row to all lanes of that row): no further LDS traffic, no extra instruction.
#include <algorithm>

#include "kernels.hpp"

namespace camd {

namespace {

constexpr int KS_THREADS = 256;
constexpr int KS_CT_TILE = 16;    // ciphertexts per workgroup
constexpr int KS_COLS = 1;        // output columns per thread
constexpr int KS_CHUNK = 32;      // mask elements decomposed per LDS refill
constexpr int KS_MAX_LEVEL = 8;

static_assert(KS_CT_TILE == 16, "the DPP row broadcast addresses one of 16 lanes: one tile = one row");

// acc += digits[lane C of each row] * k. The DPP source must not have been written by one of the
// two preceding vector instructions (a hazard the compiler does not track through inline
// assembly). It never is here: the register pair comes straight from an LDS read, which is
// awaited with s_waitcnt, not from the vector ALU; the CPU tests check this on the generated code.
template <int C>
__device__ __forceinline__ void fma_digit(double& acc, double digits16, double k) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "+v"(acc)
      : "v"(digits16), "v"(k), "n"(C));
}
template <int C, int COLS, int LIMBS>
__device__ __forceinline__ void fma_tile(double (&part)[KS_CT_TILE][COLS][LIMBS], double digits16,
                                         const double (&kf)[COLS][LIMBS]) {
  if constexpr (C < KS_CT_TILE) {
#pragma unroll
    for (int q = 0; q < COLS; q++)
#pragma unroll
      for (int m = 0; m < LIMBS; m++) fma_digit<C>(part[C][q][m], digits16, kf[q][m]);
    fma_tile<C + 1, COLS, LIMBS>(part, digits16, kf);
  }
}

// exact integer-valued double (|v| < 2^52) -> two's-complement 64-bit word
__device__ __forceinline__ uint64_t exact_to_u64(double v) {
  const double hi = floor(v * 0x1.0p-32);
  const double lo = fma(hi, -0x1.0p32, v);  // [0, 2^32)
  return ((uint64_t)(uint32_t)(int32_t)hi << 32) | (uint64_t)(uint32_t)lo;
}

template <typename T>
__global__ __launch_bounds__(KS_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void keyswitch_kernel(
    const LweInput<T> in, T* __restrict__ ct_out, size_t out_stride,
    const T* __restrict__ ksk, uint32_t in_dim, uint32_t out_dim, uint32_t base_log, uint32_t level,
    size_t count, uint32_t flush_chunks, uint32_t chunks_per_split) {
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

  // Key words travel two rows ahead of the products that use them, in two alternating register
  // sets; the row index runs across the chunks (key rows are contiguous), so the pipeline never
  // restarts. Columns past the end are clamped to the last one: those lanes compute a duplicate
  // that is dropped at the store, and no load or product sits under a branch.
  const uint32_t total_rows = in_dim * level;      // even: in_dim is a multiple of KS_CHUNK
  const T* kcol[KS_COLS];
#pragma unroll
  for (int q = 0; q < KS_COLS; q++) kcol[q] = ksk + min(col_base + q * KS_THREADS + tid, ncols - 1);
  auto fetch = [&](T (&kv)[KS_COLS], uint32_t row) {
    const size_t off = (size_t)min(row, total_rows - 1) * ncols;
#pragma unroll
    for (int q = 0; q < KS_COLS; q++) kv[q] = kcol[q][off];
  };
  // a wave whose columns all lie past the end (the last column block is mostly empty: n+1 is
  // rarely a multiple of the block) skips the products; it still takes part in the barriers
  const bool wave_active = col_base + (tid & ~63u) < ncols;
  // blockIdx.z selects a contiguous range of chunks (a small batch is split over the key rows
  // so that it still fills the chip); the ranges' sums meet in the output through atomic adds
  const uint32_t chunk_first = blockIdx.z * chunks_per_split;
  const uint32_t chunk_end = min(chunk_first + chunks_per_split, in_dim / KS_CHUNK);
  uint32_t chunk_no = 0, row_base = chunk_first * KS_CHUNK * level;
  T kva[KS_COLS], kvb[KS_COLS];
  if (wave_active) {
    fetch(kva, row_base);
    fetch(kvb, row_base + 1);
  }
  auto products = [&](T (&kv)[KS_COLS], uint32_t rr, uint32_t next_row) {
    double kf[KS_COLS][LIMBS];
#pragma unroll
    for (int q = 0; q < KS_COLS; q++) {
      kf[q][0] = (double)(uint32_t)kv[q];
      if (LIMBS == 2) {
        // opaque to the optimiser: it would otherwise widen this back into a 64-bit conversion
        // (two conversions, a scale and an add) of a value it knows to be below 2^32
        uint32_t hi = (uint32_t)((uint64_t)kv[q] >> 32);
        asm("" : "+v"(hi));
        kf[q][LIMBS - 1] = (double)hi;
      }
    }
    fetch(kv, next_row);
    // the sixteen digits of this key row, lane l holding ciphertext l mod 16
    const double digits16 = digits[(size_t)rr * KS_CT_TILE + (tid & (KS_CT_TILE - 1))];
    fma_tile<0, KS_COLS, LIMBS>(part, digits16, kf);
  };

  for (uint32_t chunk = chunk_first; chunk < chunk_end; chunk++, chunk_no++) {
    const uint32_t i0 = chunk * KS_CHUNK;
    // decompose KS_CHUNK mask elements of each ciphertext of the tile
    for (uint32_t e = tid; e < KS_CHUNK * KS_CT_TILE; e += KS_THREADS) {
      const uint32_t c = e % KS_CT_TILE, ii = e / KS_CT_TILE;
      const uint32_t i = i0 + ii;
      T a = 0;
      if (ct0 + c < count) a = in.at(ct0 + c, i);
      T state = decomp_init_state<T>(a, base_log, level);
      for (uint32_t l = level; l >= 1; l--)
        digits[(ii * level + (l - 1)) * KS_CT_TILE + c] = (double)(int32_t)decomp_next<T>(state, base_log);
    }
    __syncthreads();
    const uint32_t rows = KS_CHUNK * level;           // key rows of this chunk, (i, l) flattened
    if (wave_active) {
      for (uint32_t rr = 0; rr < rows; rr += 2) {
        products(kva, rr, row_base + rr + 2);
        products(kvb, rr + 1, row_base + rr + 3);
      }
    }
    row_base += rows;
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
        T* dst = &ct_out[(ct0 + c) * out_stride + col];
        if (gridDim.z > 1) {
          // the output was preset to (0, ..., 0, b_in); wrapping integer sums commute, so the
          // result does not depend on the order in which the ranges arrive
          if constexpr (sizeof(T) == 8) atomicAdd((unsigned long long*)dst, (unsigned long long)v);
          else atomicAdd((unsigned int*)dst, (unsigned int)v);
        } else {
          if (col == out_dim) v += in.at(ct0 + c, in_dim);
          *dst = v;
        }
      }
    }
  }
}

// out = (0, ..., 0, b_in): the start value of a keyswitch whose key rows are split over blocks
template <typename T>
__global__ __launch_bounds__(256) void keyswitch_preset_kernel(const LweInput<T> in,
                                                               T* __restrict__ ct_out, size_t out_stride,
                                                               uint32_t in_dim, uint32_t out_dim, size_t count) {
  const uint32_t col = blockIdx.x * 256 + threadIdx.x;
  const size_t c = blockIdx.y;
  if (col > out_dim || c >= count) return;
  ct_out[c * out_stride + col] = (col == out_dim) ? in.at(c, in_dim) : T(0);
}

}  // namespace

template <typename T>
int launch_keyswitch(const LweInput<T>& in, T* ct_out, size_t out_stride, const T* ksk,
                     uint32_t in_dim, uint32_t out_dim, uint32_t base_log, uint32_t level, size_t count,
                     hipStream_t stream) {
  if (count == 0) return CAMD_OK;
  if (in.body != in_dim) return fail(CAMD_EINVAL, "keyswitch: the input's body index is not the input dimension");
  if (level > KS_MAX_LEVEL) return fail(CAMD_EINVAL, "keyswitch: level count above 8 is not supported");
  if (in_dim == 0 || in_dim % KS_CHUNK != 0)
    return fail(CAMD_EINVAL, "keyswitch: input dimension must be a positive multiple of 32");
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
  // A CU holds four of these workgroups. Below KS_FILL workgroups the key rows are split as
  // well, in ranges of at least KS_MIN_SPLIT chunks.
  constexpr uint32_t KS_FILL = 1024, KS_MIN_SPLIT = 4;
  const uint32_t chunks = in_dim / KS_CHUNK;
  const uint64_t blocks = (uint64_t)grid.x * grid.y;
  uint32_t splits = 1;
  if (blocks < KS_FILL) splits = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)(KS_FILL / blocks), chunks / KS_MIN_SPLIT));
  const uint32_t chunks_per_split = (chunks + splits - 1) / splits;
  splits = (chunks + chunks_per_split - 1) / chunks_per_split;
  if (splits > 1) {
    if (count > 65535) return fail(CAMD_EINVAL, "keyswitch: internal split of a large batch");  // cannot happen: blocks < KS_FILL
    hipLaunchKernelGGL(keyswitch_preset_kernel<T>, dim3((ncols + 255) / 256, (unsigned)count), dim3(256), 0, stream,
                       in, ct_out, out_stride, in_dim, out_dim, count);
    CAMD_HIP(hipGetLastError());
  }
  grid.z = splits;
  const size_t lds = (size_t)KS_CHUNK * level * KS_CT_TILE * sizeof(double);
  hipLaunchKernelGGL(keyswitch_kernel<T>, grid, dim3(KS_THREADS), lds, stream, in, ct_out,
                     out_stride, ksk, in_dim, out_dim, base_log, level, count, flush_chunks, chunks_per_split);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

template int launch_keyswitch<uint64_t>(const LweInput<uint64_t>&, uint64_t*, size_t, const uint64_t*, uint32_t,
                                        uint32_t, uint32_t, uint32_t, size_t, hipStream_t);
template int launch_keyswitch<uint32_t>(const LweInput<uint32_t>&, uint32_t*, size_t, const uint32_t*, uint32_t,
                                        uint32_t, uint32_t, uint32_t, size_t, hipStream_t);

}  // namespace camd
