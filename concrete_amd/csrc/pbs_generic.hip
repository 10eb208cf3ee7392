// Programmable bootstrap, workgroup-per-ciphertext family: any (n, k, N, level, base_log).
//
// Replaces FftEngine::discard_bootstrap_lwe_ciphertext as the reference calls it at
// concrete-shortint/src/engine/server_side/mod.rs:148-153 and
// concrete-boolean/src/engine/bootstrapping/cpu.rs:231-236: modulus switch, blind rotation as
// n CMUXes (rotate, subtract, signed decomposition, forward negacyclic FFT, multiply-accumulate
// against the Fourier bootstrap key, inverse FFT, round, add) and sample extraction of
// coefficient 0.
//
// This family is the fallback that covers every parameter set, including N = 8192 and 16384
// whose accumulator does not fit LDS. One workgroup walks one ciphertext through all n
// iterations. The FFT runs in place in LDS (radix-2, natural order in, bit-reversed order
// out; the inverse consumes the bit-reversed order, so no reordering pass exists and the
// Fourier key is stored in that same order). The accumulator and the Fourier-domain sums live
// in a per-workgroup global scratch slab that stays L2-resident.
//
// Floating-point contract: contraction is off in this file and every fused multiply-add is an
// explicit fma(). The sequence of IEEE-754 double operations is the same as in the CPU
// restatement used by the tests (same radix-2 butterflies, same twiddle values computed on the
// host, same accumulation order, half-away rounding in from_torus), so output ciphertexts can
// be compared bit for bit.
#pragma clang fp contract(off)
#include <cmath>
#include <vector>

#include "kernels.hpp"

namespace camd {

namespace {

// in-place DIF on M complex points in LDS; all threads of the workgroup call it
__device__ inline void lds_fft_forward(double2* buf, const double2* __restrict__ stage_tw, uint32_t M) {
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  for (uint32_t len = M; len >= 2; len >>= 1) {
    const uint32_t half = len >> 1, stride = M / len;
    for (uint32_t t = tid; t < (M >> 1); t += nt) {
      const uint32_t blk = t / half, j = t - blk * half;
      const uint32_t i0 = blk * len + j, i1 = i0 + half;
      const double2 a = buf[i0], b = buf[i1];
      const double2 w = stage_tw[j * stride];
      buf[i0] = cadd(a, b);
      buf[i1] = cmul(csub(a, b), w);
    }
    __syncthreads();
  }
}
// in-place unnormalised inverse (DIT), bit-reversed order in, natural order out
__device__ inline void lds_fft_inverse(double2* buf, const double2* __restrict__ stage_tw, uint32_t M) {
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  for (uint32_t len = 2; len <= M; len <<= 1) {
    const uint32_t half = len >> 1, stride = M / len;
    for (uint32_t t = tid; t < (M >> 1); t += nt) {
      const uint32_t blk = t / half, j = t - blk * half;
      const uint32_t i0 = blk * len + j, i1 = i0 + half;
      const double2 a = buf[i0];
      const double2 b = cmul_conj(buf[i1], stage_tw[j * stride]);
      buf[i0] = cadd(a, b);
      buf[i1] = csub(a, b);
    }
    __syncthreads();
  }
}

template <typename T>
__global__ void generic_bsk_fourier_kernel(const T* __restrict__ bsk, double2* __restrict__ out,
                                           const double2* __restrict__ tables, uint32_t N) {
  using S = typename Torus<T>::Signed;
  extern __shared__ double2 buf[];
  const uint32_t M = N >> 1;
  const double2* twist = tables;
  const double2* stage_tw = tables + M;
  const T* poly = bsk + (size_t)blockIdx.x * N;
  const double scale = (Torus<T>::BITS == 64) ? 0x1.0p-64 : 0x1.0p-32;
  for (uint32_t j = threadIdx.x; j < M; j += blockDim.x) {
    const double re = signed_to_double((S)poly[j]) * scale;
    const double im = signed_to_double((S)poly[j + M]) * scale;
    buf[j] = cmul(make_double2(re, im), twist[j]);
  }
  __syncthreads();
  lds_fft_forward(buf, stage_tw, M);
  double2* dst = out + (size_t)blockIdx.x * M;
  for (uint32_t j = threadIdx.x; j < M; j += blockDim.x) dst[j] = buf[j];
}

template <typename T>
__global__ void generic_pbs_kernel(PbsLaunch<T> a) {
  using S = typename Torus<T>::Signed;
  extern __shared__ double2 buf[];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t N = a.N, M = N >> 1, k = a.k, n = a.n, L = a.level, bl = a.base_log;
  const uint32_t G = (k + 1) * N;
  const uint32_t logN = 31 - __clz(N);
  const double2* twist = a.tables;
  const double2* stage_tw = a.tables + M;
  const size_t slab = (size_t)G * sizeof(T) + (size_t)(k + 1) * M * sizeof(double2);
  char* my = (char*)a.scratch + (size_t)blockIdx.x * slab;
  double2* accf = (double2*)my;                                 // [(k+1)][M]
  T* ct0 = (T*)(my + (size_t)(k + 1) * M * sizeof(double2));   // [(k+1)][N]
  const double inv_m = 1.0 / (double)M;

  for (size_t ct = blockIdx.x; ct < a.count; ct += gridDim.x) {
    const LweInputRow<T> in = input_row(a, ct);
    const uint32_t lut_id = checked_lut_id(a, ct);
    const T* lut = a.luts + (size_t)lut_id * G;

    // ct0 <- lut * X^{-b_hat}
    const uint32_t b_hat = modswitch<T>(in[n], logN);
    const uint32_t d0 = (2 * N - b_hat) & (2 * N - 1);
    for (uint32_t idx = tid; idx < G; idx += nt) {
      const uint32_t c = idx / N, j = idx - c * N;
      ct0[idx] = rotated_coeff<T>(lut + (size_t)c * N, j, d0, N);
    }
    __syncthreads();

    for (uint32_t i = 0; i < n; i++) {
      const T ai = in[i];
      if (ai == 0) continue;  // uniform over the workgroup
      const uint32_t a_hat = modswitch<T>(ai, logN) & (2 * N - 1);
      bool first = true;
      for (uint32_t l = L; l >= 1; l--) {
        for (uint32_t r = 0; r <= k; r++) {
          const T* p = ct0 + (size_t)r * N;
          for (uint32_t j = tid; j < M; j += nt) {
            const T x0 = rotated_coeff<T>(p, j, a_hat, N) - p[j];
            const T x1 = rotated_coeff<T>(p, j + M, a_hat, N) - p[j + M];
            T s0 = decomp_init_state<T>(x0, bl, L), s1 = decomp_init_state<T>(x1, bl, L);
            S g0 = 0, g1 = 0;
            for (uint32_t lv = L; lv >= l; lv--) {
              g0 = decomp_next<T>(s0, bl);
              g1 = decomp_next<T>(s1, bl);
            }
            buf[j] = cmul(make_double2(signed_to_double(g0), signed_to_double(g1)), twist[j]);
          }
          __syncthreads();
          lds_fft_forward(buf, stage_tw, M);
          for (uint32_t c = 0; c <= k; c++) {
            const double2* row = a.fbsk + ((((size_t)i * L + (l - 1)) * (k + 1) + r) * (k + 1) + c) * M;
            double2* dst = accf + (size_t)c * M;
            for (uint32_t f = tid; f < M; f += nt) {
              const double2 v = buf[f], w = row[f];
              dst[f] = first ? cmul(w, v) : cfma(w, v, dst[f]);
            }
          }
          first = false;
          __syncthreads();
        }
      }
      for (uint32_t c = 0; c <= k; c++) {
        const double2* src = accf + (size_t)c * M;
        for (uint32_t f = tid; f < M; f += nt) buf[f] = src[f];
        __syncthreads();
        lds_fft_inverse(buf, stage_tw, M);
        T* p = ct0 + (size_t)c * N;
        for (uint32_t j = tid; j < M; j += nt) {
          const double2 v = cmul_conj(buf[j], twist[j]);
          p[j] += from_torus_faithful<T>(v.x * inv_m);
          p[j + M] += from_torus_faithful<T>(v.y * inv_m);
        }
        __syncthreads();
      }
    }

    // sample extraction of coefficient 0
    T* out = a.ct_out + ct * a.out_stride;
    for (uint32_t idx = tid; idx < k * N; idx += nt) {
      const uint32_t c = idx / N, j = idx - c * N;
      const T* p = ct0 + (size_t)c * N;
      out[idx] = (j == 0) ? p[0] : (T)(T(0) - p[N - j]);
    }
    if (tid == 0) out[k * N] = ct0[(size_t)k * N];
    __syncthreads();
  }
}

unsigned generic_threads(uint32_t N) {
  unsigned t = N / 4;  // one butterfly per thread per stage
  if (t < 64) t = 64;
  if (t > 1024) t = 1024;
  return t;
}

}  // namespace

// bit-reversed frequency order of an in-place decimation-in-frequency transform, unscaled
uint64_t generic_layout_tag(const camd_params& p) {
  constexpr uint64_t REVISION = 1;
  return (REVISION << 32) | p.polynomial_size;
}

size_t generic_tables_count(uint32_t N) { return (size_t)(N / 2) + (size_t)(N / 4 ? N / 4 : 1); }

// cos and sin go through volatile pointers so that the compiler cannot merge the pair into one
// sincos() call: the table then holds the same libm results as the CPU restatement's.
static double (*volatile libm_cos)(double) = ::cos;
static double (*volatile libm_sin)(double) = ::sin;

void generic_fill_tables(uint32_t N, double2* t) {
  const uint32_t M = N / 2;
  for (uint32_t j = 0; j < M; j++) {
    const double ang = M_PI * (double)j / (double)N;
    t[j] = make_double2(libm_cos(ang), libm_sin(ang));
  }
  for (uint32_t j = 0; j < M / 2; j++) {
    const double ang = -2.0 * M_PI * (double)j / (double)M;
    t[M + j] = make_double2(libm_cos(ang), libm_sin(ang));
  }
}

size_t generic_scratch_bytes(uint32_t k, uint32_t N, int torus_bytes, unsigned* grid_out, size_t count) {
  const size_t slab = (size_t)(k + 1) * N * torus_bytes + (size_t)(k + 1) * (N / 2) * sizeof(double2);
  // enough workgroups to fill 256 CUs at the occupancy the LDS footprint allows
  const size_t lds = (size_t)(N / 2) * sizeof(double2);
  size_t per_cu = (160 * 1024) / (lds ? lds : 1);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  size_t grid = 256 * per_cu;
  if (grid > count) grid = count;
  if (grid < 1) grid = 1;
  *grid_out = (unsigned)grid;
  return slab * grid;
}

template <typename T>
int generic_bsk_to_fourier(const T* bsk_std, double2* fbsk, const double2* tables, uint32_t N, size_t polys,
                           hipStream_t stream) {
  const size_t lds = (size_t)(N / 2) * sizeof(double2);
  if (lds > 64 * 1024)
    CAMD_HIP(hipFuncSetAttribute((const void*)generic_bsk_fourier_kernel<T>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(generic_bsk_fourier_kernel<T>, dim3((unsigned)polys), dim3(generic_threads(N)), lds, stream,
                     bsk_std, fbsk, tables, N);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

template <typename T>
int generic_launch_pbs(const PbsLaunch<T>& a, hipStream_t stream) {
  if (a.count == 0) return CAMD_OK;
  unsigned grid = 1;
  generic_scratch_bytes(a.k, a.N, (int)sizeof(T), &grid, a.count);
  const size_t lds = (size_t)(a.N / 2) * sizeof(double2);
  if (lds > 64 * 1024)
    CAMD_HIP(hipFuncSetAttribute((const void*)generic_pbs_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
  hipLaunchKernelGGL(generic_pbs_kernel<T>, dim3(grid), dim3(generic_threads(a.N)), lds, stream, a);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

template int generic_bsk_to_fourier<uint64_t>(const uint64_t*, double2*, const double2*, uint32_t, size_t,
                                              hipStream_t);
template int generic_bsk_to_fourier<uint32_t>(const uint32_t*, double2*, const double2*, uint32_t, size_t,
                                              hipStream_t);
template int generic_launch_pbs<uint64_t>(const PbsLaunch<uint64_t>&, hipStream_t);
template int generic_launch_pbs<uint32_t>(const PbsLaunch<uint32_t>&, hipStream_t);

}  // namespace camd
