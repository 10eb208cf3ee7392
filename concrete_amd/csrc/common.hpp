// Shared host/device helpers of the concrete_amd kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/concrete_amd.h"

namespace camd {

// ---------------------------------------------------------------- errors --
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define CAMD_HIP(expr)                                                                     \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return ::camd::fail(CAMD_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
  } while (0)

// ------------------------------------------------------------ torus traits --
template <typename T>
struct Torus;
template <>
struct Torus<uint64_t> {
  using Signed = int64_t;
  static constexpr int BITS = 64;
};
template <>
struct Torus<uint32_t> {
  using Signed = int32_t;
  static constexpr int BITS = 32;
};

// ------------------------------------------------------ scalar arithmetic --
// pbs_modulus_switch of concrete-core: value in [0, 2N] (2N acts as exponent 0).
template <typename T>
__host__ __device__ inline uint32_t modswitch(T x, uint32_t log2_n) {
  T out = x >> (Torus<T>::BITS - log2_n - 2);
  out += out & T(1);
  out >>= 1;
  return (uint32_t)out;
}

// SignedDecomposer state after closest_representable: (x >> non_rep) + rounding bit.
template <typename T>
__host__ __device__ inline T decomp_init_state(T x, uint32_t base_log, uint32_t level) {
  const uint32_t non_rep = Torus<T>::BITS - base_log * level;
  // the rounding carry may overflow base_log*level bits; the reference drops it (it shifts
  // the rounded value back up and down again), so it is masked here
  return ((x >> non_rep) + ((x >> (non_rep - 1)) & T(1))) & ((T(1) << (base_log * level)) - T(1));
}

// One pass of the signed decomposition (decompose_one_level); digits lie in [-B/2, B/2].
template <typename T>
__host__ __device__ inline typename Torus<T>::Signed decomp_next(T& state, uint32_t base_log) {
  const T mask = (T(1) << base_log) - T(1);
  T res = state & mask;
  state >>= base_log;
  T carry = ((res - T(1)) | state) & res;
  carry >>= (base_log - 1);
  state += carry;
  return (typename Torus<T>::Signed)(res - (carry << base_log));
}

// Coefficient j of p(X) * X^d in Z[X]/(X^N + 1), d in [0, 2N); N a power of two.
template <typename T>
__device__ inline T rotated_coeff(const T* __restrict__ p, uint32_t j, uint32_t d, uint32_t n) {
  const uint32_t s = (j - d) & (2 * n - 1);
  const T v = p[s & (n - 1)];
  return (s & n) ? (T)(T(0) - v) : v;
}

// Real number in torus units -> nearest torus word, wrapping (from_torus of concrete-core,
// with round-to-nearest-even in place of round-half-away: they differ on exact ties only).
template <typename T>
__device__ inline T from_torus(double x);

template <>
__device__ inline uint64_t from_torus<uint64_t>(double x) {
  double fr = x - rint(x);            // [-0.5, 0.5], exact
  fr = (fr >= 0.5) ? -0.5 : fr;       // +1/2 and -1/2 are the same torus point
  const double v = rint(fr * 0x1.0p64);
  const double hi = floor(v * 0x1.0p-32);          // [-2^31, 2^31)
  const double lo = fma(hi, -0x1.0p32, v);         // [0, 2^32), exact
  const uint32_t hi32 = (uint32_t)(int32_t)hi;
  const uint32_t lo32 = (uint32_t)lo;
  return ((uint64_t)hi32 << 32) | (uint64_t)lo32;
}
template <>
__device__ inline uint32_t from_torus<uint32_t>(double x) {
  double fr = x - rint(x);
  fr = (fr >= 0.5) ? -0.5 : fr;
  const double v = rint(fr * 0x1.0p32);
  return (uint32_t)(int32_t)v;
}

// from_torus exactly as concrete-core states it: both roundings are half-away-from-zero and
// the final cast saturates. Slower than from_torus; used where outputs are compared bit for
// bit with the CPU restatement.
__device__ inline double round_half_away(double x) {
  const double t = trunc(x);
  const double d = x - t;
  return (fabs(d) >= 0.5) ? t + copysign(1.0, x) : t;
}
template <typename T>
__device__ inline T from_torus_faithful(double x);
template <>
__device__ inline uint64_t from_torus_faithful<uint64_t>(double x) {
  double fr = x - round_half_away(x);
  fr *= 0x1.0p64;
  fr = round_half_away(fr);
  if (fr >= 0x1.0p63) return 0x7FFFFFFFFFFFFFFFull;
  if (fr <= -0x1.0p63) return 0x8000000000000000ull;
  const double hi = floor(fr * 0x1.0p-32);
  const double lo = fma(hi, -0x1.0p32, fr);
  return ((uint64_t)(uint32_t)(int32_t)hi << 32) | (uint64_t)(uint32_t)lo;
}
template <>
__device__ inline uint32_t from_torus_faithful<uint32_t>(double x) {
  double fr = x - round_half_away(x);
  fr *= 0x1.0p32;
  fr = round_half_away(fr);
  if (fr >= 0x1.0p31) return 0x7FFFFFFFu;
  if (fr <= -0x1.0p31) return 0x80000000u;
  return (uint32_t)(int32_t)fr;
}

// signed torus word -> double
__device__ inline double signed_to_double(int32_t v) { return (double)v; }
__device__ inline double signed_to_double(int64_t v) { return (double)v; }

// ------------------------------------------------------- complex helpers --
__device__ inline double2 cmul(double2 a, double2 b) {
  return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
__device__ inline double2 cmul_conj(double2 a, double2 b) {  // a * conj(b)
  return make_double2(fma(a.x, b.x, a.y * b.y), fma(a.y, b.x, -(a.x * b.y)));
}
__device__ inline double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ inline double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ inline double2 cfma(double2 a, double2 b, double2 c) {  // a*b + c
  return make_double2(fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y)));
}

inline uint32_t ilog2(uint32_t v) {
  uint32_t l = 0;
  while ((1u << l) < v) l++;
  return l;
}

}  // namespace camd
