// Helpers shared by the wave-per-ciphertext bootstrap kernels of both polynomial sizes
// (pbs_wave.hip: N = 1024; pbs_wave512.hip: N = 512): torus rounding, signed digits, wave-uniform
// broadcasts, the register type of the parked key rows.
#pragma once
#include <cmath>
#include <cstdlib>

#include "kernels.hpp"

namespace camd {

namespace {

constexpr int WAVES = 4;          // ciphertexts per workgroup

// Small batches go to the kernels that give a ciphertext k+1 waves: up to `fit` ciphertexts per
// CU, where `fit` is the number of such workgroups a CU holds at once. CAMD_WAVE_LATENCY=0 keeps
// the one-wave-per-ciphertext schedules for every batch size; a positive value replaces `fit`.
inline size_t wave_latency_limit(int cus, int fit) {
  static const int setting = []() {
    const char* e = getenv("CAMD_WAVE_LATENCY");
    return e ? atoi(e) : -1;
  }();
  if (setting == 0 || cus <= 0) return 0;
  return (size_t)cus * (size_t)(setting > 0 ? setting : fit);
}

__device__ __forceinline__ double2 mul_neg_i(double2 v) { return make_double2(v.y, -v.x); }
__device__ __forceinline__ double2 mul_pos_i(double2 v) { return make_double2(-v.y, v.x); }

// acc += t (torus units) rounded to 52 fractional bits, in four instructions. With fr = t mod 1,
// b = 1 + fr has the exponent of 1.0, or is exactly 2.0, so its mantissa m is fr in units of 2^-52
// (zero for b = 2.0: a whole turn) and the addend is m << 12: the low dword of b times 2^12 goes in
// as a 32 x 32 + 64 bit multiply-add, the 20 mantissa bits of the high dword are shifted to the top
// of the high dword, where the exponent bits fall off. The 12 bits this drops are 2^-52 of the
// torus, five orders of magnitude below what the f64 transforms themselves leave in a coefficient.
// The two instructions are written out: the compiler rebuilds the shift from the 64-bit pattern in
// three instructions and has no pattern for a multiply-add by a power of two. They are separate
// calls because v_mad_u64_u32 costs an idle slot when it reads a register written by the
// instruction before it: a caller with several sums in flight does all the first halves first.
struct TorusSum {
  uint64_t partial;   // acc with the top 20 mantissa bits added into the high dword
  uint32_t low;       // low dword of b = 1 + (t mod 1)
};
__device__ __forceinline__ TorusSum torus_accumulate_begin(uint64_t acc, double t) {
  const double b = __builtin_amdgcn_fract(t) + 1.0;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 w = __builtin_bit_cast(u32x2, acc);
  asm("v_lshl_add_u32 %0, %1, 12, %0" : "+v"(w.y) : "v"(__double2hiint(b)));
  TorusSum s;
  s.partial = __builtin_bit_cast(uint64_t, w);
  s.low = (uint32_t)__double2loint(b);
  return s;
}
__device__ __forceinline__ uint64_t torus_accumulate_end(const TorusSum& s) {
  uint64_t r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(s.low), "s"(4096u), "v"(s.partial) : "vcc");
  return r;
}
__device__ __forceinline__ uint64_t torus_accumulate(uint64_t acc, double t) {
  return torus_accumulate_end(torus_accumulate_begin(acc, t));
}
// 32-bit torus: with fr = t mod 1 in [0, 1), adding 1.5*2^20 leaves round(fr*2^32) in the low dword
// of the sum (a carry out of it is a whole turn of the torus)
__device__ __forceinline__ uint32_t torus_accumulate(uint32_t acc, double t) {
  const double fr = __builtin_amdgcn_fract(t);
  return acc + (uint32_t)__double2loint(fr + 0x1.8p20);
}

// Decomposition state in 32 bits. Only the top BL*L bits of the word (plus the rounding bit
// below them) enter the digits; when those lie in the high dword the 64-bit shifts of the
// generic form are not needed. The result equals decomp_init_state() for every input.
template <typename T, int BL, int L>
__device__ __forceinline__ uint32_t digit_state32(T x) {
  constexpr int BITS = Torus<T>::BITS;
  constexpr int NON_REP = BITS - BL * L;
  static_assert(BL * L <= 31, "decomposition wider than 31 bits");
  if constexpr (BITS == 64 && NON_REP >= 33) {
    const uint32_t hi = (uint32_t)(x >> 32);
    // (hi >> s) + rounding bit == (hi + 2^(s-1)) >> s up to a carry out of bit 31, which the
    // mask would drop anyway
    constexpr int S = NON_REP - 32;
    return ((hi + (1u << (S - 1))) >> S) & ((1u << (BL * L)) - 1u);
  } else if constexpr (BITS == 32) {
    // ((x >> (NON_REP - 1)) + 1) >> 1, masked to BL L bits: the same bits as one add and one shift
    // (a carry out of bit 31 is the masked-off bit)
    return ((uint32_t)x + (1u << (NON_REP - 1))) >> NON_REP;
  } else {
    return (uint32_t)decomp_init_state<T>(x, BL, L);
  }
}
// next signed digit (level L first); same recurrence as decompose_one_level, in 32 bits
template <int BL, int L>
__device__ __forceinline__ int32_t digit_next32(uint32_t& state) {
  if constexpr (L == 1) {
    // single level: state already holds BL bits; carry = (res > B/2)
    const uint32_t res = state;
    return (int32_t)res - ((res > (1u << (BL - 1))) ? (1 << BL) : 0);
  } else {
    // decomp_next() in four instructions, for states below 2^(32 - BL): the carry (res > B/2, or
    // res == B/2 and the top bit of the next digit's field set) is the carry out of bit BL - 1 of
    // res + (B/2 - 1) + that bit
    const uint32_t next = (state + ((1u << (BL - 1)) - 1u) + __builtin_amdgcn_ubfe(state, 2 * BL - 1, 1)) >> BL;
    const int32_t digit = (int32_t)(state - (next << BL));
    state = next;
    return digit;
  }
}

// Single-level decomposition in one go: the signed digit of x. With state = the top BL bits of x
// rounded to nearest, the digit is state for state <= B/2 and state - B above. Both the rounding
// and the fold are additions below the extracted field, so they merge into one constant:
//   ((hi + 2^(S-1) + (C << S)) >> S) mod B  ==  (state + C) mod B,   C = B/2 - 1,
// and (state + C) mod B - C is state for state <= B/2 (no wrap) and state - B above (one wrap).
// Three integer instructions and no condition register, against six and a compare.
template <typename T, int BL>
__device__ __forceinline__ int32_t digit_single32(T x) {
  constexpr int BITS = Torus<T>::BITS;
  constexpr int NON_REP = BITS - BL;
  if constexpr (BITS == 64 && NON_REP >= 33) {
    constexpr int S = NON_REP - 32;
    constexpr uint32_t C = (1u << (BL - 1)) - 1u;
    const uint32_t hi = (uint32_t)(x >> 32);
    const uint32_t w = __builtin_amdgcn_ubfe(hi + ((1u << (S - 1)) + (C << S)), S, BL);
    return (int32_t)w - (int32_t)C;
  } else {
    uint32_t st = digit_state32<T, BL, 1>(x);
    return digit_next32<BL, 1>(st);
  }
}

// Signed single-level digit of (wrapped ? -rot : rot) - own, where `shifted` = j - a_hat carries the
// wrap in bit LOGN. For a 64-bit torus whose digit lies in the high dword only the high dwords
// enter: the difference has the high dword (rot_hi ^ m) - own_hi with m = 0 or all ones, up to the
// borrow of the low halves (and the +1 of a negation whose low dword is zero). The digit is that
// value rounded at bit 32 - BL, so leaving the borrow out moves the value being rounded by at most
// 2^-32 of the torus against a rounding step of 2^-BL: the decomposition error bound B^-1/2 grows
// by a factor 1 + 2^(BL-31). The rounding is an arithmetic shift: digits in [-B/2, B/2) where the
// generic form has (-B/2, B/2]; the two ends are the same residue.
template <typename T, int BL, int LOGN>
__device__ __forceinline__ int32_t digit_of_rotated_difference(T rot, T own, uint32_t shifted) {
  if constexpr (sizeof(T) == 8 && BL <= 31) {
    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int32_t)shifted, LOGN, 1);
    constexpr uint32_t HALF = 1u << (32 - BL - 1);
    uint32_t d;   // (rot_hi ^ m) + HALF in one instruction; the compiler reassociates the sum and loses the pattern
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "v"((uint32_t)(rot >> 32)), "v"(m), "s"(HALF));
    return (int32_t)(d - (uint32_t)(own >> 32)) >> (32 - BL);
  } else {
    const T r = ((shifted >> LOGN) & 1u) ? (T)(T(0) - rot) : rot;
    return digit_single32<T, BL>((T)(r - own));
  }
}

// the same digit for an accumulator that holds the high dwords only (32 bits per coefficient)
template <int BL, int LOGN>
__device__ __forceinline__ int32_t digit_of_rotated_difference32(uint32_t rot_hi, uint32_t own_hi, uint32_t shifted) {
  const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int32_t)shifted, LOGN, 1);
  constexpr uint32_t HALF = 1u << (32 - BL - 1);
  uint32_t d;
  asm("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(rot_hi), "v"(m), "s"(HALF));
  return (int32_t)(d - own_hi) >> (32 - BL);
}

// broadcast lane 0's value through SGPRs so that the compiler sees a wave-uniform value
__device__ __forceinline__ uint64_t wave_uniform(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t wave_uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 as_double2(v4u v) {
  return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}

}  // namespace

}  // namespace camd
