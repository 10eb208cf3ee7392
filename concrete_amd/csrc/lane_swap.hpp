// Register transpositions between lanes of a wave (gfx950): the building blocks of transform
// exchanges that trade register-index bits with lane bits without a round trip through LDS.
// Lane bits 5 and 4 have an instruction for exactly that (v_permlane32_swap / v_permlane16_swap, one
// per register pair); bits 3 and 2 take two row shifts whose bank mask writes only the lanes that
// receive.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace camd {

namespace {

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_into(uint32_t old, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, BANK_MASK, false);
}
template <int PB>
__device__ __forceinline__ void swap_words(uint32_t& a, uint32_t& b) {
  static_assert(PB >= 2 && PB <= 5, "lane bit without a cheap transposition");
  if constexpr (PB == 5) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
  } else if constexpr (PB == 4) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
  } else if constexpr (PB == 3) {   // row_shr:8 into lanes 8..15, row_shl:8 into lanes 0..7
    const uint32_t na = dpp_into<0x118, 0xC>(a, b);
    b = dpp_into<0x108, 0x3>(b, a);
    a = na;
  } else {                          // row_shr:4 into banks 1 and 3, row_shl:4 into banks 0 and 2
    const uint32_t na = dpp_into<0x114, 0xA>(a, b);
    b = dpp_into<0x104, 0x5>(b, a);
    a = na;
  }
}
template <int PB>
__device__ __forceinline__ void swap_complex(double2& a, double2& b) {
  uint32_t a0 = (uint32_t)__double2loint(a.x), a1 = (uint32_t)__double2hiint(a.x);
  uint32_t a2 = (uint32_t)__double2loint(a.y), a3 = (uint32_t)__double2hiint(a.y);
  uint32_t b0 = (uint32_t)__double2loint(b.x), b1 = (uint32_t)__double2hiint(b.x);
  uint32_t b2 = (uint32_t)__double2loint(b.y), b3 = (uint32_t)__double2hiint(b.y);
  swap_words<PB>(a0, b0); swap_words<PB>(a1, b1); swap_words<PB>(a2, b2); swap_words<PB>(a3, b3);
  a = make_double2(__hiloint2double((int)a1, (int)a0), __hiloint2double((int)a3, (int)a2));
  b = make_double2(__hiloint2double((int)b1, (int)b0), __hiloint2double((int)b3, (int)b2));
}

}  // namespace

}  // namespace camd
