// Lane-level check of the four register-bit <-> lane-bit transpositions the register-blocked
// bootstrap kernel uses in place of LDS exchanges: v_permlane32_swap (lane bit 5), v_permlane16_swap
// (bit 4), and pairs of DPP row shifts with a bank mask (bits 3 and 2). Every lane starts with
// A = lane, B = 256 + lane; afterwards the value that sat in register rho of a lane whose bit is
// lambda must sit in register lambda of the lane whose bit is rho. Prints mismatches per bit.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/lane_transpose.hip -o build/lane_transpose
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_into(uint32_t old, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, BANK_MASK, false);
}
template <int PB>
__device__ __forceinline__ void swap_words(uint32_t& a, uint32_t& b) {
  if constexpr (PB == 5) { auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0]; b = r[1]; }
  else if constexpr (PB == 4) { auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0]; b = r[1]; }
  else if constexpr (PB == 3) { const uint32_t na = dpp_into<0x118, 0xC>(a, b); b = dpp_into<0x108, 0x3>(b, a); a = na; }
  else { const uint32_t na = dpp_into<0x114, 0xA>(a, b); b = dpp_into<0x104, 0x5>(b, a); a = na; }
}
template <int PB> __global__ void k(uint32_t* o) {
  uint32_t a = threadIdx.x, b = 256 + threadIdx.x;
  swap_words<PB>(a, b);
  o[threadIdx.x] = a; o[64 + threadIdx.x] = b;
}
template <int PB> int run(uint32_t* d) {
  uint32_t h[128];
  hipLaunchKernelGGL(k<PB>, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int n = 0; n < 2; n++) for (int l = 0; l < 64; l++) {
    const int rho = (l >> PB) & 1, lane_old = (l & ~(1 << PB)) | (n << PB);
    const uint32_t want = (uint32_t)(rho * 256 + lane_old);
    if (h[n * 64 + l] != want) bad++;
  }
  printf("lane bit %d: %d mismatches\n", PB, bad);
  return bad;
}
int main() {
  uint32_t* d; hipMalloc(&d, 512);
  int bad = run<5>(d) + run<4>(d) + run<3>(d) + run<2>(d);
  return bad != 0;
}
