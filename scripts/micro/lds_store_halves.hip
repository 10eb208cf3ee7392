// Bank behaviour of 16-byte LDS accesses under a permuted lane numbering (1024 threads, the
// renumbering exchange of the register-blocked bootstrap kernel at N = 16384). Run under
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- build/lds_store_halves
// and compare per kernel: modes 0/1 load (natural / rotated lanes), 2/3 store with the slot function
// of the hand-over direction, 4/5 store / load with the slot function of the hand-back direction.
// Measured on MI355X: loads never conflict; the rotated store of mode 3, whose eight consecutive
// lanes touch both halves of the banks, takes twice the cycles of modes 2 and 4.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_store_halves.hip -o build/lds_store_halves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// MODE 0: natural read, 1: rotated read (rot_slot), 2: natural write, 3: rotated write
__device__ __forceinline__ uint32_t elem(uint32_t r, uint32_t t, uint32_t lo) { return ((t >> lo) << (lo + 3)) | (r << lo) | (t & ((1u << lo) - 1u)); }
__device__ __forceinline__ uint32_t rot_slot(uint32_t i) { return i ^ (((i >> 7) & 3u) << 2); }
__device__ __forceinline__ uint32_t slot_g(uint32_t i) { return i ^ (((i >> 7) & 3u) << 1); }
template <int MODE> __global__ __launch_bounds__(1024) void k(double2* out, int iters) {
  extern __shared__ double2 buf[];
  const uint32_t t = threadIdx.x;
  const uint32_t tr = (t & ~63u) | ((t & 63u) >> 2) | ((t & 3u) << 4);
  for (uint32_t i = t; i < 8192; i += 1024) buf[i] = make_double2(i, 0);
  __syncthreads();
  double2 acc = make_double2(0, 0);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      uint32_t s;
      if (MODE == 0) s = rot_slot(elem(r, t, 7));
      else if (MODE == 1) s = rot_slot(elem(r, tr, 4));
      else if (MODE == 2) s = rot_slot(elem(r, t, 7));
      else if (MODE == 3) s = rot_slot(elem(r, tr, 4));
      else if (MODE == 4) s = slot_g(elem(r, tr, 4));
      else s = slot_g(elem(r, t, 7));
      if (MODE < 2 || MODE == 5) { double2 v = buf[s]; acc.x += v.x; acc.y += v.y; }
      else buf[s] = make_double2(acc.x + it, r);
    }
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  out[blockIdx.x * 1024 + t] = (MODE < 2 || MODE == 5) ? acc : buf[t];
}
template <int MODE> void run(double2* d) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 8192 * 16, 0, d, 200);
  hipDeviceSynchronize();
}
int main() {
  double2* d; hipMalloc(&d, 256 * 1024 * 16);
  run<0>(d); run<1>(d); run<2>(d); run<3>(d); run<4>(d); run<5>(d);
  printf("done\n");
  return 0;
}
