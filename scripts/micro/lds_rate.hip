// What the LDS costs four lone waves of a CU (one per SIMD) per instruction form, same bytes moved:
// 16 x ds_read_b64 / 8 x ds_read2_b64 / 8 x ds_read_b128, and the same for stores. Cycles per body
// (s_memtime), addresses conflict-free (lane-contiguous).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_rate.hip -o build/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void lds(double* out, unsigned long long* stamps, int iters) {
  extern __shared__ __align__(16) char smem[];
  const unsigned base = (threadIdx.x >> 6) * 16384u + (threadIdx.x & 63) * (KIND == 2 || KIND == 5 ? 16u : 8u);
  typedef double v2 __attribute__((ext_vector_type(2)));
  v2 r[8];
  for (int i = 0; i < 8; i++) r[i] = (v2){1.0 * threadIdx.x, 2.0 + i};
  for (unsigned i = threadIdx.x; i < 8192; i += blockDim.x) ((double*)smem)[i] = i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (KIND == 0)   // 16 x ds_read_b64, rows 512 B apart
      asm volatile(
          "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
          "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n"
          "s_waitcnt lgkmcnt(0)\n"
          "ds_read_b64 %0, %8 offset:4096\n ds_read_b64 %1, %8 offset:4608\n ds_read_b64 %2, %8 offset:5120\n ds_read_b64 %3, %8 offset:5632\n"
          "ds_read_b64 %4, %8 offset:6144\n ds_read_b64 %5, %8 offset:6656\n ds_read_b64 %6, %8 offset:7168\n ds_read_b64 %7, %8 offset:7680\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r[0].x), "=&v"(r[1].x), "=&v"(r[2].x), "=&v"(r[3].x), "=&v"(r[4].x), "=&v"(r[5].x), "=&v"(r[6].x), "=&v"(r[7].x)
          : "v"(base) : "memory");
    if (KIND == 1)   // 8 x ds_read2_b64
      asm volatile(
          "ds_read2_b64 %0, %8 offset0:0 offset1:64\n ds_read2_b64 %1, %8 offset0:128 offset1:192\n"
          "ds_read2st64_b64 %2, %8 offset0:4 offset1:5\n ds_read2st64_b64 %3, %8 offset0:6 offset1:7\n"
          "ds_read2st64_b64 %4, %8 offset0:8 offset1:9\n ds_read2st64_b64 %5, %8 offset0:10 offset1:11\n"
          "ds_read2st64_b64 %6, %8 offset0:12 offset1:13\n ds_read2st64_b64 %7, %8 offset0:14 offset1:15\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
          : "v"(base) : "memory");
    if (KIND == 2)   // 8 x ds_read_b128, rows 1024 B apart
      asm volatile(
          "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
          "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:6144\n ds_read_b128 %7, %8 offset:7168\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
          : "v"(base) : "memory");
    if (KIND == 3)   // 16 x ds_write_b64
      asm volatile(
          "ds_write_b64 %8, %0\n ds_write_b64 %8, %1 offset:512\n ds_write_b64 %8, %2 offset:1024\n ds_write_b64 %8, %3 offset:1536\n"
          "ds_write_b64 %8, %4 offset:2048\n ds_write_b64 %8, %5 offset:2560\n ds_write_b64 %8, %6 offset:3072\n ds_write_b64 %8, %7 offset:3584\n"
          "ds_write_b64 %8, %0 offset:4096\n ds_write_b64 %8, %1 offset:4608\n ds_write_b64 %8, %2 offset:5120\n ds_write_b64 %8, %3 offset:5632\n"
          "ds_write_b64 %8, %4 offset:6144\n ds_write_b64 %8, %5 offset:6656\n ds_write_b64 %8, %6 offset:7168\n ds_write_b64 %8, %7 offset:7680\n"
          "s_waitcnt lgkmcnt(0)\n"
          :: "v"(r[0].x), "v"(r[1].x), "v"(r[2].x), "v"(r[3].x), "v"(r[4].x), "v"(r[5].x), "v"(r[6].x), "v"(r[7].x), "v"(base) : "memory");
    if (KIND == 4)   // 8 x ds_write2st64_b64
      asm volatile(
          "ds_write2st64_b64 %8, %0, %1 offset0:0 offset1:1\n ds_write2st64_b64 %8, %2, %3 offset0:2 offset1:3\n"
          "ds_write2st64_b64 %8, %4, %5 offset0:4 offset1:5\n ds_write2st64_b64 %8, %6, %7 offset0:6 offset1:7\n"
          "ds_write2st64_b64 %8, %0, %1 offset0:8 offset1:9\n ds_write2st64_b64 %8, %2, %3 offset0:10 offset1:11\n"
          "ds_write2st64_b64 %8, %4, %5 offset0:12 offset1:13\n ds_write2st64_b64 %8, %6, %7 offset0:14 offset1:15\n"
          "s_waitcnt lgkmcnt(0)\n"
          :: "v"(r[0].x), "v"(r[1].x), "v"(r[2].x), "v"(r[3].x), "v"(r[4].x), "v"(r[5].x), "v"(r[6].x), "v"(r[7].x), "v"(base) : "memory");
    if (KIND == 5)   // 8 x ds_write_b128
      asm volatile(
          "ds_write_b128 %8, %0\n ds_write_b128 %8, %1 offset:1024\n ds_write_b128 %8, %2 offset:2048\n ds_write_b128 %8, %3 offset:3072\n"
          "ds_write_b128 %8, %4 offset:4096\n ds_write_b128 %8, %5 offset:5120\n ds_write_b128 %8, %6 offset:6144\n ds_write_b128 %8, %7 offset:7168\n"
          "s_waitcnt lgkmcnt(0)\n"
          :: "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(base) : "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; i++) s += r[i].x + r[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int cus, double* out, unsigned long long* stamps) {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)lds<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  lds<KIND><<<cus, 256, 65536>>>(out, stamps, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(cus);
  hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
  double cyc = 0;
  for (auto v : h) cyc += (double)v;
  printf("%-28s %7.1f cycles per 8 KiB per wave, four waves per CU\n", name, cyc / cus / iters);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out;
  unsigned long long* stamps;
  (void)hipMalloc(&out, sizeof(double) * 256 * cus);
  (void)hipMalloc(&stamps, sizeof(unsigned long long) * cus);
  run<0>("16 x ds_read_b64", cus, out, stamps);
  run<1>("8 x ds_read2(st64)_b64", cus, out, stamps);
  run<2>("8 x ds_read_b128", cus, out, stamps);
  run<3>("16 x ds_write_b64", cus, out, stamps);
  run<4>("8 x ds_write2st64_b64", cus, out, stamps);
  run<5>("8 x ds_write_b128", cus, out, stamps);
  return 0;
}
