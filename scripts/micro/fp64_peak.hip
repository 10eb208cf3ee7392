// Measured FP64 vector rate of the device: independent v_fma_f64 chains at one, two and four waves
// per SIMD, with the in-kernel cycle count per instruction and the clock the chip holds meanwhile
// (s_memtime counts shader cycles, s_memrealtime a constant 100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/fp64_peak.hip -o build/fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int CHAINS, int KIND>
__global__ void fma_loop(double* out, unsigned long long* stamps, int iters, double a, double b) {
  double x[CHAINS];
  int y[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) { x[c] = (double)(threadIdx.x + c); y[c] = threadIdx.x + c; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
   for (int rep = 0; rep < 8; rep++)   // 8 x CHAINS instructions per branch
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      if (KIND == 0) x[c] = __builtin_fma(x[c], a, b);
      if (KIND == 1) x[c] = x[c] + a;
      if (KIND == 2) y[c] = y[c] * 3 + i;           // 32-bit integer multiply-add
      if (KIND == 3) { x[c] = __builtin_fma(x[c], a, b); y[c] = (y[c] ^ i) + c; }   // one f64 and one 32-bit op
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += x[c] + y[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, int cus, double* out, unsigned long long* stamps) {
  const int iters = 5000;
  constexpr int CHAINS = 16;
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 256, blocks = cus * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    fma_loop<CHAINS, KIND><<<blocks, threads>>>(out, stamps, 200, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fma_loop<CHAINS, KIND><<<blocks, threads>>>(out, stamps, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < blocks; i++) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    const double per_inst = cyc / blocks / ((double)iters * 8 * CHAINS * (KIND == 3 ? 2 : 1));
    const double ghz = cyc / real * 0.1;
    const double ops = (double)CHAINS * 8 * iters * threads * blocks;
    printf("%-28s %d wave(s)/SIMD: %7.2f ms  %5.2f cycles per wave-instruction per wave  clock %.2f GHz  %.1f T wave-lane-ops/s\n",
           name, waves_per_simd, ms, per_inst, ghz, ops * (KIND == 3 ? 2 : 1) / ms / 1e9);
  }
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out;
  unsigned long long* stamps;
  (void)hipMalloc(&out, sizeof(double) * 1024 * cus * 8);
  (void)hipMalloc(&stamps, sizeof(unsigned long long) * 2 * cus * 8);
  run<0>("v_fma_f64", cus, out, stamps);
  run<1>("v_add_f64", cus, out, stamps);
  run<2>("v_mad_u32 (32-bit integer)", cus, out, stamps);
  run<3>("v_fma_f64 + 32-bit op", cus, out, stamps);
  return 0;
}
