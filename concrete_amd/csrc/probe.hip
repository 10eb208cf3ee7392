// Measurement helper: the FP64 FMA rate the device sustains, for putting the bootstrap kernel's
// achieved rate next to a measured ceiling rather than only the datasheet figure. Sixteen
// independent v_fma_f64 chains per lane, four waves per SIMD, about 5 ms. (Under a dense FP64 load
// MI355X holds 1.8-2.0 GHz, not its 2.4 GHz maximum: 58-61 TFLOP/s against 78.6 on the datasheet.)
#include "kernels.hpp"

namespace camd {

namespace {

constexpr int PROBE_CHAINS = 16, PROBE_REPS = 8;

__global__ __launch_bounds__(256) void fp64_fma_probe_kernel(double* out, int iters, double a, double b) {
  double x[PROBE_CHAINS];
#pragma unroll
  for (int c = 0; c < PROBE_CHAINS; c++) x[c] = (double)(threadIdx.x + c);
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int rep = 0; rep < PROBE_REPS; rep++)
#pragma unroll
      for (int c = 0; c < PROBE_CHAINS; c++) x[c] = __builtin_fma(x[c], a, b);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < PROBE_CHAINS; c++) s += x[c];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace

int measure_fp64_fma_rate(hipStream_t stream, double* tflops) {
  int dev = 0, cus = 0;
  CAMD_HIP(hipGetDevice(&dev));
  CAMD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const unsigned blocks = (unsigned)(cus > 0 ? cus : 256) * 4u;
  double* out = nullptr;
  CAMD_HIP(hipMalloc(&out, sizeof(double) * 256 * blocks));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  const int iters = 5000;
  float ms = 0;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(fp64_fma_probe_kernel, dim3(blocks), dim3(256), 0, stream, out, 200, 1.0000001, 1e-9);   // clocks settle
    e = hipEventRecord(e0, stream);
    hipLaunchKernelGGL(fp64_fma_probe_kernel, dim3(blocks), dim3(256), 0, stream, out, iters, 1.0000001, 1e-9);
    if (e == hipSuccess) e = hipEventRecord(e1, stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(out);
  if (e != hipSuccess) return fail(CAMD_EHIP, std::string("fp64 probe: ") + hipGetErrorString(e));
  *tflops = 2.0 * PROBE_CHAINS * PROBE_REPS * (double)iters * 256.0 * blocks / ((double)ms * 1e9);
  return CAMD_OK;
}

}  // namespace camd
