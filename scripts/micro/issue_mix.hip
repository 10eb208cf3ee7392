// What a lone wave pays per instruction for mixes of v_fma_f64 with 32-bit integer instructions:
// cycles per loop body (s_memtime), one wave per SIMD, bodies written in inline assembly so that the
// instruction sequence is exactly the one named.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/issue_mix.hip -o build/issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define F(i) "v_fma_f64 %" #i ", %" #i ", %16, %17\n"
#define I(i) "v_add_u32 %" #i ", %" #i ", %18\n"
#define MAD(i) "v_mad_u64_u32 %" #i ", vcc, %18, %18, %" #i "\n"
#define FRA(i) "v_fract_f64 %" #i ", %" #i "\n"
#define CVT(i, j) "v_cvt_f64_i32 %" #i ", %" #j "\n"
#define ALB(i) "v_alignbit_b32 %" #i ", %" #i ", %18, 20\n"
#define CHAIN(i, j) "v_add_co_u32 %" #i ", vcc, %" #i ", %18\nv_addc_co_u32 %" #j ", vcc, %" #j ", %18, vcc\n"

template <int KIND>
__global__ void mix(double* out, unsigned long long* stamps, int iters, double a, double b, int k) {
  double x[8];
  int y[8];
#pragma unroll
  for (int c = 0; c < 8; c++) { x[c] = threadIdx.x + c; y[c] = threadIdx.x * c; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
   for (int rep = 0; rep < 16; rep++) {   // 16 bodies per branch: the loop overhead of a lone wave (about 40 cycles) is spread thin
#define OPS "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), \
            "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]) : "v"(a), "v"(b), "v"(k)
    if (KIND == 0) asm volatile(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) : OPS);                                   // 8 f64
    if (KIND == 1) asm volatile(I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15) : OPS);                              // 8 int
    if (KIND == 2) asm volatile(F(0) I(8) F(1) I(9) F(2) I(10) F(3) I(11) F(4) I(12) F(5) I(13) F(6) I(14) F(7) I(15) : OPS);   // 8 x (f64, int)
    if (KIND == 3) asm volatile(F(0) I(8) I(9) F(1) I(10) I(11) F(2) I(12) I(13) F(3) I(14) I(15) F(4) I(8) I(9) F(5) I(10) I(11) F(6) I(12) I(13) F(7) I(14) I(15) : OPS);   // 8 x (f64, int, int)
    if (KIND == 4) asm volatile(F(0) I(8) I(9) I(10) F(1) I(11) I(12) I(13) F(2) I(14) I(15) I(8) F(3) I(9) I(10) I(11) F(4) I(12) I(13) I(14) F(5) I(15) I(8) I(9) F(6) I(10) I(11) I(12) F(7) I(13) I(14) I(15) : OPS);   // 8 x (f64, 3 int)
    if (KIND == 6) asm volatile(MAD(0) MAD(1) MAD(2) MAD(3) MAD(4) MAD(5) MAD(6) MAD(7) : OPS : "vcc");
    if (KIND == 7) asm volatile(FRA(0) FRA(1) FRA(2) FRA(3) FRA(4) FRA(5) FRA(6) FRA(7) : OPS);
    if (KIND == 8) asm volatile(CVT(0, 8) CVT(1, 9) CVT(2, 10) CVT(3, 11) CVT(4, 12) CVT(5, 13) CVT(6, 14) CVT(7, 15) : OPS);
    if (KIND == 9) asm volatile(ALB(8) ALB(9) ALB(10) ALB(11) ALB(12) ALB(13) ALB(14) ALB(15) : OPS);
    if (KIND == 10) asm volatile(CHAIN(8, 9) CHAIN(10, 11) CHAIN(12, 13) CHAIN(14, 15) : OPS : "vcc");
    if (KIND == 5) asm volatile(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15) : OPS);   // 8 f64 then 8 int
   }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int c = 0; c < 8; c++) s += x[c] + y[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int cus, int waves, double* out, unsigned long long* stamps) {
  const int iters = 2000, blocks = cus * waves;
  mix<KIND><<<blocks, 256>>>(out, stamps, iters, 1.0000001, 1e-9, 3);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double cyc = 0;
  for (auto v : h) cyc += (double)v;
  printf("%-34s %d wave(s)/SIMD: %6.1f cycles per body (oldest wave of each workgroup)\n", name, waves, cyc / blocks / iters / 16);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out;
  unsigned long long* stamps;
  (void)hipMalloc(&out, sizeof(double) * 1024 * cus * 4);
  (void)hipMalloc(&stamps, sizeof(unsigned long long) * cus * 4);
  for (int w : {1, 2}) {
    run<0>("8 x v_fma_f64", cus, w, out, stamps);
    run<1>("8 x v_add_u32", cus, w, out, stamps);
    run<2>("8 x (v_fma_f64, v_add_u32)", cus, w, out, stamps);
    run<3>("8 x (v_fma_f64, 2 v_add_u32)", cus, w, out, stamps);
    run<4>("8 x (v_fma_f64, 3 v_add_u32)", cus, w, out, stamps);
    run<5>("8 v_fma_f64 then 8 v_add_u32", cus, w, out, stamps);
    run<6>("8 x v_mad_u64_u32", cus, w, out, stamps);
    run<7>("8 x v_fract_f64", cus, w, out, stamps);
    run<8>("8 x v_cvt_f64_i32", cus, w, out, stamps);
    run<9>("8 x v_alignbit_b32", cus, w, out, stamps);
    run<10>("4 x (v_add_co_u32, v_addc_co_u32)", cus, w, out, stamps);
  }
  return 0;
}
