// Programmable bootstrap for the largest polynomial size (N = 16384, k = 1, two or three levels):
// the throughput kernel of the PARAM_MESSAGE_4_CARRY_4 class.
//
// Replaces FftEngine::discard_bootstrap_lwe_ciphertext as the reference calls it at
// concrete-shortint/src/engine/server_side/mod.rs:148-153, for the parameter sets of
// concrete-shortint/src/parameters/mod.rs:665-808 (N = 16384). Same contract as the other kernels
// of the wave family: the reference's algorithm with its own order of f64 operations, so
// decryptions and the noise bound are the reference's and the last bits of a ciphertext are not;
// the bit-exact kernel for these sets is pbs_block.hip.
//
// One 512-thread workgroup (two waves per SIMD) walks one ciphertext through its n CMUXes.
//   * The 8192-point transform is a four-step one: thread t holds the sixteen points
//     512 a + t and transforms them in registers (radix 16, the twist folded into its input and
//     output factors); ONE transposition through a 128 KB LDS image hands wave w the two
//     512-point sub-transforms 2w and 2w + 1, which run on that wave alone, eight points per
//     lane: the exchange of lane bits 5:3 in registers (v_permlane32_swap, v_permlane16_swap, row
//     shifts), that of lane bits 2:0 through the image row the wave has just read (or is about to
//     write). A transform costs two workgroup barriers, around the transposition.
//   * The Fourier-domain sums of both output columns stay in registers for the whole CMUX
//     (2 x 16 complex values per thread); the key (1 MB per CMUX for two levels) streams from
//     the L2 into the multiply-accumulate of each sub-transform.
//   * The accumulator is kept to 32 bits per coefficient (the high dwords of the torus words;
//     digits come from the high dwords anyway, and each CMUX's increment is rounded at 2^-32 of
//     the torus, orders of magnitude below the output noise) in a per-workgroup slab of 128 KB,
//     coefficients j and j + N/2 side by side: the rotated difference of a point is two 8-byte
//     loads, the update one load and one store.
//   * Signed digits are the balanced ones (each in [-B/2, B/2), lowest level first from a value
//     rounded to L base_log bits): they represent the same rounded value as the reference's.
#include "lane_swap.hpp"
#include "wave1024.hpp"

namespace camd {

namespace {

namespace w16 {

#ifndef W16_KEY_AUX
#define W16_KEY_AUX 0   // cache-policy bits of the key loads (A/B builds)
#endif

constexpr int THREADS = 512;
// Sizes that follow from the polynomial size. A polynomial's transform has N/1024 rows of 512 points;
// a workgroup always works on eight rows or sixteen: one ciphertext at N = 16384 (sixteen points per
// thread, two sub-transforms per wave) and N = 8192 (eight points, one sub-transform), two
// ciphertexts at N = 4096 (four points per thread and ciphertext) and four at N = 2048 (two points).
template <int LOGN_>
struct Geo {
  static constexpr int LOGN = LOGN_, N = 1 << LOGN_, M = N / 2, LOGM = LOGN_ - 1;
  static constexpr int PTS = M / THREADS;                  // points per thread and ciphertext: the radix of the step
  static constexpr int CPW = PTS >= 8 ? 1 : 8 / PTS;       // ciphertexts per workgroup
  static constexpr int SUBS = PTS >= 8 ? PTS / 8 : 1;      // 512-point sub-transforms per wave
  static constexpr int ROWS = CPW * PTS;                   // rows of the transposition image
  static_assert(PTS == 16 || PTS == 8 || PTS == 4 || PTS == 2, "polynomial size without an instance");
};

// table buffer, in double2 entries: the folded constants of the 512-point sub-transforms (the LDS
// image of the kernels) and the two per-thread factors of the four-step transform
constexpr int T_A = 0;                 // [pair 9][lane]: forward, second pass: input q of lane (hi,lo) carries e^{-2 pi i (8q+lo) hi/512}
constexpr int T_B = T_A + 9 * 64;      // [pair 8][lo]:   input q carries e^{-2 pi i lo q/64}
constexpr int T_C = T_B + 8 * 8;       // [pair 9][lane]: inverse, third pass: input b carries the conjugate of e^{-2 pi i lane b/512}
constexpr int T_LDS = T_C + 9 * 64;    // entries staged in LDS
constexpr int T_W = T_LDS;             // [t]: e^{i pi t/N}
constexpr int T_V = T_W + THREADS;     // [t]: e^{-2 pi i t/(N/2)} = v
constexpr int T_V2 = T_V + THREADS;    // [t]: v^2
constexpr int T_V4 = T_V2 + THREADS;   // [t]: v^4
constexpr int T_V8 = T_V4 + THREADS;   // [t]: v^8
constexpr int T_TOTAL = T_V8 + THREADS;

template <int LOGN>
constexpr size_t image_bytes() { return (size_t)Geo<LOGN>::ROWS * THREADS * sizeof(double2); }   // 128 KB or 64 KB
template <int LOGN>
constexpr size_t slab_bytes() { return 2 * (size_t)Geo<LOGN>::N * sizeof(uint32_t); }   // accumulator of one ciphertext (k + 1 = 2 polynomials)
template <int LOGN>
constexpr size_t group_slab_bytes() { return Geo<LOGN>::CPW * slab_bytes<LOGN>(); }   // of a workgroup's ciphertexts
// Two levels: what the lower level's digit leaves of a rotated difference IS the upper level's digit, 16
// bits of it. A thread parks those of its points (two coefficients to a dword) behind the accumulators
// while it forms the lower level's transform input and reads them back for the upper level's, instead
// of gathering the difference again: 4 bytes per point and direction in place of 16 gathered.
template <int LOGN>
constexpr size_t parked_bytes() { return (size_t)Geo<LOGN>::ROWS * THREADS * sizeof(uint32_t); }
// Distance between the accumulators of consecutive workgroups. A power of two makes the 32 slabs of an XCD
// fall on the same sets of its L2; one page more costs nothing and measured 3 to 4 % faster at
// N = 16384 and N = 4096 (no difference at 8192 and 2048; any pad from 4 to 40 KB did the same). The
// three-level instance is the exception (138 ms per 512 unpadded against 145 to 157 padded) and keeps
// the power of two. The scratch is sized for the padded stride in either case.
constexpr size_t SLAB_PAD = 4096;
template <int LOGN, int L>
constexpr size_t group_stride_bytes() { return group_slab_bytes<LOGN>() + (L == 3 ? 0 : SLAB_PAD); }
// ... in the LDS where the image leaves room for them (up to N = 8192), else in that global scratch
// Up to N = 8192 the LDS holds TWO images, used in turn: a transform's image is then written while the
// previous one's rows are still being read, and the barrier in front of every image write (the one
// that stands behind a phase of memory waits) is not needed; one barrier per image remains.
template <int LOGN>
constexpr int images() { return T_LDS * sizeof(double2) + 2 * image_bytes<LOGN>() <= 160 * 1024 ? 2 : 1; }
template <int LOGN>
constexpr size_t images_bytes() { return images<LOGN>() * image_bytes<LOGN>(); }
template <int LOGN>
constexpr bool parked_in_lds() { return T_LDS * sizeof(double2) + images_bytes<LOGN>() + parked_bytes<LOGN>() <= 160 * 1024; }
// N = 16384: the LDS that image and tables leave (13 KB) takes the digits of the first four points of
// every thread (8 KB); the other twelve go to the global scratch (2 % faster than all sixteen there)
template <int LOGN>
constexpr int parked_lds_groups() {
  return parked_in_lds<LOGN>() ? 0 : (int)((160 * 1024 - T_LDS * sizeof(double2) - images_bytes<LOGN>()) / (THREADS * sizeof(uint4)));
}
template <int LOGN>
constexpr size_t lds_bytes() {
  return T_LDS * sizeof(double2) + images_bytes<LOGN>() + (parked_in_lds<LOGN>() ? parked_bytes<LOGN>() : parked_lds_groups<LOGN>() * THREADS * sizeof(uint4));
}

// e^{i pi a/32}, a = 0..15: the per-register part of the twist
__device__ __forceinline__ double2 twist16(int a) {
  constexpr double C[16] = {1.0, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494,
                            0.92387953251128675613, 0.88192126434835502971, 0.83146961230254523708, 0.77301045336273696081,
                            0.70710678118654752440, 0.63439328416364549822, 0.55557023301960222474, 0.47139673682599764856,
                            0.38268343236508977173, 0.29028467725446236764, 0.19509032201612826785, 0.09801714032956060199};
  return make_double2(C[a], a == 0 ? 0.0 : C[16 - a]);
}
// e^{-2 pi i j/16}
__device__ __forceinline__ double2 w16(int j) {
  constexpr double C[5] = {1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173, 0.0};
  j &= 15;
  const int q = j >> 2, r = j & 3;                       // angle = -(q * 90 + r * 22.5) degrees
  const double c = C[r], s = C[4 - r];                   // cos, sin of r * 22.5 degrees
  // rotate (c, -s) by -q * 90 degrees
  switch (q) {
    case 0: return make_double2(c, -s);
    case 1: return make_double2(-s, -c);
    case 2: return make_double2(-c, s);
    default: return make_double2(s, c);
  }
}

template <bool INV>
__device__ __forceinline__ void dft4(double2& a, double2& b, double2& c, double2& d) {
  const double2 s0 = cadd(a, c), d0 = csub(a, c), s1 = cadd(b, d), d1 = csub(b, d);
  const double2 r = INV ? mul_pos_i(d1) : mul_neg_i(d1);
  a = cadd(s0, s1);
  c = csub(s0, s1);
  b = cadd(d0, r);
  d = csub(d0, r);
}
// 16-point DFT in registers, natural order in and out; INV selects e^{+2 pi i jk/16}
template <bool INV>
__device__ __forceinline__ void dft16(double2 (&x)[16]) {
  // a = 4 alpha + beta, k = kappa + 4 lambda
#pragma unroll
  for (int beta = 0; beta < 4; beta++) dft4<INV>(x[beta], x[4 + beta], x[8 + beta], x[12 + beta]);   // over alpha: x[4 kappa + beta]
#pragma unroll
  for (int kappa = 1; kappa < 4; kappa++)
#pragma unroll
    for (int beta = 1; beta < 4; beta++) {
      const double2 w = w16(beta * kappa);
      x[4 * kappa + beta] = INV ? cmul_conj(x[4 * kappa + beta], w) : cmul(x[4 * kappa + beta], w);
    }
  double2 y[16];
#pragma unroll
  for (int kappa = 0; kappa < 4; kappa++) {
    dft4<INV>(x[4 * kappa], x[4 * kappa + 1], x[4 * kappa + 2], x[4 * kappa + 3]);                  // over beta: lambda
#pragma unroll
    for (int lambda = 0; lambda < 4; lambda++) y[kappa + 4 * lambda] = x[4 * kappa + lambda];
  }
#pragma unroll
  for (int k = 0; k < 16; k++) x[k] = y[k];
}

// ---- lane exchanges of the 512-point sub-transforms ----
// x[q] of lane (h, lo) becomes what x[h] of lane (q, lo) was: register bits <-> lane bits 5:3, in registers
__device__ __forceinline__ void transpose_5_3(double2 (&x)[8]) {
#pragma unroll
  for (int r = 0; r < 8; r++)
    if (!(r & 4)) swap_complex<5>(x[r], x[r | 4]);
#pragma unroll
  for (int r = 0; r < 8; r++)
    if (!(r & 2)) swap_complex<4>(x[r], x[r | 2]);
#pragma unroll
  for (int r = 0; r < 8; r++)
    if (!(r & 1)) swap_complex<3>(x[r], x[r | 1]);
}
// x[q] of lane (hi, l) becomes what x[l] of lane (hi, q) was: register bits <-> lane bits 2:0. In
// registers this takes a row shift and two quad permutes with selects per word, 160 instructions;
// it goes through 8 KB of LDS that belong to the wave alone instead (the image row of the
// sub-transform it is working on): sixteen 16-byte accesses and as many address XORs. The vector
// ALU is what this kernel runs out of, the LDS is mostly idle. Lane (hi, l) stores register r in
// row 8 hi + (l ^ f), slot r ^ l, with f = bit 1 of hi: the slot XOR keeps the eight lanes of a
// store group on different banks, f keeps the two lanes of a 16-lane load group that share l on
// different halves of the 64 banks. No barrier: the LDS executes one wave's accesses in order.
__device__ __forceinline__ void transpose_2_0_lds(double2 (&x)[8], int lane, double2* row) {
  const uint32_t hi = (uint32_t)lane >> 3, l = (uint32_t)lane & 7, f = (hi >> 1) & 1;
  char* base = (char*)row;
  const uint32_t wbase = (hi * 8 + (l ^ f)) * 128 + l * 16;     // + (r * 16 XORed in)
  const uint32_t rbase = (hi * 8 + f) * 128 + l * 16;           // + (q * 128 and q * 16 XORed in)
#pragma unroll
  for (int r = 0; r < 8; r++) *(double2*)(base + (wbase ^ (uint32_t)(r * 16))) = x[r];
#pragma unroll
  for (int q = 0; q < 8; q++) x[q] = *(const double2*)(base + (rbase ^ (uint32_t)(q * 144)));
}

// 512-point transform on one wave: x[a] = point 64 a + lane in, x[g] = X[64 g + 8 (lane & 7) + (lane >> 3)] out
__device__ __forceinline__ void fft512_plain_forward_head(double2 (&x)[8], int lane, const double2* tab) {
#ifdef W16_DEBUG_NO_FFT512
  return;
#endif
  dft8<false>(x);
  transpose_5_3(x);
  {
    const Tw8 k = load_tw8<true>(tab + T_A + lane, 64);
    dft8_tw<false, true>(x, k);
  }
}
// the last of the three passes, apart so that a caller can start loads in front of it
__device__ __forceinline__ void fft512_plain_forward_tail(double2 (&x)[8], int lane, const double2* tab, double2* row) {
#ifdef W16_DEBUG_NO_FFT512
  return;
#endif
  transpose_2_0_lds(x, lane, row);
  const Tw8 k = load_tw8<false>(tab + T_B + (lane & 7), 8);
  dft8_tw<false, false>(x, k);
}
__device__ __forceinline__ void fft512_plain_forward(double2 (&x)[8], int lane, const double2* tab, double2* row) {
  fft512_plain_forward_head(x, lane, tab);
  fft512_plain_forward_tail(x, lane, tab, row);
}
// unnormalised inverse: spectrum in the order above in, x[a] = 512 * point (64 a + lane) out
__device__ __forceinline__ void fft512_plain_inverse(double2 (&x)[8], int lane, const double2* tab, double2* row) {
#ifdef W16_DEBUG_NO_FFT512
  return;
#endif
  dft8<true>(x);
  transpose_2_0_lds(x, lane, row);
  {
    const Tw8 k = load_tw8<false>(tab + T_B + (lane & 7), 8);
    dft8_tw<true, false>(x, k);
  }
  transpose_5_3(x);
  {
    const Tw8 k = load_tw8<false>(tab + T_C + lane, 64);
    dft8_tw<true, false>(x, k);
  }
}

// per-register part of the twist, e^{i pi 512 a/N}, and the radix step itself, for every size
template <int PTS>
__device__ __forceinline__ double2 twist_reg(int a) { return twist16(a * (16 / PTS)); }
template <int PTS, bool INV>
__device__ __forceinline__ void dft_step(double2 (&z)[PTS]) {
  if constexpr (PTS == 16) {
    dft16<INV>(z);
  } else if constexpr (PTS == 8) {
    dft8<INV>(z);
  } else if constexpr (PTS == 4) {
    dft4<INV>(z[0], z[1], z[2], z[3]);
  } else {
    const double2 s = cadd(z[0], z[1]), d = csub(z[0], z[1]);
    z[0] = s; z[1] = d;
  }
}

// The factors of the outputs of a thread's radix step are f[k1] = w v^k1 with w = e^{i pi t/N} (the
// thread's share of the twist) and v = e^{-2 pi i t/(N/2)} (the twiddle of the four-step transform).
// They are formed as f[j + 4m] = (w v^j) (v^4m) from the tabulated w, v, v^2, v^4 and v^8: no factor
// is more than three products away from correctly rounded values, and the transform's rounding
// error stays next to that of a transform with fully tabulated twiddles.
struct StepFactors {
  double2 w, v, v2, v4, v8;
};
template <int PTS>
__device__ __forceinline__ StepFactors load_step_factors(const double2* tables, int t) {
  const double2 one = make_double2(1.0, 0.0);
  return {tables[T_W + t], tables[T_V + t], PTS > 2 ? tables[T_V2 + t] : one, PTS > 4 ? tables[T_V4 + t] : one,
          PTS > 8 ? tables[T_V8 + t] : one};
}
template <bool CONJ, int PTS>
__device__ __forceinline__ void apply_step_factors(double2 (&z)[PTS], const StepFactors& f) {
  double2 b[4];
  b[0] = f.w; b[1] = cmul(f.w, f.v); b[2] = cmul(f.w, f.v2); b[3] = cmul(b[1], f.v2);
  const double2 c[4] = {make_double2(1.0, 0.0), f.v4, f.v8, PTS == 16 ? cmul(f.v4, f.v8) : f.v8};
#pragma unroll
  for (int k1 = 0; k1 < PTS; k1++) {
    const int j = k1 & 3, m = k1 >> 2;
    const double2 g = m == 0 ? b[j] : cmul(b[j], c[m]);
    z[k1] = CONJ ? cmul_conj(z[k1], g) : cmul(z[k1], g);
  }
}

// First half of the forward transform: z[a] = point 512 a + t (untwisted) in; out: the values
// Y[k1] that go to row k1 of the transposition image.
template <int PTS>
__device__ __forceinline__ void forward_step1(double2 (&z)[PTS], const StepFactors& f) {
#ifdef W16_DEBUG_NO_STEP
  return;
#endif
#pragma unroll
  for (int a = 1; a < PTS; a++) z[a] = cmul(z[a], twist_reg<PTS>(a));
  dft_step<PTS, false>(z);
  apply_step_factors<false, PTS>(z, f);
}
// Last half of the inverse transform: the values of image column t in, z[a] = N/2 *
// (coefficient 512 a + t, coefficient 512 a + t + N/2) out (the key carries the 1/(N/2)).
template <int PTS>
__device__ __forceinline__ void inverse_step2(double2 (&z)[PTS], const StepFactors& f) {
#ifdef W16_DEBUG_NO_STEP
  return;
#endif
  apply_step_factors<true, PTS>(z, f);
  dft_step<PTS, true>(z);
#pragma unroll
  for (int a = 1; a < PTS; a++) z[a] = cmul_conj(z[a], twist_reg<PTS>(a));
}

// ---- key conversion: one workgroup per polynomial; spectrum in the order the bootstrap reads it:
// [row = SUBS w + e][register g][lane], frequency row + PTS (64 g + 8 (lane & 7) + (lane >> 3)),
// values scaled by 2^-64 / (N/2) ----
template <typename T, int LOGN>
__global__ __launch_bounds__(THREADS) void bsk_fourier_kernel(const T* __restrict__ bsk, double2* __restrict__ out,
                                                              const double2* __restrict__ tables, size_t polys) {
  using S = typename Torus<T>::Signed;
  using G = Geo<LOGN>;
  constexpr int N = G::N, M = G::M, PTS = G::PTS, SUBS = G::SUBS;
  extern __shared__ __align__(16) char smem[];
  double2* tab = (double2*)smem;
  double2* img = (double2*)(smem + T_LDS * sizeof(double2));
  for (int i = threadIdx.x; i < T_LDS; i += THREADS) tab[i] = tables[i];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const T* poly = bsk + (size_t)blockIdx.x * N;
  const double scale = 0x1.0p-64;
  double2 z[PTS];
#pragma unroll
  for (int a = 0; a < PTS; a++) {
    const int j = 512 * a + t;
    z[a] = make_double2(signed_to_double((S)poly[j]) * scale, signed_to_double((S)poly[j + M]) * scale);
  }
  forward_step1<PTS>(z, load_step_factors<PTS>(tables, t));
#pragma unroll
  for (int k = 0; k < PTS; k++) img[k * THREADS + t] = z[k];
  __syncthreads();
  double2* dst = out + (size_t)blockIdx.x * M + (size_t)wave * (SUBS * 512);
  if (SUBS * wave >= PTS) return;          // a polynomial of the small sizes has fewer rows than the workgroup has waves
#pragma unroll
  for (int e = 0; e < SUBS; e++) {
    double2 x[8];
#pragma unroll
    for (int g = 0; g < 8; g++) x[g] = img[(SUBS * wave + e) * THREADS + 64 * g + lane];
    fft512_plain_forward(x, lane, tab, img + (SUBS * wave + e) * THREADS);
#pragma unroll
    for (int g = 0; g < 8; g++) dst[(e * 8 + g) * 64 + lane] = make_double2(x[g].x * (1.0 / M), x[g].y * (1.0 / M));
  }
}

// position of coefficient j in a polynomial of the slab: j and j + N/2 side by side
template <int LOGN>
__device__ __forceinline__ uint32_t slab_index(uint32_t j) { return ((j & (Geo<LOGN>::M - 1)) << 1) | (j >> Geo<LOGN>::LOGM); }

#ifdef W16_DEBUG_KERNELS
__device__ double2 dump_digits[4 * 8192], dump_sums[8192], dump_back[8192];
__device__ int dump_iteration = 0, dump_column = 0;
#endif

template <typename T, int LOGN, int L, int BL>
__global__ __launch_bounds__(THREADS, 2) void pbs_kernel(PbsLaunch<T> a) {
  static_assert(sizeof(T) == 8 && BL * L <= 33 && BL * L != 31 && BL * L != 32, "64-bit torus whose digits lie in the high dword");
  static_assert(L != 2 || BL <= 15, "the parked upper digit has 16 bits");
  using G = Geo<LOGN>;
  constexpr int N = G::N, M = G::M, PTS = G::PTS, SUBS = G::SUBS, CPW = G::CPW;
  constexpr int R = 2;                    // k + 1
  // The barrier that frees the image after the last column's read replaces the one in front of the next
  // CMUX's first image write (1 to 3.5 % faster at every size once the slabs were padded and the upper
  // digits parked; before that the N = 16384 instances lost 1 % to it).
  constexpr bool FREE_EARLY = true;
  extern __shared__ __align__(16) char smem[];
  double2* tab = (double2*)smem;
  double2* img = (double2*)(smem + T_LDS * sizeof(double2));
  for (int i = threadIdx.x; i < T_LDS; i += THREADS) tab[i] = a.tables[i];
  __syncthreads();

  const int t0 = threadIdx.x;
  uint32_t* acc = (uint32_t*)((char*)a.scratch + (size_t)blockIdx.x * group_stride_bytes<LOGN, L>());   // [ciphertext][R][N] paired layout
  constexpr bool PARK = L == 2 && PTS % 4 == 0;
  // [ciphertext][PTS / 4][thread]: this thread's points 4g .. 4g+3
  uint4* parked = parked_in_lds<LOGN>() ? (uint4*)(smem + T_LDS * sizeof(double2) + images_bytes<LOGN>())
                                        : (uint4*)((char*)a.scratch + (size_t)gridDim.x * group_stride_bytes<LOGN, L>() +
                                                   (size_t)blockIdx.x * parked_bytes<LOGN>());   // behind all accumulators
  const __amdgpu_buffer_rsrc_t key_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.fbsk, 0, (int)0x7FFFFFFF, 0x00020000);   // offsets are formed per iteration: the base moves with it
  const uint32_t n = a.n;

  for (size_t group = blockIdx.x; group * CPW < a.count; group += gridDim.x) {
    // the ciphertexts of this round; a slot past the end of the batch repeats the last ciphertext (its
    // result is not written)
    const T* in[CPW];
#pragma unroll
    for (int b = 0; b < CPW; b++) {
      const size_t ct = group * CPW + b < a.count ? group * CPW + b : a.count - 1;
      in[b] = a.ct_in + ct * a.in_stride;
      const uint32_t lut_id = checked_lut_id(a, ct);
      const T* lut = a.luts + (size_t)lut_id * (R * N);
      // acc <- lut * X^{-b_hat}, high dwords
      const uint32_t b_hat = modswitch<T>(in[b][n], LOGN);
      const uint32_t d0 = (2 * N - b_hat) & (2 * N - 1);
      for (int idx = t0; idx < R * N; idx += THREADS) {
        const int c = idx >> LOGN, j = idx & (N - 1);
        acc[b * R * N + c * N + slab_index<LOGN>(j)] = (uint32_t)(rotated_coeff<T>(lut + (size_t)c * N, j, d0, N) >> 32);
      }
    }
    __syncthreads();

    for (uint32_t i = 0; i < n; i++) {
      // a mask element that is zero rotates by nothing: its digits are all zero and nothing is added;
      // the CMUX is skipped when that holds for every ciphertext of the workgroup (the reference skips it too)
      uint32_t a_hat[CPW];
      bool any = false;
#pragma unroll
      for (int b = 0; b < CPW; b++) {
        const T a_i = wave_uniform(in[b][i]);
        any |= a_i != 0;
        a_hat[b] = a_i == 0 ? 0u : (modswitch<T>(a_i, LOGN) & (2 * N - 1));
      }
      if (!any) continue;
      // key of this iteration: [level][row][column][8192] complex

      double2 S[R][SUBS][8];   // Fourier sums: [column][sub-transform][register]
#pragma unroll
      for (int c = 0; c < R; c++)
#pragma unroll
        for (int e = 0; e < SUBS; e++)
#pragma unroll
          for (int g = 0; g < 8; g++) S[c][e][g] = make_double2(0.0, 0.0);

#pragma unroll 1
      for (int r = 0; r < R; r++) {
        // The rotated difference of row r is gathered once per level (16 points = 32 coefficients, from the
        // L1/L2-resident slab) instead of keeping decomposition states alive across a transform: the sums
        // already take half of the register file. Balanced digits, lowest level first.
#pragma unroll L == 3 ? 1 : L          // (three levels: one copy of the body, or the registers run out)
        for (int l = L; l >= 1; l--) {
          // the thread index is made opaque per transform: indices that do not change from one gather to the
          // next are recomputed then, where hoisting them would cost registers that are not there
          int t = t0;
          asm volatile("" : "+v"(t));
          const int lane = t & 63, wave = t >> 6;
          // this wave's rows of a key polynomial (the waves of different ciphertexts read the same rows)
          const uint32_t key_lane = (uint32_t)lane * 16u + (uint32_t)(CPW == 1 ? wave : wave % PTS) * (uint32_t)(SUBS * 8192);
          double2 z[CPW][PTS];
          if (PARK && l == 1) {
            // the upper level's digits were parked by this thread when it formed the lower level's
#pragma unroll
            for (int b = 0; b < CPW; b++)
#pragma unroll
              for (int g = 0; g < PTS / 4; g++) {
                const uint4 w = (b * (PTS / 4) + g < parked_lds_groups<LOGN>())
                                    ? ((const uint4*)(smem + T_LDS * sizeof(double2) + images_bytes<LOGN>()))[(b * (PTS / 4) + g) * THREADS + t]
                                    : parked[(b * (PTS / 4) + g) * THREADS + t];
                const uint32_t pk[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int h = 0; h < 4; h++)
                  z[b][4 * g + h] = make_double2((double)(int32_t)__builtin_amdgcn_sbfe((int32_t)pk[h], 0, 16),
                                                 (double)((int32_t)pk[h] >> 16));
              }
          } else {
#pragma unroll
          for (int b = 0; b < CPW; b++) {
          const uint32_t* p = acc + b * R * N + r * N;
          uint32_t pk[PARK ? PTS : 1];
#pragma unroll
          for (int q = 0; q < PTS; q++) {
            const uint32_t j0 = 512 * q + t;                                   // and j0 + M
            const uint32_t s0 = (j0 - a_hat[b]) & (2 * N - 1);                     // rotated index of j0; that of j0 + M is s0 + M
#ifdef W16_TIMING_NO_GATHER                                                     // timing experiments only: wrong results
            const uint2 own = make_uint2(j0 * 2654435761u, s0 * 40503u), pr = make_uint2(s0 * 2246822519u, j0 * 3266489917u);
#else
            const uint2 own = *(const uint2*)(p + 2 * j0);
            const uint2 pr = *(const uint2*)(p + 2 * (s0 & (M - 1)));
#endif
            const bool hi_half = (s0 & M) != 0;
            uint32_t r0 = hi_half ? pr.y : pr.x, r1 = hi_half ? pr.x : pr.y;   // coefficients s0 mod N and (s0 + M) mod N
            const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int32_t)s0, LOGN, 1);
            const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int32_t)(s0 + M), LOGN, 1);
            // (+-rot - own) rounded to the top BL L bits, as a signed value. The difference is known to 2^-32
            // only, so a tie is one case in 2^(32 - BL L), not one in 2^32 as with 64-bit words: ties go to
            // the even neighbour (and the negation is exact) so that the rounding has no mean, which a
            // binary key would add up coherently over the N/2 coefficients it selects.
            // When the levels ask for more bits than the 32 the accumulator keeps (three levels of 11), the
            // difference is decomposed exactly and the lowest level's digit has SHORT bits fewer: its
            // values are multiples of 2^SHORT.
            constexpr int SHORT = BL * L > 32 ? BL * L - 32 : 0;
            constexpr int DROP = BL * L > 32 ? 0 : 32 - BL * L;
            const uint32_t u0 = ((r0 ^ m0) - m0) - own.x, u1 = ((r1 ^ m1) - m1) - own.y;
            int32_t v0 = (int32_t)u0, v1 = (int32_t)u1;
            if constexpr (DROP > 0) {
              constexpr uint32_t BELOW_HALF = (1u << (DROP > 0 ? DROP - 1 : 0)) - 1u;
              v0 = (int32_t)(u0 + BELOW_HALF + __builtin_amdgcn_ubfe(u0, DROP, 1)) >> DROP;
              v1 = (int32_t)(u1 + BELOW_HALF + __builtin_amdgcn_ubfe(u1, DROP, 1)) >> DROP;
            }
#pragma unroll
            for (int m = L; m > l; m--) {                                        // strip the digits below level l
              const int width = m == L ? BL - SHORT : BL;
              v0 = (v0 - (int32_t)__builtin_amdgcn_sbfe(v0, 0, width)) >> width;
              v1 = (v1 - (int32_t)__builtin_amdgcn_sbfe(v1, 0, width)) >> width;
            }
            if (l > 1) {
              const int width = l == L ? BL - SHORT : BL, scale = l == L ? SHORT : 0;
              const int32_t d0 = (int32_t)__builtin_amdgcn_sbfe(v0, 0, width), d1 = (int32_t)__builtin_amdgcn_sbfe(v1, 0, width);
              if constexpr (PARK)    // (v - d) >> BL lies in [-2^(BL-1), 2^(BL-1)]: 16 bits hold it for BL = 15
                pk[q] = ((uint32_t)((v0 - d0) >> BL) & 0xFFFFu) | ((uint32_t)((v1 - d1) >> BL) << 16);
              v0 = d0 << scale;
              v1 = d1 << scale;
            }
            z[b][q] = make_double2((double)v0, (double)v1);
            if ((q & 3) == 3) {
              if constexpr (PARK)
                if (b * (PTS / 4) + (q >> 2) < parked_lds_groups<LOGN>())
                  ((uint4*)(smem + T_LDS * sizeof(double2) + images_bytes<LOGN>()))[(b * (PTS / 4) + (q >> 2)) * THREADS + t] = make_uint4(pk[q - 3], pk[q - 2], pk[q - 1], pk[q]);
                else
                  parked[(b * (PTS / 4) + (q >> 2)) * THREADS + t] = make_uint4(pk[q - 3], pk[q - 2], pk[q - 1], pk[q]);
              __builtin_amdgcn_sched_barrier(0);   // four points' loads in flight at a time: registers are short
            }
          }
          }
          }
#ifdef W16_DEBUG_KERNELS
          if (L == 2 && CPW == 1 && blockIdx.x == 0 && (int)i == dump_iteration)
            for (int q = 0; q < PTS; q++) dump_digits[((r * 2 + (l - 1)) * M + 512 * q + t) & (4 * 8192 - 1)] = z[0][q];
#endif
          __builtin_amdgcn_sched_barrier(0);
          // (the factors are loaded where they are used, per ciphertext: held in a variable across the
          // loop they are fetched early and cost the registers of the largest instance)
#pragma unroll
          for (int b = 0; b < CPW; b++) forward_step1<PTS>(z[b], load_step_factors<PTS>(a.tables, t));
          __builtin_amdgcn_sched_barrier(0);
          // everyone is done with the image's previous content (with FREE_EARLY, the first transform of a
          // CMUX has that from the barrier behind the last read of the inverse part)
          // (with two images the previous content is two barriers old: nothing to wait for)
          constexpr bool TWO = images<LOGN>() == 2;
          double2* const im = img + (TWO ? ((r * L + (L - l)) & 1) * (G::ROWS * THREADS) : 0);
          if (!TWO && (!FREE_EARLY || r != 0 || l != L)) __syncthreads();
#pragma unroll
          for (int b = 0; b < CPW; b++)
#pragma unroll
            for (int k = 0; k < PTS; k++) im[(b * PTS + k) * THREADS + t] = z[b][k];
          __syncthreads();
          // the two sub-transforms of this wave, one after the other, each followed by its products
          const uint32_t key_base = (uint32_t)(((size_t)i * L + (l - 1)) * R + r) * (uint32_t)(R * M * sizeof(double2));
#pragma unroll
          for (int e = 0; e < SUBS; e++) {
            double2 x[8];
#pragma unroll
            for (int g = 0; g < 8; g++) x[g] = im[(SUBS * wave + e) * THREADS + 64 * g + lane];
            fft512_plain_forward_head(x, lane, tab);
            __builtin_amdgcn_sched_barrier(0);
            // the key values of the first column are requested in front of the last exchange and pass, those of the second
            // one behind it: their way from the L2 overlaps with that pass and with the first column's products
            double2 kq[R][8];
            auto request = [&](int c) {
#pragma unroll
              for (int g = 0; g < 8; g++) {
                const auto w = __builtin_amdgcn_raw_buffer_load_b128(
                    key_rsrc, (int)(key_lane + (g & 3) * 1024),
                    (int)(key_base + (uint32_t)c * (uint32_t)(M * sizeof(double2)) + e * 8192 + (g >> 2) * 4096), W16_KEY_AUX);
                kq[c][g] = as_double2(__builtin_bit_cast(v4u, w));
#ifdef W16_TIMING_NO_KEY
                kq[c][g] = make_double2(1e-3 * (double)(g + 1), 2e-3);
#endif
              }
            };
            request(0);
            if constexpr (PTS <= 8) request(1);     // (the smaller sizes have the registers for both)
            __builtin_amdgcn_sched_barrier(0);
            fft512_plain_forward_tail(x, lane, tab, im + (SUBS * wave + e) * THREADS);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PTS > 8) request(1);
#pragma unroll
            for (int c = 0; c < R; c++)
#pragma unroll
              for (int g = 0; g < 8; g++) S[c][e][g] = cfma(kq[c][g], x[g], S[c][e][g]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }

      // back: per column, the two inverse sub-transforms of this wave, the transposition, the radix-16 step, the update
#pragma unroll                        // a run-time column index would put the sums in scratch memory
      for (int c = 0; c < R; c++) {
        int t = t0;
        asm volatile("" : "+v"(t));
        const int lane = t & 63, wave = t >> 6;
        constexpr bool TWO = images<LOGN>() == 2;
        double2* const im = img + (TWO ? (c & 1) * (G::ROWS * THREADS) : 0);   // (the forward part took an even number of images)
        // No barrier ahead of the sub-transforms: the first column follows this wave's own forward
        // sub-transforms, which touched its own rows only, and the second one the barrier below.
#ifdef W16_DEBUG_KERNELS
        if (blockIdx.x == 0 && (int)i == dump_iteration && c == dump_column)
          for (int e = 0; e < SUBS; e++)
            for (int g = 0; g < 8; g++) dump_sums[(wave * 1024 + (e * 8 + g) * 64 + lane) & 8191] = S[c][e][g];
#endif
#pragma unroll
        for (int e = 0; e < SUBS; e++) {
          fft512_plain_inverse(S[c][e], lane, tab, im + (SUBS * wave + e) * THREADS);
#pragma unroll
          for (int g = 0; g < 8; g++) im[(SUBS * wave + e) * THREADS + 64 * g + lane] = S[c][e][g];
        }
        __syncthreads();
        double2 z[CPW][PTS];
#pragma unroll
        for (int b = 0; b < CPW; b++)
#pragma unroll
          for (int k = 0; k < PTS; k++) z[b][k] = im[(b * PTS + k) * THREADS + t];
        // The image is free for its next writer (the next column's sub-transforms, or the first forward
        // transform of the next CMUX) as soon as everyone has read it: with the barrier here, where the
        // waves still run together, a wave goes from its update straight into compute of its own while
        // its neighbour on the SIMD waits for memory.
        if (!TWO && (FREE_EARLY || c + 1 < R)) __syncthreads();
#pragma unroll
        for (int b = 0; b < CPW; b++) inverse_step2<PTS>(z[b], load_step_factors<PTS>(a.tables, t));
#ifdef W16_DEBUG_KERNELS
        if (CPW == 1 && blockIdx.x == 0 && (int)i == dump_iteration && c == dump_column)
          for (int q = 0; q < PTS; q++) dump_back[(512 * q + t) & 8191] = z[0][q];
#endif
#pragma unroll
        for (int b = 0; b < CPW; b++) {
          uint32_t* pc = acc + b * R * N + c * N;
#pragma unroll
          for (int q = 0; q < PTS; q++) {
            const uint32_t j0 = 512 * q + t;
#ifdef W16_TIMING_NO_UPDATE_LOAD
            uint2 old = make_uint2(j0, j0 * 7u);
#else
            uint2 old = *(const uint2*)(pc + 2 * j0);
#endif
            old.x = torus_accumulate(old.x, z[b][q].x);
            old.y = torus_accumulate(old.y, z[b][q].y);
            *(uint2*)(pc + 2 * j0) = old;
          }
        }
      }
      // No barrier here: the next gather reads row 0, whose update was complete before the barriers of
      // the second column; row 1 is read after the barriers of row 0's transforms.
    }
    __syncthreads();                     // the extraction reads what other threads just wrote

    // sample extraction of coefficient 0; the low dwords are zero
#pragma unroll
    for (int b = 0; b < CPW; b++) {
      if (group * CPW + b >= a.count) break;
      const uint32_t* pb = acc + b * R * N;
      T* out = a.ct_out + (group * CPW + b) * a.out_stride;
      for (int j = t0; j < N; j += THREADS) {
        const uint32_t w = (j == 0) ? pb[slab_index<LOGN>(0)] : (uint32_t)(0u - pb[slab_index<LOGN>(N - j)]);
        out[j] = (T)w << 32;
      }
      if (t0 == 0) out[N] = (T)pb[N + slab_index<LOGN>(0)] << 32;
    }
    __syncthreads();                     // the slab is rewritten by the next ciphertext
  }
}

#ifdef W16_DEBUG_KERNELS
// transform pieces on their own, for checks against a host transform (debug builds only)
__global__ __launch_bounds__(THREADS) void debug_kernel(int mode, const double2* in, double2* out, const double2* tables) {
  constexpr int PTS = 16;
  extern __shared__ __align__(16) char smem[];
  double2* tab = (double2*)smem;
  double2* img = (double2*)(smem + T_LDS * sizeof(double2));
  for (int i = threadIdx.x; i < T_LDS; i += THREADS) tab[i] = tables[i];
  __syncthreads();
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (mode == 1 || mode == 2) {
    double2 x[8];
#pragma unroll
    for (int g = 0; g < 8; g++) x[g] = in[wave * 512 + 64 * g + lane];
    if (mode == 1) fft512_plain_forward(x, lane, tab, img + wave * THREADS); else fft512_plain_inverse(x, lane, tab, img + wave * THREADS);
#pragma unroll
    for (int g = 0; g < 8; g++) out[wave * 512 + 64 * g + lane] = x[g];
  } else if (mode == 3) {
    double2 z[16];
#pragma unroll
    for (int a = 0; a < 16; a++) z[a] = in[512 * a + t];
    forward_step1<PTS>(z, load_step_factors<PTS>(tables, t));
#pragma unroll
    for (int k = 0; k < 16; k++) img[k * THREADS + t] = z[k];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 2; e++) {
      double2 x[8];
#pragma unroll
      for (int g = 0; g < 8; g++) x[g] = img[(2 * wave + e) * THREADS + 64 * g + lane];
      fft512_plain_forward(x, lane, tab, img + (2 * wave + e) * THREADS);
#pragma unroll
      for (int g = 0; g < 8; g++) out[wave * 1024 + (e * 8 + g) * 64 + lane] = x[g];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 2; e++) {
      double2 x[8];
#pragma unroll
      for (int g = 0; g < 8; g++) x[g] = in[wave * 1024 + (e * 8 + g) * 64 + lane];
      fft512_plain_inverse(x, lane, tab, img + (2 * wave + e) * THREADS);
#pragma unroll
      for (int g = 0; g < 8; g++) img[(2 * wave + e) * THREADS + 64 * g + lane] = x[g];
    }
    __syncthreads();
    double2 z[16];
#pragma unroll
    for (int k = 0; k < 16; k++) z[k] = img[k * THREADS + t];
    inverse_step2<PTS>(z, load_step_factors<PTS>(tables, t));
#pragma unroll
    for (int a = 0; a < 16; a++) out[512 * a + t] = z[a];
  }
}
#endif

}  // namespace w16

}  // namespace

bool wave16k_supported(const camd_params& p) {
  if (p.torus_bits != 64 || p.glwe_dimension != 1) return false;
  const uint32_t N = p.polynomial_size;
  if (p.pbs_level == 2 && p.pbs_base_log == 15) return N == 16384 || N == 8192 || N == 4096;
  if (p.pbs_level == 3 && p.pbs_base_log == 11) return N == 16384;
  return (N == 8192 || N == 4096 || N == 2048) && p.pbs_level == 1 && p.pbs_base_log == 23;
}
size_t wave16k_tables_count() { return (size_t)w16::T_TOTAL; }
size_t wave16k_scratch_bytes(uint32_t N, unsigned* grid_out, size_t count, int cus) {
  const size_t per_group = N >= 8192 ? 1 : 8192 / N;      // ciphertexts per workgroup (Geo::CPW)
  const unsigned grid = (unsigned)std::min<size_t>((count + per_group - 1) / per_group, (size_t)(cus > 0 ? cus : 1));
  if (grid_out) *grid_out = grid;
  const size_t rows = N >= 16384 ? 16 : 8;                  // image rows of a workgroup (Geo::ROWS)
  // per workgroup: accumulators at the padded stride (w16::group_stride_bytes), then the parked digits of all
  return (size_t)grid * (per_group * 2 * N * sizeof(uint32_t) + w16::SLAB_PAD + rows * 512 * sizeof(uint32_t));
}

void wave16k_fill_tables(uint32_t N, double2* t) {
  // folded constants of a twiddled radix-8 pass (dft8_tw in wave1024.hpp): th[q] is the angle of input q
  auto fold = [&](double2* dst, int stride, const double (&th)[8], bool own_w0) {
    double c[8], tn[8];
    for (int q = 0; q < 8; q++) {
      c[q] = std::cos(th[q]);
      const double sn = std::sin(th[q]);
      if (std::fabs(c[q]) < 1e-9) c[q] = 0x1.0p-60;   // a quarter turn: finite tangent, value off by 2^-60
      tn[q] = sn / c[q];
    }
    dst[0 * stride] = make_double2(tn[1], tn[2]);
    dst[1 * stride] = make_double2(tn[3], tn[4]);
    dst[2 * stride] = make_double2(tn[5], tn[6]);
    dst[3 * stride] = make_double2(tn[7], c[4]);
    dst[4 * stride] = make_double2(c[5] / c[1], c[6] / c[2]);
    dst[5 * stride] = make_double2(c[7] / c[3], c[2]);
    dst[6 * stride] = make_double2(c[3] / c[1], c[1]);
    dst[7 * stride] = make_double2(c[1] * HALF_SQRT2, 0.0);
    if (own_w0) dst[8 * stride] = make_double2(std::cos(th[0]), std::sin(th[0]));
  };
  for (int lane = 0; lane < 64; lane++) {
    const int hi = lane >> 3, lo = lane & 7;
    double th[8];
    for (int q = 0; q < 8; q++) th[q] = -2.0 * M_PI * (double)((8 * q + lo) * hi) / 512.0;
    fold(t + w16::T_A + lane, 64, th, true);
    for (int b = 0; b < 8; b++) th[b] = -2.0 * M_PI * (double)(lane * b) / 512.0;
    fold(t + w16::T_C + lane, 64, th, true);
  }
  for (int lo = 0; lo < 8; lo++) {
    double th[8];
    for (int q = 0; q < 8; q++) th[q] = -2.0 * M_PI * (double)(lo * q) / 64.0;
    fold(t + w16::T_B + lo, 8, th, false);
  }
  for (int i = 0; i < w16::THREADS; i++) {
    const double aw = M_PI * (double)i / (double)N;
    t[w16::T_W + i] = make_double2(std::cos(aw), std::sin(aw));
    const int power_tables[4] = {w16::T_V, w16::T_V2, w16::T_V4, w16::T_V8};
    for (int s = 0; s < 4; s++) {
      const double av = -2.0 * M_PI * (double)(i << s) / (double)(N / 2);
      t[power_tables[s] + i] = make_double2(std::cos(av), std::sin(av));
    }
  }
}

template <typename T>
int wave16k_bsk_to_fourier(const T* bsk_std, double2* fbsk, const double2* tables, uint32_t N, size_t polys, hipStream_t stream) {
  auto run = [&](auto kern, size_t lds) -> int {
    CAMD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)polys), dim3(w16::THREADS), lds, stream, bsk_std, fbsk, tables, polys);
    return CAMD_OK;
  };
  int rc = CAMD_EINVAL;
  if (N == 16384) rc = run(w16::bsk_fourier_kernel<T, 14>, w16::lds_bytes<14>());
  if (N == 8192) rc = run(w16::bsk_fourier_kernel<T, 13>, w16::lds_bytes<13>());
  if (N == 4096) rc = run(w16::bsk_fourier_kernel<T, 12>, w16::lds_bytes<12>());
  if (N == 2048) rc = run(w16::bsk_fourier_kernel<T, 11>, w16::lds_bytes<11>());
  if (rc != CAMD_OK) return rc;
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

namespace {
template <typename T, int LOGN, int L, int BL>
int launch_instance16k(const PbsLaunch<T>& a, unsigned grid, hipStream_t stream) {
  static_assert(w16::lds_bytes<LOGN>() <= 160 * 1024, "LDS budget exceeded");
  auto kern = w16::pbs_kernel<T, LOGN, L, BL>;
  CAMD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w16::lds_bytes<LOGN>()));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(w16::THREADS), w16::lds_bytes<LOGN>(), stream, a);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}
}  // namespace

template <typename T>
int wave16k_launch_pbs(const PbsLaunch<T>& a, hipStream_t stream) {
  if constexpr (sizeof(T) == 8) {
    if (a.count == 0) return CAMD_OK;
    if (!a.scratch) return fail(CAMD_EINVAL, "wave kernel (N >= 2048): the accumulator slabs were not reserved");
    int dev = 0, cus = 0;
    CAMD_HIP(hipGetDevice(&dev));
    CAMD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    unsigned grid = 1;
    wave16k_scratch_bytes(a.N, &grid, a.count, cus);
    if (a.k == 1 && a.N == 16384 && a.level == 2 && a.base_log == 15) return launch_instance16k<T, 14, 2, 15>(a, grid, stream);
    if (a.k == 1 && a.N == 16384 && a.level == 3 && a.base_log == 11) return launch_instance16k<T, 14, 3, 11>(a, grid, stream);
    if (a.k == 1 && a.N == 8192 && a.level == 2 && a.base_log == 15) return launch_instance16k<T, 13, 2, 15>(a, grid, stream);
    if (a.k == 1 && a.N == 8192 && a.level == 1 && a.base_log == 23) return launch_instance16k<T, 13, 1, 23>(a, grid, stream);
    if (a.k == 1 && a.N == 4096 && a.level == 2 && a.base_log == 15) return launch_instance16k<T, 12, 2, 15>(a, grid, stream);
    if (a.k == 1 && a.N == 4096 && a.level == 1 && a.base_log == 23) return launch_instance16k<T, 12, 1, 23>(a, grid, stream);
    if (a.k == 1 && a.N == 2048 && a.level == 1 && a.base_log == 23) return launch_instance16k<T, 11, 1, 23>(a, grid, stream);
  }
  return fail(CAMD_EINVAL, "wave kernel (N >= 2048): no instance for this parameter set");
}

template int wave16k_bsk_to_fourier<uint64_t>(const uint64_t*, double2*, const double2*, uint32_t, size_t, hipStream_t);
template int wave16k_launch_pbs<uint64_t>(const PbsLaunch<uint64_t>&, hipStream_t);
template int wave16k_launch_pbs<uint32_t>(const PbsLaunch<uint32_t>&, hipStream_t);

#ifdef W16_DEBUG_KERNELS
extern "C" int camd_debug_w16_dump(int which, double* out_host) {
  const size_t bytes = (size_t)8192 * sizeof(double2);
  CAMD_HIP(hipDeviceSynchronize());
  if (which >= 100) {
    const int col = which - 100;
    CAMD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(w16::dump_column), &col, sizeof(int)));
    return CAMD_OK;
  }
  if (which >= 10) {
    const int it = which - 10;
    CAMD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(w16::dump_iteration), &it, sizeof(int)));
    return CAMD_OK;
  }
  if (which == 0) CAMD_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(w16::dump_digits), 4 * bytes));
  if (which == 1) CAMD_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(w16::dump_sums), bytes));
  if (which == 2) CAMD_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(w16::dump_back), bytes));
  return CAMD_OK;
}
extern "C" int camd_debug_w16(int mode, const double* in_host, double* out_host) {
  std::vector<double2> tables(w16::T_TOTAL);
  wave16k_fill_tables(16384, tables.data());
  double2 *d_in = nullptr, *d_out = nullptr, *d_tab = nullptr;
  const size_t bytes = (size_t)8192 * sizeof(double2);
  CAMD_HIP(hipMalloc(&d_in, bytes));
  CAMD_HIP(hipMalloc(&d_out, bytes));
  CAMD_HIP(hipMalloc(&d_tab, tables.size() * sizeof(double2)));
  CAMD_HIP(hipMemcpy(d_in, in_host, bytes, hipMemcpyHostToDevice));
  CAMD_HIP(hipMemcpy(d_tab, tables.data(), tables.size() * sizeof(double2), hipMemcpyHostToDevice));
  auto kern = w16::debug_kernel;
  CAMD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w16::lds_bytes<14>()));
  hipLaunchKernelGGL(kern, dim3(1), dim3(w16::THREADS), w16::lds_bytes<14>(), 0, mode, d_in, d_out, d_tab);
  CAMD_HIP(hipGetLastError());
  CAMD_HIP(hipDeviceSynchronize());
  CAMD_HIP(hipMemcpy(out_host, d_out, bytes, hipMemcpyDeviceToHost));
  hipFree(d_in); hipFree(d_out); hipFree(d_tab);
  return CAMD_OK;
}
#endif

}  // namespace camd
