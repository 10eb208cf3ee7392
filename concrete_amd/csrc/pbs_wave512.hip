// Programmable bootstrap, wave-per-ciphertext family, N = 512: the boolean default parameters
// (32-bit torus, k = 2, two levels, base 2^8) and the three small shortint sets (64-bit torus, k = 3,
// one level, base 2^18). Replaces FftEngine::discard_bootstrap_lwe_ciphertext as the reference calls
// it at concrete-boolean/src/engine/bootstrapping/cpu.rs:231-236 and
// concrete-shortint/src/engine/server_side/mod.rs:148-153.
//
// N = 512 instance (the boolean default parameters: 32-bit torus, k = 2, two levels, base 2^8).
// Same schedule as wave_pbs_kernel_v2 — one wave per ciphertext, the transforms of a CMUX
// interleaved as independent streams, the key rows of the next CMUX parked in accumulation
// registers — with four points per lane: a 256-point transform is four radix-4 passes with three
// exchanges, each a 4 x 4 transpose between the register index and one base-4 digit of the lane
// index. With n = 64 a + lane, lane = 16 l2 + 4 l1 + l0 and f = k1 + 4 k2 + 16 k3 + 64 k4:
//   pass 1 over a  (twiddle W_256^(lane k1)),      exchange register <-> l2,
//   pass 2 over l2 (twiddle W_64^((lane & 15) k2)), exchange register <-> l1,
//   pass 3 over l1 (twiddle W_16^((lane & 3) k3)),  exchange register <-> l0,
//   pass 4 over l0; register k4 of lane 16 k1 + 4 k2 + k3 then holds X[f].
// The inverse runs the mirror image (conjugate twiddles, same exchanges), unnormalised; the key
// is pre-multiplied by 1/256. The exchange image is 4 rows of 64 values per plane, element
// (r, lane) at r * 64 + (lane ^ (r << 2 * digit)): conflict-free for the lane groups of
// ds_write_b64 and ds_read_b64 on both sides of every exchange (checked by enumeration).
#include "lane_swap.hpp"
#include "wave_common.hpp"

namespace camd {

namespace {

namespace w4 {
constexpr int N = 512, M = 256, LOGN = 9;
constexpr int TAB_TWIST = 0;                 // [j] e^{+i pi j / N}, j < 256
constexpr int TAB_A = 256;                   // [k-1][lane] W_256^(lane k), k = 1..3
constexpr int TAB_B = TAB_A + 3 * 64;        // [k-1][m]    W_64^(m k),  m < 16
constexpr int TAB_C = TAB_B + 3 * 16;        // [k-1][m]    W_16^(m k),  m < 4
constexpr int TAB_COUNT = TAB_C + 3 * 4;      // what the key conversion stages in LDS
// The bootstrap kernels apply every twiddle on the receiving side of the exchange that follows it,
// as the input factor of the next radix-4 pass, folded into the butterflies (dft4_tw). Written
// that way the factor of input j of a pass is e^{i j alpha} with one angle alpha per lane (the part
// that is common to the four inputs commutes with the pass and joins the next one), so input 0 is
// never multiplied; the per-lane part of the twist rides along (theta(k) = pi (1 - 4k)/N below).
// Only the last inverse pass keeps a factor on input 0. Constants per pass and lane, as pairs of
// doubles: (t1,t2) (t3,c2) (r31,c1) and, for that last pass, w0. With lane = 16 d2 + 4 d1 + d0:
constexpr int TWS_F1 = 0;                     // [3][d2]       forward pass 2: alpha = 16 theta(d2)
constexpr int TWS_F2 = TWS_F1 + 3 * 4;        // [3][lane>>2]  forward pass 3: alpha = 4 theta(d2) - pi d1/8
constexpr int TWS_F3 = TWS_F2 + 3 * 16;       // [3][lane]     forward pass 4: alpha = theta(d2) - 2 pi (4 d0 + d1)/64
constexpr int TWS_I1 = TWS_F3 + 3 * 64;       // [3][d0]       inverse pass 2: input k carries conj W_16^(d0 k)
constexpr int TWS_I2 = TWS_I1 + 3 * 4;        // [3][lane&15]  inverse pass 3: conj W_64^((lane & 15) k)
constexpr int TWS_I3 = TWS_I2 + 3 * 16;       // [4][lane]     inverse pass 4: conj e^{i theta(k) lane}, k = 0..3
constexpr int TWS_COUNT = TWS_I3 + 4 * 64;    // the bootstrap kernels' LDS image
constexpr int TAB_TWS = TAB_COUNT;            // where it starts in the table buffer
constexpr int TAB_TOTAL = TAB_TWS + TWS_COUNT;
// the per-register part of the twist e^{i pi (64a + lane)/N} = e^{i pi a/8} e^{i pi lane/N}, a = 0..3
constexpr double TWIST4_RE[4] = {1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173};
constexpr double TWIST4_IM[4] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613};
constexpr int PLANE = 4 * 64;
// per wave: the accumulator ((k+1) N words of `word` bytes) and the two exchange planes
template <int K>
constexpr size_t slab_bytes(size_t word) { return (size_t)(K + 1) * N * word + 2 * PLANE * sizeof(double); }
template <int K>
constexpr size_t lds_bytes(size_t word, int waves = WAVES) {
  return TWS_COUNT * sizeof(double2) + waves * slab_bytes<K>(word) + waves * sizeof(uint32_t);   // + progress words
}
constexpr int STREAM_WAVES = 8;   // waves per workgroup of the streamed form: two per SIMD, partners w and w ^ 4
// The streamed form of a 64-bit single-level instance keeps the high dwords of the accumulator only
// (the digits come from them; every CMUX rounds its increment at 2^-32 of the torus): half the LDS,
// 4-byte gathers at 4-byte stride, and two workgroups fit a CU.
template <typename T, int L, bool STREAM>
struct AccWord { typedef T type; };
template <>
struct AccWord<uint64_t, 1, true> { typedef uint32_t type; };

template <bool INV>
__device__ __forceinline__ void dft4(double2 (&x)[4]) {
  const double2 s0 = cadd(x[0], x[2]), d0 = csub(x[0], x[2]);
  const double2 s1 = cadd(x[1], x[3]), d1 = csub(x[1], x[3]);
  const double2 t = INV ? mul_pos_i(d1) : mul_neg_i(d1);
  x[0] = cadd(s0, s1);
  x[2] = csub(s0, s1);
  x[1] = cadd(d0, t);
  x[3] = csub(d0, t);
}

// transpose between the register index and base-4 digit DIGIT of the lane index, all streams
template <int S, int DIGIT>
__device__ __forceinline__ void exchange4(double2 (&x)[S][4], int lane, double* sre, double* sim) {
#ifndef W4_REGISTER_DIGITS
#define W4_REGISTER_DIGITS 2   // exchanges of lane digits >= this one run in registers
#endif
  if constexpr (DIGIT >= W4_REGISTER_DIGITS) {
    // register bit 1 <-> lane bit 2 DIGIT + 1, register bit 0 <-> lane bit 2 DIGIT: for the upper two lane
    // bits one v_permlane32_swap / v_permlane16_swap per register pair and dword, for bits 3:2 two row
    // shifts; the LDS store path, which these kernels run out of first, is left to the other exchanges
#pragma unroll
    for (int s = 0; s < S; s++) {
      swap_complex<2 * DIGIT + 1>(x[s][0], x[s][2]);
      swap_complex<2 * DIGIT + 1>(x[s][1], x[s][3]);
      swap_complex<2 * DIGIT>(x[s][0], x[s][1]);
      swap_complex<2 * DIGIT>(x[s][2], x[s][3]);
    }
    return;
  }
  constexpr int SH = 2 * DIGIT;
  const int d = (lane >> SH) & 3;
  const int rest = lane & ~(3 << SH);
#pragma unroll
  for (int s = 0; s < S; s++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int pos = r * 64 + (lane ^ (r << SH));
      sre[pos] = x[s][r].x; sim[pos] = x[s][r].y;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int idx = d * 64 + (rest | ((j ^ d) << SH));
      x[s][j] = make_double2(sre[idx], sim[idx]);
    }
  }
}

// plain forward transform (twist applied by the caller): the key conversion
template <int S>
__device__ __forceinline__ void fft256_forward_multi(double2 (&x)[S][4], int lane, const double2* tab, double* sre,
                                                     double* sim) {
  double2 w[3];
#pragma unroll
  for (int k = 1; k < 4; k++) w[k - 1] = tab[TAB_A + (k - 1) * 64 + lane];
#pragma unroll
  for (int s = 0; s < S; s++) {
    dft4<false>(x[s]);
#pragma unroll
    for (int k = 1; k < 4; k++) x[s][k] = cmul(x[s][k], w[k - 1]);
  }
  exchange4<S, 2>(x, lane, sre, sim);
#pragma unroll
  for (int k = 1; k < 4; k++) w[k - 1] = tab[TAB_B + (k - 1) * 16 + (lane & 15)];
#pragma unroll
  for (int s = 0; s < S; s++) {
    dft4<false>(x[s]);
#pragma unroll
    for (int k = 1; k < 4; k++) x[s][k] = cmul(x[s][k], w[k - 1]);
  }
  exchange4<S, 1>(x, lane, sre, sim);
#pragma unroll
  for (int k = 1; k < 4; k++) w[k - 1] = tab[TAB_C + (k - 1) * 4 + (lane & 3)];
#pragma unroll
  for (int s = 0; s < S; s++) {
    dft4<false>(x[s]);
#pragma unroll
    for (int k = 1; k < 4; k++) x[s][k] = cmul(x[s][k], w[k - 1]);
  }
  exchange4<S, 0>(x, lane, sre, sim);
#pragma unroll
  for (int s = 0; s < S; s++) dft4<false>(x[s]);
}

// ---- radix-4 pass with its input twiddles folded into the butterflies ----
// A twiddle w = c (1 + i t) times x is c y with y = (x.re - t x.im, x.im + t x.re): two fused
// multiply-adds. The cosine c is not applied there: the butterfly that combines y_b with its
// partner y_a computes y_a + (c_b / c_a) y_b, and the factor c_a left on the pair is taken by the
// second stage in the same way against the chain of input 0, whose factor is 1. The eight
// additions of the 4-point transform become fused multiply-adds and its three complex
// multiplications take 6 instructions instead of 12. A twiddle of exactly +-i (c = 0) is stored
// with c = 2^-60: its tangent is then finite and the value moves by 2^-60 of itself.
struct Tw4 {
  double t1, t2, t3;     // tangents of the angles of inputs 1..3
  double c2, r31, c1;    // cosines: c2, c3 / c1, c1
  double2 w0;            // twiddle of input 0 (last inverse pass only)
};
__device__ __forceinline__ double2 rfma(double s, double2 a, double2 b) {   // s a + b, s real
  return make_double2(fma(s, a.x, b.x), fma(s, a.y, b.y));
}
template <bool INV>
__device__ __forceinline__ double2 half_twiddle(double2 x, double t) {
  return INV ? make_double2(fma(t, x.y, x.x), fma(-t, x.x, x.y)) : make_double2(fma(-t, x.y, x.x), fma(t, x.x, x.y));
}
// INV: the inputs carry the conjugates of the twiddles and the transform is e^{+2 pi i jk/4}
template <bool INV, bool X0TW>
__device__ __forceinline__ void dft4_tw(double2 (&x)[4], const Tw4& k) {
  const double2 y0 = X0TW ? (INV ? cmul_conj(x[0], k.w0) : cmul(x[0], k.w0)) : x[0];
  const double2 y1 = half_twiddle<INV>(x[1], k.t1), y2 = half_twiddle<INV>(x[2], k.t2), y3 = half_twiddle<INV>(x[3], k.t3);
  const double2 s0 = rfma(k.c2, y2, y0), d0 = rfma(-k.c2, y2, y0);
  const double2 S1 = rfma(k.r31, y3, y1), D1 = rfma(-k.r31, y3, y1);   // pending c1
  const double2 T = INV ? mul_pos_i(D1) : mul_neg_i(D1);
  x[0] = rfma(k.c1, S1, s0);
  x[2] = rfma(-k.c1, S1, s0);
  x[1] = rfma(k.c1, T, d0);
  x[3] = rfma(-k.c1, T, d0);
}
template <bool X0TW>
__device__ __forceinline__ Tw4 load_tw4(const double2* p, int stride) {
  Tw4 k;
  const double2 p0 = p[0], p1 = p[stride], p2 = p[2 * stride];
  k.t1 = p0.x; k.t2 = p0.y; k.t3 = p1.x; k.c2 = p1.y; k.r31 = p2.x; k.c1 = p2.y;
  k.w0 = X0TW ? p[3 * stride] : make_double2(1.0, 0.0);
  return k;
}
// the per-register part of the twist, e^{i pi a/8} on input a of the first forward pass: compile-time constants
__device__ __forceinline__ Tw4 twist_tw4() {
  Tw4 k;
  k.t1 = TWIST4_IM[1] / TWIST4_RE[1]; k.t2 = TWIST4_IM[2] / TWIST4_RE[2]; k.t3 = TWIST4_IM[3] / TWIST4_RE[3];
  k.c2 = TWIST4_RE[2]; k.r31 = TWIST4_RE[3] / TWIST4_RE[1]; k.c1 = TWIST4_RE[1];
  k.w0 = make_double2(1.0, 0.0);
  return k;
}

// Twisted forward transform of the bootstrap kernels: x[a] = (untwisted) point 64a + lane in, the
// spectrum of fft256_forward_multi out. `tab` is the kernels' own LDS image (TWS_*).
template <int S>
__device__ __forceinline__ void fft256_forward_tw(double2 (&x)[S][4], int lane, const double2* tab, double* sre,
                                                  double* sim) {
  {
    const Tw4 k = twist_tw4();
#pragma unroll
    for (int s = 0; s < S; s++) dft4_tw<false, false>(x[s], k);
  }
  exchange4<S, 2>(x, lane, sre, sim);
  {
    const Tw4 k = load_tw4<false>(tab + TWS_F1 + (lane >> 4), 4);
#pragma unroll
    for (int s = 0; s < S; s++) dft4_tw<false, false>(x[s], k);
  }
  exchange4<S, 1>(x, lane, sre, sim);
  {
    const Tw4 k = load_tw4<false>(tab + TWS_F2 + (lane >> 2), 16);
#pragma unroll
    for (int s = 0; s < S; s++) dft4_tw<false, false>(x[s], k);
  }
  exchange4<S, 0>(x, lane, sre, sim);
  {
    const Tw4 k = load_tw4<false>(tab + TWS_F3 + lane, 64);
#pragma unroll
    for (int s = 0; s < S; s++) dft4_tw<false, false>(x[s], k);
  }
}

// its unnormalised inverse up to the per-register part of the twist (the caller applies untwist4)
template <int S>
__device__ __forceinline__ void fft256_inverse_tw(double2 (&x)[S][4], int lane, const double2* tab, double* sre,
                                                  double* sim) {
#pragma unroll
  for (int s = 0; s < S; s++) dft4<true>(x[s]);
  exchange4<S, 0>(x, lane, sre, sim);
  {
    const Tw4 k = load_tw4<false>(tab + TWS_I1 + (lane & 3), 4);
#pragma unroll
    for (int s = 0; s < S; s++) dft4_tw<true, false>(x[s], k);
  }
  exchange4<S, 1>(x, lane, sre, sim);
  {
    const Tw4 k = load_tw4<false>(tab + TWS_I2 + (lane & 15), 16);
#pragma unroll
    for (int s = 0; s < S; s++) dft4_tw<true, false>(x[s], k);
  }
  exchange4<S, 2>(x, lane, sre, sim);
  {
    const Tw4 k = load_tw4<true>(tab + TWS_I3 + lane, 64);
#pragma unroll
    for (int s = 0; s < S; s++) dft4_tw<true, true>(x[s], k);
  }
}
__device__ __forceinline__ double2 untwist4(double2 v, int a) {
  return a == 0 ? v : cmul_conj(v, make_double2(TWIST4_RE[a], TWIST4_IM[a]));
}
// the bootstrap kernels' LDS image of the tables
__device__ __forceinline__ void stage_tws(double2* tab, const double2* __restrict__ tables, int threads) {
  for (int t = threadIdx.x; t < TWS_COUNT; t += threads) tab[t] = tables[TAB_TWS + t];
}

// key conversion: one wave per polynomial, same transform, pre-scaled by 1/256
template <typename T>
__global__ __launch_bounds__(256) void bsk_fourier_kernel(const T* __restrict__ bsk, double2* __restrict__ out,
                                                          const double2* __restrict__ tables, size_t polys) {
  extern __shared__ __align__(16) char smem[];
  double2* tab = (double2*)smem;
  for (int t = threadIdx.x; t < TAB_COUNT; t += 256) tab[t] = tables[t];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* sre = (double*)(smem + TAB_COUNT * sizeof(double2)) + (size_t)wave * 2 * PLANE;
  double* sim = sre + PLANE;
  const size_t poly = (size_t)blockIdx.x * WAVES + wave;
  if (poly >= polys) return;
  using SG = typename Torus<T>::Signed;
  constexpr double SCALE = (sizeof(T) == 8 ? 0x1.0p-64 : 0x1.0p-32) / 256.0;
  const T* p = bsk + poly * N;
  double2 x[1][4];
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int j = 64 * a + lane;
    const double re = signed_to_double((SG)p[j]) * SCALE;
    const double im = signed_to_double((SG)p[j + M]) * SCALE;
    x[0][a] = cmul(make_double2(re, im), tab[TAB_TWIST + j]);
  }
  fft256_forward_multi<1>(x, lane, tab, sre, sim);
  double2* dst = out + poly * M;
#pragma unroll
  for (int g = 0; g < 4; g++) dst[g * 64 + lane] = x[0][g];
}

// One step of the balanced decomposition, the digits of decomp_next() for every state below
// 2^(32 - BL): with res the low BL bits, the carry is 1 for res > B/2 and, at res == B/2, the top bit of
// the next digit's field. That is the carry out of bit BL - 1 of res + (B/2 - 1) + that bit, so the next
// state is one add and one shift away and the digit is what is left of the state (four instructions
// for seven).
template <int BL>
__device__ __forceinline__ int32_t next_digit(uint32_t& state) {
  const uint32_t next = (state + ((1u << (BL - 1)) - 1u) + __builtin_amdgcn_ubfe(state, 2 * BL - 1, 1)) >> BL;
  const int32_t digit = (int32_t)(state - (next << BL));
  state = next;
  return digit;
}

// STREAM: no key slab is parked; the key values of a CMUX come from the L2 inside its
// multiply-accumulate, a few slabs ahead of their products. The wave then fits half the registers
// of a SIMD, and two workgroups share a CU: while one wave sits in an LDS exchange or waits for a
// key slab, its partner on the SIMD has the vector ALU.
template <typename T, int K, int L, int BL, bool STREAM>
__device__ __forceinline__ void pbs_body(const PbsLaunch<T>& a) {
  constexpr int R = K + 1;
  constexpr int WPB = STREAM ? STREAM_WAVES : WAVES;
  extern __shared__ __align__(16) char smem[];
  double2* tab = (double2*)smem;
  typedef typename AccWord<T, L, STREAM>::type A;
  uint32_t* progress = (uint32_t*)(smem + TWS_COUNT * sizeof(double2) + (size_t)WPB * slab_bytes<K>(sizeof(A)));
  stage_tws(tab, a.tables, WPB * 64);
  if (STREAM && threadIdx.x < WPB) progress[threadIdx.x] = 0;   // a wave without a ciphertext never writes its word
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr bool HIGH = sizeof(A) < sizeof(T);      // accumulator of high dwords
  char* slab = smem + TWS_COUNT * sizeof(double2) + (size_t)wave * slab_bytes<K>(sizeof(A));
  A* ct0 = (A*)slab;
  double* sre = (double*)(slab + (size_t)R * N * sizeof(A));
  double* sim = sre + PLANE;
  constexpr int S = R * L;          // forward streams: (level - 1) * R + row
  constexpr int NS = S * R;         // key slabs per iteration: [level][row][col], 4 KiB each
  constexpr int NA = STREAM ? 0 : (NS < 14 ? NS : 14);   // slabs parked in accumulation registers (at most 224 AGPRs)
  constexpr int NAD = NA ? NA : 1, NVD = (!STREAM && NS > NA) ? NS - NA : 1;   // array extents
#ifndef W4_AHEAD
#define W4_AHEAD 2
#endif
  constexpr int AHEAD = W4_AHEAD;         // STREAM: key slabs in flight
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const __amdgpu_buffer_rsrc_t key_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.fbsk, 0, (int)0x7FFFFFFF, 0x00020000);
  // persistent waves restart together per XCD (see wave_pbs_kernel_v2): key rows are then fetched
  // into each L2 once instead of once per drifting wave
  XcdRendezvous rendezvous;
  if (a.scratch) rendezvous.enrol(a.scratch, lane);
  const size_t per_round = (size_t)gridDim.x * WPB;
  uint32_t round = 0;
  for (size_t ct = (size_t)blockIdx.x * WPB + wave; ct < a.count; ct += per_round, round++) {
    const T* in = a.ct_in + ct * a.in_stride;
    const uint32_t n = a.n;
    const uint32_t lut_id = checked_lut_id(a, ct);
    const T* lut = a.luts + (size_t)lut_id * (R * N);
    if (STREAM && lane == 0) progress[wave] = 0;
    {
      const uint32_t b_hat = modswitch<T>(in[n], LOGN);
      const uint32_t d0 = (2 * N - b_hat) & (2 * N - 1);
      for (int idx = lane; idx < R * N; idx += 64) {
        const int c = idx >> LOGN, j = idx & (N - 1);
        const T w = rotated_coeff<T>(lut + (size_t)c * N, j, d0, N);
        if constexpr (HIGH) ct0[idx] = (A)(w >> 32); else ct0[idx] = (A)w;
      }
    }
    uint32_t i = 0;
    T a_cur = wave_uniform(in[0]);
    while (i < n && a_cur == 0) {
      i++;
      a_cur = wave_uniform(in[i]);
    }
    v4u bqa[NAD][4];
#pragma unroll
    for (int sl = 0; sl < NA; sl++)
#pragma unroll
      for (int g = 0; g < 4; g++) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(bqa[sl][g].x));
    bool work = false;
    uint32_t a_hat = 0, cur = 0;
    for (;;) {
      double2 x[S][4];
      if constexpr (STREAM) {
        // The SIMD's arbiter favours the older of its two waves; each wave publishes its iteration count
        // and takes the higher priority while it is behind its partner, so that the pair stays in step
        // (plain LDS accesses between compiler fences).
        asm volatile("" ::: "memory");
        if (lane == 0) progress[wave] = cur;
        const uint32_t other = __builtin_amdgcn_readfirstlane(progress[wave ^ 4]);
        asm volatile("" ::: "memory");
        if (other > cur)
          __builtin_amdgcn_s_setprio(2);
        else
          __builtin_amdgcn_s_setprio(0);
      }
      v4u bqv[NVD][4];
      v4u kst[STREAM ? AHEAD : 1][4];
      uint32_t key_off = 0;
      auto request = [&](int sl) {          // STREAM: slab sl of the current CMUX into its slot
#pragma unroll
        for (int g = 0; g < 4; g++)
          kst[sl % AHEAD][g] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(
              key_rsrc, (int)(lane_off + g * 1024), (int)(key_off + (uint32_t)sl * 4096u), 0));
      };
      if (work) {
#pragma unroll
        for (int r = 0; r < R; r++) {
          const A* p = ct0 + r * N;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint32_t j0 = 64 * q + lane, j1 = j0 + M;
            const uint32_t s0 = j0 - a_hat, s1 = j1 - a_hat;
            const A rot0 = p[s0 & (N - 1)], rot1 = p[s1 & (N - 1)];
            if constexpr (HIGH) {
              const int32_t g0 = digit_of_rotated_difference32<BL, LOGN>(rot0, p[j0], s0);
              const int32_t g1 = digit_of_rotated_difference32<BL, LOGN>(rot1, p[j1], s1);
              x[r][q] = make_double2((double)g0, (double)g1);
            } else if constexpr (L == 1) {
              const int32_t g0 = digit_of_rotated_difference<T, BL, LOGN>(rot0, p[j0], s0);
              const int32_t g1 = digit_of_rotated_difference<T, BL, LOGN>(rot1, p[j1], s1);
              x[r][q] = make_double2((double)g0, (double)g1);   // the twist is in the transform
            } else {
              const T r0 = (s0 & N) ? (T)(T(0) - rot0) : rot0;
              const T r1 = (s1 & N) ? (T)(T(0) - rot1) : rot1;
              uint32_t st0 = digit_state32<T, BL, L>((T)(r0 - p[j0]));
              uint32_t st1 = digit_state32<T, BL, L>((T)(r1 - p[j1]));
#pragma unroll
              for (int l = L; l >= 1; l--) {
                const int32_t g0 = next_digit<BL>(st0), g1 = next_digit<BL>(st1);
                x[(l - 1) * R + r][q] = make_double2((double)g0, (double)g1);
              }
            }
          }
        }
        if constexpr (STREAM) {
          key_off = cur * (uint32_t)(NS * M * sizeof(double2));
#pragma unroll
          for (int sl = 0; sl < AHEAD && sl < NS; sl++) request(sl);
        }
        // the slabs that do not fit the accumulation registers: requested now, needed after the transforms
#pragma unroll
        for (int sl = NA; sl < (STREAM ? 0 : NS); sl++) {
          const char* base = (const char*)(a.fbsk + (size_t)cur * NS * M) + (size_t)sl * (M * sizeof(double2)) + 2048;
#pragma unroll
          for (int g = 0; g < 4; g++)
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"
                         : "=v"(bqv[sl - NA][g])
                         : "v"(lane_off), "s"(base), "n"(g * 1024 - 2048)
                         : "memory");
        }
        fft256_forward_tw<S>(x, lane, tab, sre, sim);
      }
      double2 acc[R][4];
      if (!work) {
#pragma unroll
        for (int c = 0; c < R; c++)
#pragma unroll
          for (int g = 0; g < 4; g++) acc[c][g] = make_double2(0.0, 0.0);
      } else if constexpr (STREAM) {
        // slabs in memory order [stream s][column c]: every (column, register) sum takes its products in
        // the order of the parked form, so both forms give the same words
#pragma unroll
        for (int sl = 0; sl < NS; sl++) {
          const int sidx = sl / R, c = sl % R;
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const double2 kv = as_double2(kst[sl % AHEAD][g]);
            acc[c][g] = sidx == 0 ? cmul(kv, x[0][g]) : cfma(kv, x[sidx][g], acc[c][g]);
          }
          if (sl + AHEAD < NS) request(sl + AHEAD);
        }
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int sl = 0; sl < NA; sl++)
          asm volatile("" : "+a"(bqa[sl][0]), "+a"(bqa[sl][1]), "+a"(bqa[sl][2]), "+a"(bqa[sl][3]));
#pragma unroll
        for (int sl = NA; sl < NS; sl++)
          asm volatile("" : "+v"(bqv[sl - NA][0]), "+v"(bqv[sl - NA][1]), "+v"(bqv[sl - NA][2]), "+v"(bqv[sl - NA][3]));
        auto key = [&](int s, int c, int g) -> double2 {
          const int sl = s * R + c;
          return as_double2(sl < NA ? bqa[sl < NA ? sl : 0][g] : bqv[sl >= NA ? sl - NA : 0][g]);
        };
#pragma unroll
        for (int c = 0; c < R; c++)
#pragma unroll
          for (int g = 0; g < 4; g++) {
            double2 v = cmul(key(0, c, g), x[0][g]);
#pragma unroll
            for (int s = 1; s < S; s++) v = cfma(key(s, c, g), x[s][g], v);
            acc[c][g] = v;
          }
      }
      // key rows of the next active iteration (index i), if any
      const bool more = i < n;
      const char* ggsw = (const char*)(a.fbsk + (size_t)(more ? i : 0) * NS * M);
      if (more && !STREAM) {
#pragma unroll
        for (int sl = 0; sl < NA; sl++) {
          const char* base = ggsw + (size_t)sl * (M * sizeof(double2)) + 2048;
#pragma unroll
          for (int g = 0; g < 4; g++)
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"
                         : "=a"(bqa[sl][g])
                         : "v"(lane_off), "s"(base), "n"(g * 1024 - 2048)
                         : "memory");
        }
      }
      const uint32_t next_i = i;
      const uint32_t next_hat = modswitch<T>(a_cur, LOGN) & (2 * N - 1);
      if (more) {
        i++;
        a_cur = wave_uniform(in[i]);
        while (i < n && a_cur == 0) {
          i++;
          a_cur = wave_uniform(in[i]);
        }
      }
      fft256_inverse_tw<R>(acc, lane, tab, sre, sim);
#pragma unroll
      for (int c = 0; c < R; c++) {
        A* pc = ct0 + c * N;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t j0 = 64 * q + lane;
          const double2 v = untwist4(acc[c][q], q);
          pc[j0] = torus_accumulate(pc[j0], v.x);
          pc[j0 + M] = torus_accumulate(pc[j0 + M], v.y);
        }
      }
      if (!more) break;
      work = true;
      a_hat = next_hat;
      cur = next_i;
    }
    T* out = a.ct_out + ct * a.out_stride;
    for (int idx = lane; idx < K * N; idx += 64) {
      const int c = idx >> LOGN, j = idx & (N - 1);
      const A* p = ct0 + c * N;
      const A w = (j == 0) ? p[0] : (A)(A(0) - p[N - j]);
      if constexpr (HIGH) out[idx] = (T)w << 32; else out[idx] = (T)w;
    }
    if (lane == 0) {
      if constexpr (HIGH) out[K * N] = (T)ct0[K * N] << 32; else out[K * N] = (T)ct0[K * N];
      if (STREAM) progress[wave] = n;   // done: the partner keeps the higher priority until it is, too
    }
    if ((size_t)(round + 1) * per_round < a.count) rendezvous.arrive_and_wait(round, lane);
  }
}

template <typename T, int K, int L, int BL>
__global__ __launch_bounds__(256, 1) void pbs_kernel(PbsLaunch<T> a) {
  pbs_body<T, K, L, BL, false>(a);
}
template <typename T, int K, int L, int BL>
__global__ __launch_bounds__(64 * STREAM_WAVES, 2) void pbs_kernel_stream(PbsLaunch<T> a) {
  pbs_body<T, K, L, BL, true>(a);
}

// Latency variant (see wave_pbs_kernel_lat): k+1 waves per ciphertext, wave c owning polynomial
// c. It transforms the L digit polynomials of its own polynomial, publishes them, and after the
// barrier accumulates column c over all (k+1) L spectra in the order of pbs_kernel, so both give
// the same ciphertext words.
template <typename T>
constexpr size_t lat_slab_bytes() { return (size_t)N * sizeof(T) + 2 * PLANE * sizeof(double); }
template <typename T, int K, int L>
constexpr size_t lat_lds_bytes() {
  return TWS_COUNT * sizeof(double2) + (K + 1) * lat_slab_bytes<T>() + 2 * (size_t)(K + 1) * L * M * sizeof(double2);
}

template <typename T, int K, int L, int BL>
__global__ __launch_bounds__(64 * (K + 1)) void pbs_kernel_lat(PbsLaunch<T> a) {
  constexpr int R = K + 1, S = R * L, NS = S * R;
#ifndef W4_LOOP_PEEK
#define W4_LOOP_PEEK (L > 1)
#endif
  constexpr bool LOOP_PEEK = W4_LOOP_PEEK;
  extern __shared__ __align__(16) char smem[];
  double2* tab = (double2*)smem;
  stage_tws(tab, a.tables, 64 * R);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // = the polynomial this wave owns
  char* slab = smem + TWS_COUNT * sizeof(double2) + (size_t)wave * lat_slab_bytes<T>();
  T* stage = (T*)slab;
  double* sre = (double*)(slab + (size_t)N * sizeof(T));
  double* sim = sre + PLANE;
  double2* spectra = (double2*)(smem + TWS_COUNT * sizeof(double2) + (size_t)R * lat_slab_bytes<T>());  // [2][S][4][64]
  const uint32_t n = a.n;
  uint32_t image = 0;

  for (size_t ct = blockIdx.x; ct < a.count; ct += gridDim.x) {
    const T* in = a.ct_in + ct * a.in_stride;
    const uint32_t lut_id = checked_lut_id(a, ct);
    const T* lut = a.luts + (size_t)lut_id * (R * N) + (size_t)wave * N;
    // acc[2q] = coefficient 64q + lane, acc[2q+1] = coefficient 64q + lane + N/2 of the own polynomial
    T acc[8];
    {
      const uint32_t b_hat = modswitch<T>(in[n], LOGN);
      const uint32_t d0 = (2 * N - b_hat) & (2 * N - 1);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        acc[2 * q] = rotated_coeff<T>(lut, 64 * q + lane, d0, N);
        acc[2 * q + 1] = rotated_coeff<T>(lut, 64 * q + lane + M, d0, N);
      }
    }
    auto cmux = [&](uint32_t i, T a_cur) {
      const uint32_t a_hat = modswitch<T>(a_cur, LOGN) & (2 * N - 1);

      // column `wave` of key row i, every stream
      const double2* key = a.fbsk + (size_t)i * NS * M + (size_t)wave * M + lane;
      double2 kk[S][4];
#pragma unroll
      for (int st = 0; st < S; st++)
#pragma unroll
        for (int g = 0; g < 4; g++) kk[st][g] = key[(size_t)(st * R) * M + g * 64];

      double2 x[L][4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        stage[64 * q + lane] = acc[2 * q];
        stage[64 * q + lane + M] = acc[2 * q + 1];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t j0 = 64 * q + lane, j1 = j0 + M;
        const uint32_t s0 = j0 - a_hat, s1 = j1 - a_hat;
        const T rot0 = stage[s0 & (N - 1)], rot1 = stage[s1 & (N - 1)];
        if constexpr (L == 1) {
          const int32_t g0 = digit_of_rotated_difference<T, BL, LOGN>(rot0, acc[2 * q], s0);
          const int32_t g1 = digit_of_rotated_difference<T, BL, LOGN>(rot1, acc[2 * q + 1], s1);
          x[0][q] = make_double2((double)g0, (double)g1);   // the twist is in the transform
        } else {
          const T r0 = (s0 & N) ? (T)(T(0) - rot0) : rot0;
          const T r1 = (s1 & N) ? (T)(T(0) - rot1) : rot1;
          uint32_t st0 = digit_state32<T, BL, L>((T)(r0 - acc[2 * q]));
          uint32_t st1 = digit_state32<T, BL, L>((T)(r1 - acc[2 * q + 1]));
#pragma unroll
          for (int l = L; l >= 1; l--) {
            const int32_t g0 = next_digit<BL>(st0), g1 = next_digit<BL>(st1);
            x[l - 1][q] = make_double2((double)g0, (double)g1);
          }
        }
      }
      fft256_forward_tw<L>(x, lane, tab, sre, sim);

      double2* img = spectra + (size_t)image * (S * M);
#pragma unroll
      for (int l = 0; l < L; l++)
#pragma unroll
        for (int g = 0; g < 4; g++) img[((l * R + wave) * 4 + g) * 64 + lane] = x[l][g];
      __syncthreads();
      image ^= 1u;

      double2 y[1][4];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        double2 v = cmul(kk[0][g], img[(0 * 4 + g) * 64 + lane]);
#pragma unroll
        for (int st = 1; st < S; st++) v = cfma(kk[st][g], img[(st * 4 + g) * 64 + lane], v);
        y[0][g] = v;
      }
      fft256_inverse_tw<1>(y, lane, tab, sre, sim);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double2 v = untwist4(y[0][q], q);
        acc[2 * q] = torus_accumulate(acc[2 * q], v.x);
        acc[2 * q + 1] = torus_accumulate(acc[2 * q + 1], v.y);
      }
    };
    // Iterations whose mask element is zero are skipped (by every wave of the group alike). Two
    // loop forms, as in the N = 1024 variant: each instance takes the one that measured faster.
    if constexpr (LOOP_PEEK) {
      uint32_t i = 0;
      T a_cur = wave_uniform(in[0]);
      while (i < n && a_cur == 0) {
        i++;
        a_cur = wave_uniform(in[i]);  // in[n] is the body: readable
      }
      while (i < n) {
        const T a_peek = in[i + 1];   // requested before the CMUX, looked at after it
        cmux(i, a_cur);
        uint32_t j = i + 1;
        T a_nx = wave_uniform(a_peek);
        while (j < n && a_nx == 0) {
          j++;
          a_nx = wave_uniform(in[j]);
        }
        i = j;
        a_cur = a_nx;
      }
    } else {
      T a_next = wave_uniform(in[0]);
      for (uint32_t i = 0; i < n; i++) {
        const T a_cur = a_next;
        a_next = wave_uniform(in[i + 1]);
        if (a_cur == 0) continue;
        cmux(i, a_cur);
      }
    }

    // sample extraction of coefficient 0: out[c N + (N - j) mod N] = (j == 0) ? p[j] : -p[j]
    T* out = a.ct_out + ct * a.out_stride;
    if (wave < K) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t j0 = 64 * q + lane, j1 = j0 + M;
        out[(size_t)wave * N + ((N - j0) & (N - 1))] = (j0 == 0) ? acc[2 * q] : (T)(T(0) - acc[2 * q]);
        out[(size_t)wave * N + (N - j1)] = (T)(T(0) - acc[2 * q + 1]);
      }
    } else if (lane == 0) {
      out[K * N] = acc[0];
    }
  }
}

inline bool supported(const camd_params& p) {
  if (p.polynomial_size != 512) return false;
  if (p.torus_bits == 32) return p.glwe_dimension == 2 && p.pbs_level == 2 && p.pbs_base_log == 8;
  return p.glwe_dimension == 3 && p.pbs_level == 1 && p.pbs_base_log == 18;   // the three small shortint sets
}

inline void fill_tables(double2* t) {
  for (int j = 0; j < 256; j++) {
    const double ang = M_PI * (double)j / (double)N;
    t[TAB_TWIST + j] = make_double2(std::cos(ang), std::sin(ang));
  }
  for (int k = 1; k < 4; k++) {
    for (int m = 0; m < 64; m++) {
      const double ang = -2.0 * M_PI * (double)(m * k) / 256.0;
      t[TAB_A + (k - 1) * 64 + m] = make_double2(std::cos(ang), std::sin(ang));
    }
    for (int m = 0; m < 16; m++) {
      const double ang = -2.0 * M_PI * (double)(m * k) / 64.0;
      t[TAB_B + (k - 1) * 16 + m] = make_double2(std::cos(ang), std::sin(ang));
    }
    for (int m = 0; m < 4; m++) {
      const double ang = -2.0 * M_PI * (double)(m * k) / 16.0;
      t[TAB_C + (k - 1) * 4 + m] = make_double2(std::cos(ang), std::sin(ang));
    }
  }
  // folded constants of the twiddled passes (see dft4_tw): th[j] is the angle of input j
  auto fold = [&](double2* dst, int stride, const double (&th)[4], bool own_w0) {
    double c[4], tn[4];
    for (int j = 0; j < 4; j++) {
      c[j] = std::cos(th[j]);
      const double sn = std::sin(th[j]);
      if (std::fabs(c[j]) < 1e-9) c[j] = 0x1.0p-60;   // a quarter turn: finite tangent, value off by 2^-60
      tn[j] = sn / c[j];
    }
    dst[0] = make_double2(tn[1], tn[2]);
    dst[stride] = make_double2(tn[3], c[2]);
    dst[2 * stride] = make_double2(c[3] / c[1], c[1]);
    if (own_w0) dst[3 * stride] = make_double2(std::cos(th[0]), std::sin(th[0]));
  };
  auto theta = [](int k) { return M_PI * (double)(1 - 4 * k) / (double)N; };
  double2* tws = t + TAB_TWS;
  for (int lane = 0; lane < 64; lane++) {
    const int d2 = lane >> 4, d1 = (lane >> 2) & 3, d0 = lane & 3;
    double th[4];
    for (int j = 0; j < 4; j++) th[j] = j * 16.0 * theta(d2);
    fold(tws + TWS_F1 + d2, 4, th, false);
    for (int j = 0; j < 4; j++) th[j] = j * (4.0 * theta(d2) - M_PI * d1 / 8.0);
    fold(tws + TWS_F2 + (lane >> 2), 16, th, false);
    for (int j = 0; j < 4; j++) th[j] = j * (theta(d2) - 2.0 * M_PI * (4 * d0 + d1) / 64.0);
    fold(tws + TWS_F3 + lane, 64, th, false);
    for (int k = 0; k < 4; k++) th[k] = -2.0 * M_PI * (double)(d0 * k) / 16.0;
    fold(tws + TWS_I1 + d0, 4, th, false);
    for (int k = 0; k < 4; k++) th[k] = -2.0 * M_PI * (double)((lane & 15) * k) / 64.0;
    fold(tws + TWS_I2 + (lane & 15), 16, th, false);
    for (int k = 0; k < 4; k++) th[k] = theta(k) * lane;
    fold(tws + TWS_I3 + lane, 64, th, true);
  }
}

template <typename T, int K, int L, int BL>
int launch(const PbsLaunch<T>& a, hipStream_t stream) {
  static_assert(lds_bytes<K>(sizeof(T)) <= 160 * 1024, "LDS budget exceeded");
  // CAMD_WAVE_SCHEDULE=1: the parked-key form, one workgroup per CU; default: the streamed form, two per CU
  static const bool parked = []() {
    const char* e = getenv("CAMD_WAVE_SCHEDULE");
    return e && atoi(e) == 1;
  }();
  auto kern = parked ? pbs_kernel<T, K, L, BL> : pbs_kernel_stream<T, K, L, BL>;
  const size_t lds = parked ? lds_bytes<K>(sizeof(T))
                            : lds_bytes<K>(sizeof(typename AccWord<T, L, true>::type), STREAM_WAVES);
  static_assert(lds_bytes<K>(sizeof(typename AccWord<T, L, true>::type), STREAM_WAVES) <= 160 * 1024, "LDS budget exceeded");
  if (lds > 64 * 1024) CAMD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int dev = 0, cus = 0;
  CAMD_HIP(hipGetDevice(&dev));
  CAMD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  constexpr int lat_fit = (int)((160 * 1024) / lat_lds_bytes<T, K, L>());
  if (a.count <= wave_latency_limit(cus, lat_fit)) {
    constexpr size_t ldsl = lat_lds_bytes<T, K, L>();
    static_assert(ldsl <= 160 * 1024, "LDS budget exceeded");
    auto kernl = pbs_kernel_lat<T, K, L, BL>;
    if (ldsl > 64 * 1024) CAMD_HIP(hipFuncSetAttribute((const void*)kernl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsl));
    hipLaunchKernelGGL(kernl, dim3((unsigned)a.count), dim3(64 * (K + 1)), ldsl, stream, a);
    CAMD_HIP(hipGetLastError());
    return CAMD_OK;
  }
  const unsigned wpb = parked ? (unsigned)WAVES : (unsigned)STREAM_WAVES;
  const unsigned grid = (unsigned)((a.count + wpb - 1) / wpb);
  const unsigned persistent = cus > 0 && grid > (unsigned)cus ? (unsigned)cus : grid;   // one workgroup per CU
  if (persistent < grid) {
    if (!a.scratch) return fail(CAMD_EINVAL, "wave kernel: arrival counters were not reserved");
    CAMD_HIP(hipMemsetAsync(a.scratch, 0, RENDEZVOUS_BYTES, stream));
  }
  hipLaunchKernelGGL(kern, dim3(persistent), dim3(64 * wpb), lds, stream, a);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}
}  // namespace w4

}  // namespace

bool wave512_supported(const camd_params& p) { return w4::supported(p); }
size_t wave512_tables_count() { return (size_t)w4::TAB_TOTAL; }
void wave512_fill_tables(double2* t) { w4::fill_tables(t); }

template <typename T>
int wave512_bsk_to_fourier(const T* bsk_std, double2* fbsk, const double2* tables, size_t polys, hipStream_t stream) {
  const size_t lds = w4::TAB_COUNT * sizeof(double2) + WAVES * 2 * w4::PLANE * sizeof(double);
  hipLaunchKernelGGL(w4::bsk_fourier_kernel<T>, dim3((unsigned)((polys + WAVES - 1) / WAVES)), dim3(256), lds, stream,
                     bsk_std, fbsk, tables, polys);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

template <typename T>
int wave512_launch_pbs(const PbsLaunch<T>& a, hipStream_t stream) {
  if constexpr (sizeof(T) == 8) {
    if (a.k == 3 && a.level == 1 && a.base_log == 18) return w4::launch<T, 3, 1, 18>(a, stream);
  } else {
    if (a.k == 2 && a.level == 2 && a.base_log == 8) return w4::launch<T, 2, 2, 8>(a, stream);
  }
  return fail(CAMD_EINVAL, "wave kernel: no instance for this parameter set");
}

template int wave512_bsk_to_fourier<uint64_t>(const uint64_t*, double2*, const double2*, size_t, hipStream_t);
template int wave512_bsk_to_fourier<uint32_t>(const uint32_t*, double2*, const double2*, size_t, hipStream_t);
template int wave512_launch_pbs<uint64_t>(const PbsLaunch<uint64_t>&, hipStream_t);
template int wave512_launch_pbs<uint32_t>(const PbsLaunch<uint32_t>&, hipStream_t);

}  // namespace camd
