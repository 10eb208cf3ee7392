// The 512-point transforms of the N = 1024 wave kernels (pbs_wave.hip, pbs_wave8.hip): radix
// 8 x 8 x 8 on one 64-lane wave, eight points per lane, two exchanges through a padded LDS image;
// twist and twiddles folded into the butterflies. See pbs_wave.hip for the design notes.
#pragma once
#include "wave_common.hpp"

namespace camd {

namespace {

constexpr int RADIX = 8;          // points per lane (N = 1024)
constexpr int ROW = 72;           // padded row of an exchange plane, in doubles: every ds_write_b64 / ds_read_b64 lane group of both exchanges stays on distinct banks
constexpr int PLANE = RADIX * ROW;

// table layout, in double2 entries
constexpr int TAB_TWIST = 0;                   // [a][lane]  e^{+i pi (64a+lane)/N}
constexpr int TAB_TW0 = 512;                   // [b-1][lane] e^{-2 pi i lane b / 512}, b = 1..7
constexpr int TAB_TW1 = TAB_TW0 + 7 * 64;      // [b-1][m]    e^{-2 pi i m b / 64},    b = 1..7, m < 8
constexpr int TAB_COUNT = TAB_TW1 + 7 * 8;     // what the key-conversion kernel stages in LDS
// The bootstrap kernels split the twist e^{i pi (64a+lane)/N} into a per-register part
// e^{i pi a/16}, eight compile-time constants, and a per-lane part e^{i pi lane/N}. The per-lane part
// commutes with the first radix-8 pass (it is the same for the eight inputs of a lane), so it joins
// that pass's output twiddles: e^{i pi lane (1 - 4b)/N} for output b. No twist table is read.
// Every twiddle is applied on the receiving side of the exchange that follows it, as the input
// factor of the next radix-8 pass, in the folded form of dft8_tw below. The constants of a pass
// are stored as pairs of doubles: (t1,t2) (t3,t4) (t5,t6) (t7,c4) (r51,r62) (r73,c2) (r31,c1)
// (hc1,-) and, where input 0 has a twiddle of its own, w0.
constexpr int TWS_A = 0;                       // [pair 9][lane]: forward, second pass: input q of lane (hi,lo) carries e^{i pi (8q+lo)(1-4hi)/N}
constexpr int TWS_B = TWS_A + 9 * 64;          // [pair 8][lo]:   input q carries e^{-2 pi i lo q/64} (forward third pass; conjugated: inverse second pass)
constexpr int TWS_C = TWS_B + 8 * 8;           // [pair 9][lane]: inverse, third pass: input b carries the conjugate of e^{i pi lane (1-4b)/N}
constexpr int TWS_COUNT = TWS_C + 9 * 64;      // the bootstrap kernels' LDS image
constexpr int TAB_TWS = TAB_COUNT;             // where that image starts in the table buffer
constexpr int TAB_TOTAL = TAB_TWS + TWS_COUNT;

constexpr double HALF_SQRT2 = 0.70710678118654752440;
// e^{i pi a / 16}, a = 0..7
constexpr double TWIST8_RE[8] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                                 0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                                 0.19509032201612826785};
constexpr double TWIST8_IM[8] = {0.0, 0.19509032201612826785, 0.38268343236508977173, 0.55557023301960222474,
                                 0.70710678118654752440, 0.83146961230254523708, 0.92387953251128675613,
                                 0.98078528040323044913};

// 8-point DFT in registers, natural order in and out. INV selects e^{+2 pi i jk/8}.
// The two odd eighth-roots (W8 and W8^3) scale by 1/sqrt(2); that factor is not applied where it
// arises but folded into the last butterfly stage, whose additions become fused multiply-adds:
// four instructions fewer per transform of eight points.
template <bool INV>
__device__ __forceinline__ void dft8(double2 (&x)[8]) {
  const double2 a0 = cadd(x[0], x[4]), a4 = csub(x[0], x[4]);
  const double2 a1 = cadd(x[1], x[5]), d5 = csub(x[1], x[5]);
  const double2 a2 = cadd(x[2], x[6]), d6 = csub(x[2], x[6]);
  const double2 a3 = cadd(x[3], x[7]), d7 = csub(x[3], x[7]);
  // sqrt(2) d5 W8^(+-1) and sqrt(2) d7 W8^(+-3), unscaled
  double2 s5, s7;
  if (!INV) {
    s5 = make_double2(d5.x + d5.y, d5.y - d5.x);
    s7 = make_double2(d7.y - d7.x, -(d7.x + d7.y));
  } else {
    s5 = make_double2(d5.x - d5.y, d5.x + d5.y);
    s7 = make_double2(-(d7.x + d7.y), d7.x - d7.y);
  }
  const double2 a6 = INV ? mul_pos_i(d6) : mul_neg_i(d6);
  const double2 b0 = cadd(a0, a2), b2 = csub(a0, a2);
  const double2 b1 = cadd(a1, a3), e3 = csub(a1, a3);
  const double2 b4 = cadd(a4, a6), b6 = csub(a4, a6);
  const double2 u = cadd(s5, s7), v = csub(s5, s7);   // b5 = H u, e7 = H v
  const double2 b3 = INV ? mul_pos_i(e3) : mul_neg_i(e3);
  x[0] = cadd(b0, b1);
  x[4] = csub(b0, b1);
  x[2] = cadd(b2, b3);
  x[6] = csub(b2, b3);
  x[1] = make_double2(fma(HALF_SQRT2, u.x, b4.x), fma(HALF_SQRT2, u.y, b4.y));
  x[5] = make_double2(fma(-HALF_SQRT2, u.x, b4.x), fma(-HALF_SQRT2, u.y, b4.y));
  // b7 = -+i e7 = H (+-v.y, -+v.x)
  if (!INV) {
    x[3] = make_double2(fma(HALF_SQRT2, v.y, b6.x), fma(-HALF_SQRT2, v.x, b6.y));
    x[7] = make_double2(fma(-HALF_SQRT2, v.y, b6.x), fma(HALF_SQRT2, v.x, b6.y));
  } else {
    x[3] = make_double2(fma(-HALF_SQRT2, v.y, b6.x), fma(HALF_SQRT2, v.x, b6.y));
    x[7] = make_double2(fma(HALF_SQRT2, v.y, b6.x), fma(-HALF_SQRT2, v.x, b6.y));
  }
}

// Forward transform of 512 points held as x[a] = point 64a + lane (twist already applied).
// Result: x[g] = X[64 g + 8 (lane & 7) + (lane >> 3)]. Used by the key conversion.
__device__ __forceinline__ void fft512_forward(double2 (&x)[8], int lane, const double2* tab, double* sre,
                                               double* sim) {
  const int hi = lane >> 3, lo = lane & 7;
  dft8<false>(x);
#pragma unroll
  for (int b = 1; b < 8; b++) x[b] = cmul(x[b], tab[TAB_TW0 + (b - 1) * 64 + lane]);
  // exchange 1: register index b <-> lane bits 5:3
#pragma unroll
  for (int b = 0; b < 8; b++) {
    sre[b * ROW + lane] = x[b].x; sim[b * ROW + lane] = x[b].y;
  }
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int idx = hi * ROW + q * 8 + lo;
    x[q] = make_double2(sre[idx], sim[idx]);
  }
  dft8<false>(x);
#pragma unroll
  for (int b = 1; b < 8; b++) x[b] = cmul(x[b], tab[TAB_TW1 + (b - 1) * 8 + lo]);
  // exchange 2: register index b' <-> lane bits 2:0 (XOR keeps both sides conflict-free)
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const int idx = hi * ROW + b * 8 + (lo ^ b);
    sre[idx] = x[b].x; sim[idx] = x[b].y;
  }
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int idx = hi * ROW + lo * 8 + (q ^ lo);
    x[q] = make_double2(sre[idx], sim[idx]);
  }
  dft8<false>(x);
}

// ---- radix-8 pass with its input twiddles folded into the butterflies ----
// A twiddle w = c (1 + i t) (c the cosine, t the tangent of its angle) times x is c y with
// y = (x.re - t x.im, x.im + t x.re): two fused multiply-adds. The real factor c is not applied
// there. The butterfly stage that combines y_b with its partner y_a takes the ratio as the
// multiplier of a fused multiply-add, y_a + (c_b / c_a) y_b, and the factor c_a left on the pair
// moves on to the next stage in the same way, until it meets the chain of input 0, whose factor is
// 1. Every addition of the 8-point transform becomes a fused multiply-add and the seven complex
// multiplications in front of it take 14 instructions instead of 28; the result is the transform
// of the twiddled inputs with nothing left to apply. A twiddle of exactly -i (c = 0) is stored
// with c = 2^-60: its tangent is then finite and the value moves by 2^-60 of itself.
struct Tw8 {
  double t[8];                                  // t[b], b = 1..7: tangent of the angle of input b
  double c4, r51, r62, r73, c2, r31, c1, hc1;   // c_b: cosine of input b; r_ba = c_b / c_a; hc1 = c1 / sqrt 2
  double2 w0;                                   // twiddle of input 0, for the passes where it is not 1
};
__device__ __forceinline__ double2 rfma(double s, double2 a, double2 b) {   // s a + b, s real
  return make_double2(fma(s, a.x, b.x), fma(s, a.y, b.y));
}
// INV: the inputs carry the conjugates of the twiddles and the transform is e^{+2 pi i jk/8}.
template <bool INV, bool X0TW>
__device__ __forceinline__ void dft8_tw(double2 (&x)[8], const Tw8& k) {
  double2 y[8];
  y[0] = X0TW ? (INV ? cmul_conj(x[0], k.w0) : cmul(x[0], k.w0)) : x[0];
#pragma unroll
  for (int b = 1; b < 8; b++)
    y[b] = INV ? make_double2(fma(k.t[b], x[b].y, x[b].x), fma(-k.t[b], x[b].x, x[b].y))
               : make_double2(fma(-k.t[b], x[b].y, x[b].x), fma(k.t[b], x[b].x, x[b].y));
  // stage 1; capital letters carry a pending factor: A1, D5 c1; A2, D6 c2; A3, D7 c3
  const double2 a0 = rfma(k.c4, y[4], y[0]), a4 = rfma(-k.c4, y[4], y[0]);
  const double2 A1 = rfma(k.r51, y[5], y[1]), D5 = rfma(-k.r51, y[5], y[1]);
  const double2 A2 = rfma(k.r62, y[6], y[2]), D6 = rfma(-k.r62, y[6], y[2]);
  const double2 A3 = rfma(k.r73, y[7], y[3]), D7 = rfma(-k.r73, y[7], y[3]);
  double2 S5, S7;   // sqrt(2) D5 W8^(+-1) (pending c1), sqrt(2) D7 W8^(+-3) (pending c3)
  if (!INV) {
    S5 = make_double2(D5.x + D5.y, D5.y - D5.x);
    S7 = make_double2(D7.y - D7.x, -(D7.x + D7.y));
  } else {
    S5 = make_double2(D5.x - D5.y, D5.x + D5.y);
    S7 = make_double2(-(D7.x + D7.y), D7.x - D7.y);
  }
  const double2 A6 = INV ? mul_pos_i(D6) : mul_neg_i(D6);
  // stage 2; B1, E3, U, V carry c1
  const double2 b0 = rfma(k.c2, A2, a0), b2 = rfma(-k.c2, A2, a0);
  const double2 B1 = rfma(k.r31, A3, A1), E3 = rfma(-k.r31, A3, A1);
  const double2 b4 = rfma(k.c2, A6, a4), b6 = rfma(-k.c2, A6, a4);
  const double2 U = rfma(k.r31, S7, S5), V = rfma(-k.r31, S7, S5);
  const double2 B3 = INV ? mul_pos_i(E3) : mul_neg_i(E3);
  // stage 3
  x[0] = rfma(k.c1, B1, b0);
  x[4] = rfma(-k.c1, B1, b0);
  x[2] = rfma(k.c1, B3, b2);
  x[6] = rfma(-k.c1, B3, b2);
  x[1] = rfma(k.hc1, U, b4);
  x[5] = rfma(-k.hc1, U, b4);
  if (!INV) {
    x[3] = make_double2(fma(k.hc1, V.y, b6.x), fma(-k.hc1, V.x, b6.y));
    x[7] = make_double2(fma(-k.hc1, V.y, b6.x), fma(k.hc1, V.x, b6.y));
  } else {
    x[3] = make_double2(fma(-k.hc1, V.y, b6.x), fma(k.hc1, V.x, b6.y));
    x[7] = make_double2(fma(k.hc1, V.y, b6.x), fma(-k.hc1, V.x, b6.y));
  }
}
// the constants of a pass from the LDS image: `p` points at the lane's entry of pair 0
template <bool X0TW>
__device__ __forceinline__ Tw8 load_tw8(const double2* p, int stride) {
  Tw8 k;
  const double2 p0 = p[0], p1 = p[stride], p2 = p[2 * stride], p3 = p[3 * stride], p4 = p[4 * stride],
                p5 = p[5 * stride], p6 = p[6 * stride], p7 = p[7 * stride];
  k.t[0] = 0.0;
  k.t[1] = p0.x; k.t[2] = p0.y; k.t[3] = p1.x; k.t[4] = p1.y; k.t[5] = p2.x; k.t[6] = p2.y; k.t[7] = p3.x;
  k.c4 = p3.y; k.r51 = p4.x; k.r62 = p4.y; k.r73 = p5.x; k.c2 = p5.y; k.r31 = p6.x; k.c1 = p6.y; k.hc1 = p7.x;
  k.w0 = X0TW ? p[8 * stride] : make_double2(1.0, 0.0);
  return k;
}
// the per-register part of the twist, e^{i pi a/16} on input a of the first forward pass: compile-time constants
__device__ __forceinline__ Tw8 twist_tw8() {
  Tw8 k;
  k.t[0] = 0.0;
#pragma unroll
  for (int b = 1; b < 8; b++) k.t[b] = TWIST8_IM[b] / TWIST8_RE[b];
  k.c4 = TWIST8_RE[4]; k.r51 = TWIST8_RE[5] / TWIST8_RE[1]; k.r62 = TWIST8_RE[6] / TWIST8_RE[2];
  k.r73 = TWIST8_RE[7] / TWIST8_RE[3]; k.c2 = TWIST8_RE[2]; k.r31 = TWIST8_RE[3] / TWIST8_RE[1];
  k.c1 = TWIST8_RE[1]; k.hc1 = TWIST8_RE[1] * HALF_SQRT2;
  k.w0 = make_double2(1.0, 0.0);
  return k;
}

// Twisted forward transform for the bootstrap kernels: x[a] = (untwisted) point 64a + lane in,
// spectrum out in the order of fft512_forward. `tab` is the kernels' own LDS image (TWS_*).
// EARLY: the constants of a pass are requested before the exchange in front of it (they do not
// depend on the data, and a lone stream has nothing else to put under their latency); this costs
// registers, which the latency kernel has and the first schedule does not.
template <bool EARLY>
__device__ __forceinline__ void fft512_forward_tw(double2 (&x)[8], int lane, const double2* tab, double* sre,
                                                  double* sim) {
  const int hi = lane >> 3, lo = lane & 7;
  Tw8 ka, kb;
  if (EARLY) ka = load_tw8<true>(tab + TWS_A + lane, 64);
  dft8_tw<false, false>(x, twist_tw8());
#pragma unroll
  for (int b = 0; b < 8; b++) {
    sre[b * ROW + lane] = x[b].x; sim[b * ROW + lane] = x[b].y;
  }
  if (EARLY) kb = load_tw8<false>(tab + TWS_B + lo, 8);
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int idx = hi * ROW + q * 8 + lo;
    x[q] = make_double2(sre[idx], sim[idx]);
  }
  if (!EARLY) ka = load_tw8<true>(tab + TWS_A + lane, 64);
  dft8_tw<false, true>(x, ka);
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const int idx = hi * ROW + b * 8 + (lo ^ b);
    sre[idx] = x[b].x; sim[idx] = x[b].y;
  }
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int idx = hi * ROW + lo * 8 + (q ^ lo);
    x[q] = make_double2(sre[idx], sim[idx]);
  }
  if (!EARLY) kb = load_tw8<false>(tab + TWS_B + lo, 8);
  dft8_tw<false, false>(x, kb);
}

// Unnormalised inverse of fft512_forward_tw up to the per-register part of the twist:
// spectrum in, x[a] = 512 e^{i pi a/16} * point (64a + lane) out (the caller applies untwist8).
template <bool EARLY>
__device__ __forceinline__ void fft512_inverse_tw(double2 (&x)[8], int lane, const double2* tab, double* sre,
                                                  double* sim) {
  const int hi = lane >> 3, lo = lane & 7;
  Tw8 kb, kc;
  if (EARLY) kb = load_tw8<false>(tab + TWS_B + lo, 8);
  dft8<true>(x);
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int idx = hi * ROW + lo * 8 + (q ^ lo);
    sre[idx] = x[q].x; sim[idx] = x[q].y;
  }
  if (EARLY) kc = load_tw8<true>(tab + TWS_C + lane, 64);
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const int idx = hi * ROW + b * 8 + (lo ^ b);
    x[b] = make_double2(sre[idx], sim[idx]);
  }
  if (!EARLY) kb = load_tw8<false>(tab + TWS_B + lo, 8);
  dft8_tw<true, false>(x, kb);
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int idx = hi * ROW + q * 8 + lo;
    sre[idx] = x[q].x; sim[idx] = x[q].y;
  }
#pragma unroll
  for (int b = 0; b < 8; b++) x[b] = make_double2(sre[b * ROW + lane], sim[b * ROW + lane]);
  if (!EARLY) kc = load_tw8<true>(tab + TWS_C + lane, 64);
  dft8_tw<true, true>(x, kc);
}
// the inverse of the per-register part of the twist (a is a compile-time index at every call site)
__device__ __forceinline__ double2 untwist8(double2 v, int a) {
  return a == 0 ? v : cmul_conj(v, make_double2(TWIST8_RE[a], TWIST8_IM[a]));
}
// the bootstrap kernels' LDS image of the tables
__device__ __forceinline__ void stage_folded_tables(double2* tab, const double2* __restrict__ tables, int threads) {
  for (int t = threadIdx.x; t < TWS_COUNT; t += threads) tab[t] = tables[TAB_TWS + t];
}

// ---- interleaved transforms: S independent polynomials advance through the passes together ----
// One wave, one exchange image: LDS operations of a wave execute in order, so the write/read
// pair of stream s completes before the pair of stream s+1 touches the image, and while the
// data of one stream is in flight the butterflies of the others keep the vector ALU busy.
// Twiddle factors are read once per pass and reused by every stream.
template <int S>
__device__ __forceinline__ void fft512_forward_multi(double2 (&x)[S][8], int lane, const double2* tab, double* sre,
                                                     double* sim) {
  const int hi = lane >> 3, lo = lane & 7;
  {
    const Tw8 k = twist_tw8();
#pragma unroll
    for (int s = 0; s < S; s++) dft8_tw<false, false>(x[s], k);
  }
  {
    const Tw8 k = load_tw8<true>(tab + TWS_A + lane, 64);
#pragma unroll
    for (int s = 0; s < S; s++) {
#pragma unroll
      for (int b = 0; b < 8; b++) {
        sre[b * ROW + lane] = x[s][b].x; sim[b * ROW + lane] = x[s][b].y;
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int idx = hi * ROW + q * 8 + lo;
        x[s][q] = make_double2(sre[idx], sim[idx]);
      }
    }
#pragma unroll
    for (int s = 0; s < S; s++) dft8_tw<false, true>(x[s], k);
  }
  {
    const Tw8 k = load_tw8<false>(tab + TWS_B + lo, 8);
#pragma unroll
    for (int s = 0; s < S; s++) {
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const int idx = hi * ROW + b * 8 + (lo ^ b);
        sre[idx] = x[s][b].x; sim[idx] = x[s][b].y;
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int idx = hi * ROW + lo * 8 + (q ^ lo);
        x[s][q] = make_double2(sre[idx], sim[idx]);
      }
    }
#pragma unroll
    for (int s = 0; s < S; s++) dft8_tw<false, false>(x[s], k);
  }
}

struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};

// `hook(k)` is called at three points (k = 0, 1, 2), between the passes: the caller uses it to
// slip independent memory requests into the stream.
template <int S, typename Hook = NoHook>
__device__ __forceinline__ void fft512_inverse_multi(double2 (&x)[S][8], int lane, const double2* tab, double* sre,
                                                     double* sim, Hook hook = Hook()) {
  const int hi = lane >> 3, lo = lane & 7;
  hook(0);
#pragma unroll
  for (int s = 0; s < S; s++) dft8<true>(x[s]);
  {
    const Tw8 k = load_tw8<false>(tab + TWS_B + lo, 8);
#pragma unroll
    for (int s = 0; s < S; s++) {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int idx = hi * ROW + lo * 8 + (q ^ lo);
        sre[idx] = x[s][q].x; sim[idx] = x[s][q].y;
      }
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const int idx = hi * ROW + b * 8 + (lo ^ b);
        x[s][b] = make_double2(sre[idx], sim[idx]);
      }
    }
    hook(1);
#pragma unroll
    for (int s = 0; s < S; s++) dft8_tw<true, false>(x[s], k);
  }
  {
    const Tw8 k = load_tw8<true>(tab + TWS_C + lane, 64);
#pragma unroll
    for (int s = 0; s < S; s++) {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int idx = hi * ROW + q * 8 + lo;
        sre[idx] = x[s][q].x; sim[idx] = x[s][q].y;
      }
#pragma unroll
      for (int b = 0; b < 8; b++) x[s][b] = make_double2(sre[b * ROW + lane], sim[b * ROW + lane]);
    }
    hook(2);
#pragma unroll
    for (int s = 0; s < S; s++) dft8_tw<true, true>(x[s], k);
  }
}

}  // namespace

}  // namespace camd
