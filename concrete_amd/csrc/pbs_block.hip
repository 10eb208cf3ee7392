// Programmable bootstrap, register-blocked workgroup-per-ciphertext family.
//
// Replaces FftEngine::discard_bootstrap_lwe_ciphertext as the reference calls it at
// concrete-shortint/src/engine/server_side/mod.rs:148-153 and
// concrete-boolean/src/engine/bootstrapping/cpu.rs:231-236, for every polynomial size the
// wave-per-ciphertext kernel has no instance for (N = 256 ... 16384, any level count).
//
// One workgroup of M/P threads (M = N/2 complex points, P = 8 points per thread; 4 or 2 for the
// two smallest sizes) walks one ciphertext through all n CMUXes.
//   * Transforms: each thread keeps P points in registers and runs log2(P) radix-2 stages on
//     them between exchanges; an exchange goes through one LDS image of the polynomial with
//     16-byte accesses and an XOR swizzle chosen per exchange so that both the store and the
//     load side are free of bank conflicts, or, where it only trades lane bits that have a cheap
//     register transposition, through lane swaps and row shifts (see block_shuffles).
//     Stage twiddles of the first pass (the top log2(P)
//     stages, whose tables are the large ones) come from global memory, all others from a
//     compact per-stage copy in LDS.
//   * The Fourier-domain sums of the external product stay in registers ((k+1) x P complex per
//     thread): the multiply-accumulate against the key follows each forward transform directly
//     and the inverse transforms start from registers.
//   * The accumulator ((k+1) N torus words) lives in a per-workgroup global slab that stays in
//     L2 / Infinity Cache; for multi-level decompositions the decomposer state left after the
//     first level is parked there too (32 bits per coefficient), so the rotated accumulator is
//     gathered once per row and CMUX.
//   * Key layout: per polynomial [register r][thread t], i.e. exactly the order the forward
//     transform leaves its outputs in; every key access is one coalesced 16-byte-per-lane load.
//
// Floating-point contract: the same as the radix-2 fallback kernel (pbs_generic.hip). A
// radix-2 flow graph fixes, for every element, the sequence of IEEE-754 operations applied to
// it; which thread executes a butterfly does not matter. Contraction is off, fused operations
// are explicit fma(), twiddles are the host-computed tables of the fallback kernel, the
// accumulation order over (level, row) and the rounding are the same. The one product left out
// is the one by W^0 = (1, -0) (see unit_twiddle), which can only change the sign of a zero. Output
// ciphertexts are therefore bit-identical to the fallback kernel's and to the CPU restatement's.
#pragma clang fp contract(off)
#include "kernels.hpp"
#include "lane_swap.hpp"

namespace camd {

namespace {

// Configuration of the N >= 8192 instances: points per thread (log2), Fourier-sum columns kept in
// registers, waves per SIMD they are compiled for. Measured on MESSAGE_4_CARRY_4 (N = 16384, 1024
// ciphertexts), (PL, columns, waves):
//   (4, 1, 2)  504 ms   512 threads of 16 points; the second column round-trips the global slab
//                       (1 MB per CMUX and CU) and the compiler spills 123 registers
//   (4, 2, 2)  439 ms   both columns in registers, 11 spilled
//   (3, 2, 4)  405 ms   1024 threads of 8 points: one more exchange per transform, but sixteen waves
//                       per CU to hide the latency of the slab and key accesses
// MESSAGE_3_CARRY_4 (N = 8192): 201 / 184 / 153 ms; MESSAGE_1_CARRY_7 (three levels): 338 / 300 / 275 ms.
#ifndef BLOCK_BIG_PL
#define BLOCK_BIG_PL 3
#endif
#ifndef BLOCK_BIG_CREG
#define BLOCK_BIG_CREG 2
#endif
#ifndef BLOCK_BIG_WAVES
#define BLOCK_BIG_WAVES 4
#endif

constexpr uint32_t SWZ_BITS = 4;  // a 16-lane group of ds_read_b128 must touch 16 distinct 16-byte units

// Global accesses as (wave-uniform base, 32-bit byte offset): the hardware adds the scalar base
// itself, so no 64-bit address registers are built or kept alive per access.
template <typename V>
__device__ __forceinline__ V gld(const void* base, uint32_t byte_off) {
  return *(const V*)((const char*)base + byte_off);
}
template <typename V>
__device__ __forceinline__ void gst(void* base, uint32_t byte_off, V v) {
  *(V*)((char*)base + byte_off) = v;
}

// Signed decomposition on the high dword. When the decomposed field (base_log * level bits, plus
// the rounding bit below it) lies in the upper 32 bits of a 64-bit torus word, the 64-bit variable
// shifts of the generic form (two to four instructions each) are not needed. Both helpers return
// exactly what decomp_init_state / decomp_next return; `hi` is the upper dword of the word.
//   single level: digit = ((hi + 2^(S-1) + (C << S)) >> S) - C with C = B/2 - 1 and S = 32 - base_log:
//   the rounding increment and the fold of the upper half of the digit range are both additions
//   below the extracted field, so they merge into one constant (the field ends at bit 31, so the
//   wrap of the 32-bit addition is the wrap modulo B).
__device__ __forceinline__ int32_t block_digit_single(uint32_t hi, uint32_t bl) {
  const uint32_t S = 32u - bl;
  const uint32_t C = (1u << (bl - 1u)) - 1u;
  const uint32_t K = (1u << (S - 1u)) + (C << S);   // wave-uniform: scalar instructions
  return (int32_t)((hi + K) >> S) - (int32_t)C;
}
//   several levels: the state in 32 bits (base_log * level <= 31)
__device__ __forceinline__ uint32_t block_state32(uint32_t hi, uint32_t bl, uint32_t levels) {
  const uint32_t S = 32u - bl * levels;
  return ((hi + (1u << (S - 1u))) >> S) & ((1u << (bl * levels)) - 1u);
}
__device__ __forceinline__ int32_t block_next32(uint32_t& state, uint32_t bl) {
  const uint32_t mask = (1u << bl) - 1u;
  const uint32_t res = state & mask;
  state >>= bl;
  uint32_t carry = ((res - 1u) | state) & res;
  carry >>= (bl - 1u);
  state += carry;
  return (int32_t)(res - (carry << bl));
}

constexpr int pass_lo(int B, int PL, int q) { return B - (q + 1) * PL > 0 ? B - (q + 1) * PL : 0; }

template <int PL>
struct Blk {
  static constexpr int P = 1 << PL;
  // element held in register r of thread t while index bits [lo, lo+PL) are the in-register ones
  static __device__ __forceinline__ uint32_t elem(uint32_t r, uint32_t t, uint32_t lo) {
    return ((t >> lo) << (lo + PL)) | (r << lo) | (t & ((1u << lo) - 1u));
  }
  // LDS slot of element i for an exchange between passes whose smaller `lo` is key
  static __device__ __forceinline__ uint32_t slot(uint32_t i, uint32_t key) {
    if (key >= SWZ_BITS) return i;
    return i ^ (((i >> (key + PL)) & ((1u << (SWZ_BITS - key)) - 1u)) << key);
  }
};

// twiddle of stage bit b, butterfly index j (< 2^b): W_{2^(b+1)}^j
__device__ __forceinline__ double2 stage_twiddle(bool from_global, const double2* __restrict__ gtw, const double2* ltw,
                                                 uint32_t b, uint32_t j, uint32_t B) {
  if (from_global) return gld<double2>(gtw, (j << (B - 1u - b)) * (uint32_t)sizeof(double2));
  return ltw[((1u << b) - 1u) + j];
}

// The arguments lo, top, B and from_global of the helpers below are compile-time constants at
// every call site (the pass loops are fully unrolled), so all index arithmetic folds.

// In the pass whose in-register bits start at bit 0 the butterfly index is the register index
// alone, and index 0 selects W^0 = (1, -0) of every stage: that product is left out. It returns its
// operand up to the sign of a zero, which no later step can tell apart (sums, products and the
// rounding to the torus treat both zeros alike), so the output words are unchanged.
__device__ __forceinline__ constexpr bool unit_twiddle(uint32_t lo, int r, int s) {
  return lo == 0 && (r & ((1 << s) - 1)) == 0;
}

// forward (decimation in frequency) stages lo+PL-1 ... lo, those above `top` already done
template <int PL>
__device__ __forceinline__ void forward_stages(double2 (&x)[1 << PL], uint32_t lo, uint32_t top, uint32_t B,
                                               uint32_t t, bool from_global, const double2* __restrict__ gtw,
                                               const double2* ltw) {
  constexpr int P = 1 << PL;
  const uint32_t tl = t & ((1u << lo) - 1u);
#pragma unroll
  for (int s = PL - 1; s >= 0; s--) {
    const uint32_t b = lo + (uint32_t)s;
    if (b > top) continue;
#pragma unroll
    for (int r = 0; r < P; r++) {
      if (r & (1 << s)) continue;
      const uint32_t j = ((uint32_t)(r & ((1 << s) - 1)) << lo) | tl;
      const double2 a = x[r], c = x[r | (1 << s)];
      x[r] = cadd(a, c);
      if (unit_twiddle(lo, r, s)) {
        x[r | (1 << s)] = csub(a, c);
        continue;
      }
      const double2 w = stage_twiddle(from_global, gtw, ltw, b, j, B);
      x[r | (1 << s)] = cmul(csub(a, c), w);
    }
    __builtin_amdgcn_sched_barrier(0);  // keeps the twiddle loads of later stages from piling up in registers
  }
}

// inverse (decimation in time, unnormalised) stages lo ... min(lo+PL-1, top)
template <int PL>
__device__ __forceinline__ void inverse_stages(double2 (&x)[1 << PL], uint32_t lo, uint32_t top, uint32_t B,
                                               uint32_t t, bool from_global, const double2* __restrict__ gtw,
                                               const double2* ltw) {
  constexpr int P = 1 << PL;
  const uint32_t tl = t & ((1u << lo) - 1u);
#pragma unroll
  for (int s = 0; s < PL; s++) {
    const uint32_t b = lo + (uint32_t)s;
    if (b > top) continue;
#pragma unroll
    for (int r = 0; r < P; r++) {
      if (r & (1 << s)) continue;
      const uint32_t j = ((uint32_t)(r & ((1 << s) - 1)) << lo) | tl;
      const double2 a = x[r];
      const double2 c = unit_twiddle(lo, r, s) ? x[r | (1 << s)]
                                               : cmul_conj(x[r | (1 << s)], stage_twiddle(from_global, gtw, ltw, b, j, B));
      x[r] = cadd(a, c);
      x[r | (1 << s)] = csub(a, c);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int PL>
__device__ __forceinline__ void exchange(double2 (&x)[1 << PL], double2* buf, uint32_t t, uint32_t lo_from,
                                         uint32_t lo_to) {
  constexpr int P = 1 << PL;
  const uint32_t key = lo_from < lo_to ? lo_from : lo_to;
  // The exchange transposes the register index with the PL thread-index bits just below the larger
  // `lo`. When those are lane bits (larger lo <= 6) the eight threads that trade values sit in one
  // wave, every wave reads back exactly the slots it wrote, those are the slots it read in the
  // exchange before, and no other wave touches them (enumerated for every size: the XOR swizzle
  // moves a value inside its aligned 16-slot block only, and a wave owns whole blocks). LDS
  // operations of a wave execute in order, so such an exchange needs no workgroup barrier at all:
  // two of the four exchanges of a transform at N = 16384, two of three at N = 2048..8192, all of
  // them up to N = 1024.
  const bool crosses_waves = (lo_from > lo_to ? lo_from : lo_to) > 6u;
  if (crosses_waves) __syncthreads();  // every load of the previous exchange has completed
#pragma unroll
  for (int r = 0; r < P; r++) buf[Blk<PL>::slot(Blk<PL>::elem(r, t, lo_from), key)] = x[r];
  if (crosses_waves) __syncthreads();
#pragma unroll
  for (int r = 0; r < P; r++) x[r] = buf[Blk<PL>::slot(Blk<PL>::elem(r, t, lo_to), key)];
}

// waves per SIMD an instance is compiled for: the Fourier sums ((k+1) P complex) and one polynomial
// slice (P complex) stay in registers, plus about 40 registers of addresses and temporaries
constexpr int block_waves(int PL, int KP1, int B = 0) {
  const int need = KP1 * (1 << PL) * 4 + (1 << PL) * 4 + 40;
  const int w = 512 / need;
  if (B >= 12) return BLOCK_BIG_WAVES;
  if (PL >= 3) return 2;  // room to keep a whole phase's loads in flight beats a third wave per SIMD
  return w > 4 ? 4 : (w < 1 ? 1 : w);
}

// largest power of two <= want, within [1, P]
constexpr int group_points(int want, int P) {
  int g = 1;
  while (2 * g <= want && 2 * g <= P) g *= 2;
  return g;
}

// accumulator in LDS: the 32-bit torus sets (6-8 KB per ciphertext). Measured both ways: for the
// 64-bit sets of the same sizes the loss of resident workgroups costs more than the traffic saved.
constexpr bool block_acc_in_lds(int torus_bytes, int B) { return torus_bytes == 4 && B <= 9; }


// ---- exchanges that trade lane bits only: register shuffles instead of an LDS round trip ----
// Transposing one register-index bit with one lane bit moves half of the values to the lane whose
// number differs in that bit. Lane bits 5 and 4 have an instruction for exactly that
// (v_permlane32_swap / v_permlane16_swap, one per register pair); bits 3 and 2 take two row shifts
// whose bank mask writes only the lanes that receive. Bits 1 and 0 have no cheap form and stay in LDS.
// value (register bit S = rho, lane bit PB = lambda) moves to (register bit S = lambda, lane bit PB = rho)
template <int PL, int S, int PB>
__device__ __forceinline__ void transpose_bit(double2 (&x)[1 << PL]) {
#pragma unroll
  for (int r = 0; r < (1 << PL); r++)
    if (!(r & (1 << S))) swap_complex<PB>(x[r], x[r | (1 << S)]);
}
// The exchange between the passes with in-register bits [LO_FROM, LO_FROM + PL) and [LO_TO, LO_TO + PL),
// LO_FROM > LO_TO, for threads whose index bit b sits on physical lane bit b + SHIFT: the top
// D = LO_FROM - LO_TO register bits trade places with thread bits [LO_TO, LO_FROM), then the
// registers are renamed so that the new low bits come first.
template <int PL, int LO_FROM, int LO_TO, int SHIFT, bool INVERSE>
__device__ __forceinline__ void shuffle_exchange(double2 (&x)[1 << PL]) {
  constexpr int D = LO_FROM - LO_TO, P = 1 << PL;
  static_assert(D >= 1 && D <= 3 && D <= PL, "unsupported exchange width");
  if constexpr (INVERSE && D < PL) {
    double2 y[P];
#pragma unroll
    for (int rn = 0; rn < P; rn++) y[((rn & ((1 << D) - 1)) << (PL - D)) | (rn >> D)] = x[rn];
#pragma unroll
    for (int r = 0; r < P; r++) x[r] = y[r];
  }
  if constexpr (D >= 1) transpose_bit<PL, PL - D + 0, LO_TO + 0 + SHIFT>(x);
  if constexpr (D >= 2) transpose_bit<PL, PL - D + 1, LO_TO + 1 + SHIFT>(x);
  if constexpr (D >= 3) transpose_bit<PL, PL - D + 2, LO_TO + 2 + SHIFT>(x);
  if constexpr (!INVERSE && D < PL) {
    double2 y[P];
#pragma unroll
    for (int rn = 0; rn < P; rn++) y[rn] = x[((rn & ((1 << D) - 1)) << (PL - D)) | (rn >> D)];
#pragma unroll
    for (int r = 0; r < P; r++) x[r] = y[r];
  }
}

// Instances whose transforms use the shuffles (measured per instance against the LDS form, same
// results bit for bit): N = 16384 with two levels +6 %, N = 2048 +2 %, N = 256 +4..7 %; N = 512,
// 4096 and 8192 within 1 % either way, N = 1024 and N = 16384 with three levels 2-3 % slower.
constexpr bool block_shuffles(int torus_bytes, int B, int levels) {
  return torus_bytes == 8 && (B == 7 || B == 10 || (B == 13 && levels <= 2));
}

// How the exchanges of a transform are carried out. `rot_pass`: from this pass on (-1: never) a
// thread works under the lane number rotated right by two bits, so that its index bits 0..3 sit on
// physical lane bits 2..5, the ones with a cheap transposition; the last exchange that crosses
// waves (through LDS in any case) hands the values over to the rotated numbering.
constexpr int plan_rot_pass(int B, int PL, bool shuf) {
  if (!shuf) return -1;
  const int Q = (B + PL - 1) / PL;
  for (int q = 1; q < Q; q++)
    if (pass_lo(B, PL, q - 1) <= 6) return (q >= 2 && pass_lo(B, PL, q - 1) == 4) ? q - 1 : -1;
  return -1;
}
// lane-bit offset of exchange q (between passes q-1 and q) when it is done by shuffles, -1 for LDS
constexpr int plan_shift(int B, int PL, int q, bool shuf) {
  if (!shuf) return -1;
  const int rot = plan_rot_pass(B, PL, shuf);
  const int from = pass_lo(B, PL, q - 1), to = pass_lo(B, PL, q);
  if (rot >= 0 && q > rot) return 2;
  if (from <= 6 && to >= 2) return 0;
  return -1;
}
__device__ __forceinline__ uint32_t rotated_lane_index(uint32_t t) {
  return (t & ~63u) | ((t & 63u) >> 2) | ((t & 3u) << 4);
}
// The exchange that hands over to (or back from) the rotated numbering: the in-register bits on
// the rotated side start at bit 4, and there physical lane bits 0, 1 carry element bits 4 + PL and
// 5 + PL while lane bits 2..5 carry element bits 0..3. A 16-byte LDS load serves sixteen lanes per
// cycle from all 64 banks; a 16-byte store serves eight consecutive lanes per cycle into one half
// of the banks, so their eight slots must agree in slot bit 3 (measured: a store whose eight lanes
// straddle both halves takes twice as long). Handing over (natural stores, rotated loads) folds
// the two element bits into slot bits 2, 3; handing back (rotated stores, natural loads) into
// slot bits 1, 2. Both are free of bank conflicts on both sides.
template <int PL, bool BACK>
__device__ __forceinline__ uint32_t rot_slot(uint32_t i) { return i ^ (((i >> (4 + PL)) & 3u) << (BACK ? 1 : 2)); }
template <int PL, bool BACK>
__device__ __forceinline__ void exchange_renumbering(double2 (&x)[1 << PL], double2* buf, uint32_t t_from,
                                                     uint32_t t_to, uint32_t lo_from, uint32_t lo_to) {
  constexpr int P = 1 << PL;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < P; r++) buf[rot_slot<PL, BACK>(Blk<PL>::elem(r, t_from, lo_from))] = x[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < P; r++) x[r] = buf[rot_slot<PL, BACK>(Blk<PL>::elem(r, t_to, lo_to))];
}

template <int B, int PL, bool SHUF, int QI>
__device__ __forceinline__ void forward_pass_chain(double2 (&x)[1 << PL], double2* buf, const double2* ltw,
                                                   const double2* __restrict__ gtw, uint32_t t, uint32_t tr) {
  constexpr int Q = (B + PL - 1) / PL;
  if constexpr (QI < Q) {
    constexpr int rot = plan_rot_pass(B, PL, SHUF);
    if constexpr (QI > 0) {
      constexpr int from = pass_lo(B, PL, QI - 1), to = pass_lo(B, PL, QI);
      constexpr int shift = plan_shift(B, PL, QI, SHUF);
      if constexpr (shift >= 0) shuffle_exchange<PL, from, to, shift, false>(x);
      else if constexpr (QI == rot) exchange_renumbering<PL, false>(x, buf, t, tr, from, to);
      else exchange<PL>(x, buf, t, from, to);
    }
    forward_stages<PL>(x, pass_lo(B, PL, QI), B - 1 - QI * PL, B, (rot >= 0 && QI >= rot) ? tr : t, QI == 0, gtw, ltw);
    forward_pass_chain<B, PL, SHUF, QI + 1>(x, buf, ltw, gtw, t, tr);
  }
}
template <int B, int PL, bool SHUF, int QI>
__device__ __forceinline__ void inverse_pass_chain(double2 (&x)[1 << PL], double2* buf, const double2* ltw,
                                                   const double2* __restrict__ gtw, uint32_t t, uint32_t tr) {
  if constexpr (QI >= 0) {
    constexpr int rot = plan_rot_pass(B, PL, SHUF);
    inverse_stages<PL>(x, pass_lo(B, PL, QI), B - 1 - QI * PL, B, (rot >= 0 && QI >= rot) ? tr : t, QI == 0, gtw, ltw);
    if constexpr (QI > 0) {
      constexpr int from = pass_lo(B, PL, QI - 1), to = pass_lo(B, PL, QI);   // as the forward exchange names them
      constexpr int shift = plan_shift(B, PL, QI, SHUF);
      if constexpr (shift >= 0) shuffle_exchange<PL, from, to, shift, true>(x);
      else if constexpr (QI == rot) exchange_renumbering<PL, true>(x, buf, tr, t, to, from);
      else exchange<PL>(x, buf, t, to, from);
    }
    inverse_pass_chain<B, PL, SHUF, QI - 1>(x, buf, ltw, gtw, t, tr);
  }
}

// in: x[r] = element r*T + t (natural order). out: x[r] = position (t' << PL) | r of the
// bit-reversed spectrum, t' = t or, for the instances that renumber their lanes, rotated_lane_index(t):
// the key conversion, the multiply-accumulate and the inverse transform of an instance agree on it.
// Pass q keeps index bits [lo_q, lo_q + PL) in registers.
template <int B, int PL, bool SHUF>
__device__ __forceinline__ void block_fft_forward(double2 (&x)[1 << PL], double2* buf, const double2* ltw,
                                                  const double2* __restrict__ gtw, uint32_t t) {
  // the thread index is made opaque so that the many per-pass addresses are recomputed here
  // (a few integer operations) instead of being hoisted out of the CMUX loop into registers
  asm volatile("" : "+v"(t));
  forward_pass_chain<B, PL, SHUF, 0>(x, buf, ltw, gtw, t, rotated_lane_index(t));
}

// exact mirror of block_fft_forward: same passes in reverse order
template <int B, int PL, bool SHUF>
__device__ __forceinline__ void block_fft_inverse(double2 (&x)[1 << PL], double2* buf, const double2* ltw,
                                                  const double2* __restrict__ gtw, uint32_t t) {
  constexpr int Q = (B + PL - 1) / PL;
  asm volatile("" : "+v"(t));
  inverse_pass_chain<B, PL, SHUF, Q - 1>(x, buf, ltw, gtw, t, rotated_lane_index(t));
}

// per-stage compact copy of the stage twiddles of every pass but the first: stage b at [2^b - 1, 2^(b+1) - 1)
__device__ __forceinline__ void fill_lds_twiddles(double2* ltw, const double2* __restrict__ gtw, uint32_t B,
                                                  uint32_t stages) {
  for (uint32_t e = threadIdx.x; e + 1 < (1u << stages); e += blockDim.x) {
    const uint32_t b = 31u - __clz(e + 1u);
    const uint32_t j = e + 1u - (1u << b);
    ltw[e] = gtw[(size_t)j << (B - 1u - b)];
  }
}

template <typename T, int B, int PL, bool SHUF>
__global__ __launch_bounds__(1 << (B - PL)) void block_bsk_fourier_kernel(const T* __restrict__ bsk,
                                                                          double2* __restrict__ out,
                                                                          const double2* __restrict__ tables) {
  using S = typename Torus<T>::Signed;
  constexpr int P = 1 << PL;
  extern __shared__ double2 lds[];
  constexpr uint32_t M = 1u << B, N = 2 * M, Tn = 1u << (B - PL);
  const uint32_t t = threadIdx.x;
  double2* buf = lds;
  double2* ltw = lds + M;
  const double2* twist = tables;
  const double2* gtw = tables + M;
  fill_lds_twiddles(ltw, gtw, B, B - PL);
  const T* poly = bsk + (size_t)blockIdx.x * N;
  const double scale = (Torus<T>::BITS == 64) ? 0x1.0p-64 : 0x1.0p-32;
  double2 x[P];
#pragma unroll
  for (int r = 0; r < P; r++) {
    const uint32_t j = (uint32_t)r * Tn + t;
    const double re = signed_to_double((S)poly[j]) * scale;
    const double im = signed_to_double((S)poly[j + M]) * scale;
    x[r] = cmul(make_double2(re, im), twist[j]);
  }
  block_fft_forward<B, PL, SHUF>(x, buf, ltw, gtw, t);
  double2* dst = out + (size_t)blockIdx.x * M;
#pragma unroll
  for (int r = 0; r < P; r++) dst[(uint32_t)r * Tn + t] = x[r];
}

// number of Fourier-sum columns kept in registers; the others live in the workgroup's global
// slab (N >= 8192: two columns of 16 points do not fit 256 registers next to the transform)
constexpr int block_reg_columns(int B, int KP1) { return B >= 12 ? BLOCK_BIG_CREG : KP1; }

template <typename T, int B, int PL, int KP1, int LV>
__global__ __launch_bounds__(1 << (B - PL))
__attribute__((amdgpu_waves_per_eu(block_waves(PL, block_reg_columns(B, KP1), B), block_waves(PL, block_reg_columns(B, KP1), B))))
void block_pbs_kernel(PbsLaunch<T> a) {
  using S = typename Torus<T>::Signed;
  constexpr int P = 1 << PL;
  extern __shared__ double2 lds[];
  constexpr uint32_t M = 1u << B, N = 2 * M, Tn = 1u << (B - PL);
  const uint32_t tid = threadIdx.x;
  uint32_t t = tid;
  const uint32_t n = a.n, bl = a.base_log;
  constexpr uint32_t L = LV;
  constexpr uint32_t k = KP1 - 1;
  constexpr uint32_t G = KP1 * N;
  constexpr uint32_t logN = B + 1;
  constexpr int CREG = block_reg_columns(B, KP1);
  constexpr bool SHUF = block_shuffles((int)sizeof(T), B, LV);
  // points handled between two scheduling fences in the load-heavy phases: bounds the registers
  // the compiler may spend on loads in flight
  // points per load group of each phase: what fits the registers left next to the Fourier sums
  // and the polynomial slice (a point costs 12 registers in the gather, 6 when the parked states
  // are read back, 4 or 8 in the multiply-accumulate, 8 in the update)
  constexpr int FREE0 = 512 / block_waves(PL, CREG, B) - (CREG * P * 4 + P * 4) - (PL >= 4 ? 64 : 40);
  constexpr int FREE = FREE0 < 12 ? 12 : FREE0;
  constexpr int GG = group_points(FREE / 12, P);
  constexpr int GS = group_points(FREE / 6, P);
  constexpr int GM = group_points(FREE / (KP1 > CREG ? 8 : 4), P);
  constexpr int GU = group_points((FREE + (CREG < KP1 ? 0 : 0)) / 8, P);
  double2* buf = lds;
  double2* ltw = lds + M;
  const double2* twist = a.tables;
  const double2* gtw = a.tables + M;
  fill_lds_twiddles(ltw, gtw, B, B - PL);

  constexpr size_t ST_BYTES = L > 1 ? (size_t)G * sizeof(uint32_t) : 0;
  constexpr size_t SLAB = (size_t)G * sizeof(T) + ST_BYTES + (size_t)(KP1 - CREG) * M * sizeof(double2);
  // Boolean sets: the accumulator fits LDS next to the transform image (in the global slab the 16
  // workgroups of a CU cycle 6 MB per XCD through a 4 MB L2 and the fabric carries 5 TB/s); the
  // parked decomposer states stay in the slab, LDS for them would cost two resident workgroups.
  constexpr bool ACC_LDS = block_acc_in_lds((int)sizeof(T), B);
  T* ct0;        // [(k+1)][N]
  uint32_t* st;  // [(k+1)][2P][Tn], decomposer states
  double2* accg = nullptr;  // [KP1-CREG][P][Tn]
  if constexpr (ACC_LDS) {
    static_assert(CREG == KP1, "accumulator in LDS is for the boolean sets only");
    ct0 = (T*)(lds + M + (M >> PL));
    st = (uint32_t*)((char*)a.scratch + (size_t)blockIdx.x * SLAB + (size_t)G * sizeof(T));
  } else {
    char* my = (char*)a.scratch + (size_t)blockIdx.x * SLAB;
    ct0 = (T*)my;
    st = (uint32_t*)(my + (size_t)G * sizeof(T));
    accg = (double2*)(my + (size_t)G * sizeof(T) + ST_BYTES);
  }
  const double inv_m = 1.0 / (double)M;
  // the decomposed field and its rounding bit lie in the upper dword (all 64-bit sets do)
  const bool high_dword = sizeof(T) == 8 && bl * L <= 31u;
  // the state left after the first digit has base_log (levels - 1) + 1 bits: when that is at most
  // 16 (two levels of up to 15 bits) the two states of a point share one parked word
  const bool pack16 = high_dword && L > 1 && bl * (L - 1u) + 1u <= 16u;

  for (size_t ct = blockIdx.x; ct < a.count; ct += gridDim.x) {
    const LweInputRow<T> in = input_row(a, ct);
    const uint32_t lut_id = checked_lut_id(a, ct);
    const T* lut = a.luts + (size_t)lut_id * G;

    // ct0 <- lut * X^{-b_hat}
    const uint32_t b_hat = modswitch<T>(in[n], logN);
    const uint32_t d0 = (2 * N - b_hat) & (2 * N - 1);
    for (uint32_t idx = tid; idx < G; idx += Tn) {
      const uint32_t c = idx / N, j = idx - c * N;
      ct0[idx] = rotated_coeff<T>(lut + (size_t)c * N, j, d0, N);
    }
    __syncthreads();

#ifdef BLOCK_PROFILE
    uint64_t tG = 0, tF = 0, tM = 0, tI = 0, tU = 0, t0, t1;
#define STAMP(acc) t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1;
#else
#define STAMP(acc)
#endif
    for (uint32_t i = 0; i < n; i++) {
      const T ai = in[i];
      if (ai == 0) continue;  // uniform over the workgroup
#ifdef BLOCK_PROFILE
      t0 = __builtin_amdgcn_s_memtime();
#endif
      const uint32_t a_hat = modswitch<T>(ai, logN) & (2 * N - 1);
      double2 accf[CREG][P];
#pragma unroll 1
      for (uint32_t l = L; l >= 1; l--) {
#pragma unroll 1
        for (uint32_t r = 0; r <= k; r++) {
          // opaque copies: the addresses derived from them are recomputed per row (a few integer
          // operations) instead of being hoisted out of the loops and kept live in registers
          uint32_t a_rot = a_hat;
          asm volatile("" : "+v"(t), "+v"(a_rot));
          const T* p = ct0 + (size_t)r * N;
          uint32_t* str = st + (size_t)r * N;
          double2 x[P];
          // Every load-heavy phase is written as: issue all loads of a group, a scheduling fence,
          // then the arithmetic. Left alone the compiler serialises load / wait / use per point
          // when registers are tight, and each wait is a full trip to L2 or HBM.
          if (l == L) {
            // rotated difference of row r, first (finest) digit; the rest of the state is parked
#pragma unroll
            for (int q0 = 0; q0 < P; q0 += GG) {
              T rot0[GG], rot1[GG], own0[GG], own1[GG];
              double2 tw[GG];
#pragma unroll
              for (int u = 0; u < GG; u++) {
                const uint32_t j = (uint32_t)(q0 + u) * Tn + t;
                const uint32_t s0 = (j - a_rot) & (2 * N - 1), s1 = (j + M - a_rot) & (2 * N - 1);
                rot0[u] = gld<T>(p, (s0 & (N - 1)) * (uint32_t)sizeof(T));
                rot1[u] = gld<T>(p, (s1 & (N - 1)) * (uint32_t)sizeof(T));
                own0[u] = gld<T>(p, j * (uint32_t)sizeof(T));
                own1[u] = gld<T>(p, (j + M) * (uint32_t)sizeof(T));
                tw[u] = gld<double2>(twist, j * (uint32_t)sizeof(double2));
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int u = 0; u < GG; u++) {
                const int q = q0 + u;
                const uint32_t j = (uint32_t)q * Tn + t;
                const uint32_t s0 = (j - a_rot) & (2 * N - 1), s1 = (j + M - a_rot) & (2 * N - 1);
                const T x0 = ((s0 & N) ? (T)(T(0) - rot0[u]) : rot0[u]) - own0[u];
                const T x1 = ((s1 & N) ? (T)(T(0) - rot1[u]) : rot1[u]) - own1[u];
                if (high_dword) {
                  // wave-uniform choice: every parameter set of the reference with a 64-bit torus
                  const uint32_t h0 = (uint32_t)((uint64_t)x0 >> 32), h1 = (uint32_t)((uint64_t)x1 >> 32);
                  int32_t g0, g1;
                  if constexpr (L == 1) {
                    g0 = block_digit_single(h0, bl);
                    g1 = block_digit_single(h1, bl);
                  } else {
                    uint32_t st0 = block_state32(h0, bl, L), st1 = block_state32(h1, bl, L);
                    g0 = block_next32(st0, bl);
                    g1 = block_next32(st1, bl);
                    if (pack16) {
                      gst<uint32_t>(str, ((uint32_t)q * Tn + t) * 4u, st0 | (st1 << 16));
                    } else {
                      gst<uint32_t>(str, ((uint32_t)(2 * q) * Tn + t) * 4u, st0);
                      gst<uint32_t>(str, ((uint32_t)(2 * q + 1) * Tn + t) * 4u, st1);
                    }
                  }
                  x[q] = cmul(make_double2((double)g0, (double)g1), tw[u]);
                } else {
                  T st0 = decomp_init_state<T>(x0, bl, L);
                  T st1 = decomp_init_state<T>(x1, bl, L);
                  const S g0 = decomp_next<T>(st0, bl);
                  const S g1 = decomp_next<T>(st1, bl);
                  if constexpr (L > 1) {
                    gst<uint32_t>(str, ((uint32_t)(2 * q) * Tn + t) * 4u, (uint32_t)st0);
                    gst<uint32_t>(str, ((uint32_t)(2 * q + 1) * Tn + t) * 4u, (uint32_t)st1);
                  }
                  x[q] = cmul(make_double2(signed_to_double(g0), signed_to_double(g1)), tw[u]);
                }
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          } else {
#pragma unroll
            for (int q0 = 0; q0 < P; q0 += GS) {
              uint32_t sa[GS], sb[GS];
              double2 tw[GS];
#pragma unroll
              for (int u = 0; u < GS; u++) {
                if (pack16) {
                  sa[u] = gld<uint32_t>(str, ((uint32_t)(q0 + u) * Tn + t) * 4u);
                  sb[u] = 0;
                } else {
                  sa[u] = gld<uint32_t>(str, ((uint32_t)(2 * (q0 + u)) * Tn + t) * 4u);
                  sb[u] = gld<uint32_t>(str, ((uint32_t)(2 * (q0 + u) + 1) * Tn + t) * 4u);
                }
                tw[u] = gld<double2>(twist, ((uint32_t)(q0 + u) * Tn + t) * (uint32_t)sizeof(double2));
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int u = 0; u < GS; u++) {
                const int q = q0 + u;
                // the parked state never exceeds 32 bits (base_log * (level - 1) + 1 bits)
                uint32_t st0 = pack16 ? (sa[u] & 0xFFFFu) : sa[u], st1 = pack16 ? (sa[u] >> 16) : sb[u];
                const int32_t g0 = block_next32(st0, bl);
                const int32_t g1 = block_next32(st1, bl);
                if constexpr (L > 2) {  // with two levels the state is spent after this digit
                  gst<uint32_t>(str, ((uint32_t)(2 * q) * Tn + t) * 4u, st0);
                  gst<uint32_t>(str, ((uint32_t)(2 * q + 1) * Tn + t) * 4u, st1);
                }
                x[q] = cmul(make_double2((double)g0, (double)g1), tw[u]);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          STAMP(tG)
          const double2* rows = a.fbsk + ((((size_t)i * L + (l - 1)) * KP1 + r) * KP1) * M;
          block_fft_forward<B, PL, SHUF>(x, buf, ltw, gtw, t);
          STAMP(tF)
          asm volatile("" : "+v"(t));
          if (l == L && r == 0) {
#pragma unroll
            for (int c = 0; c < KP1; c++) {
#pragma unroll
              for (int q0 = 0; q0 < P; q0 += GM) {
                double2 w[GM];
#pragma unroll
                for (int u = 0; u < GM; u++) w[u] = gld<double2>(rows, ((uint32_t)c * M + (uint32_t)(q0 + u) * Tn + t) * 16u);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < GM; u++) {
                  const double2 v = cmul(w[u], x[q0 + u]);
                  if (c < CREG) accf[c < CREG ? c : 0][q0 + u] = v;
                  else gst<double2>(accg, ((uint32_t)(c - CREG) * M + (uint32_t)(q0 + u) * Tn + t) * 16u, v);
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          } else {
#pragma unroll
            for (int c = 0; c < KP1; c++) {
#pragma unroll
              for (int q0 = 0; q0 < P; q0 += GM) {
                double2 w[GM], old[GM];
#pragma unroll
                for (int u = 0; u < GM; u++) {
                  w[u] = gld<double2>(rows, ((uint32_t)c * M + (uint32_t)(q0 + u) * Tn + t) * 16u);
                  if (c >= CREG) old[u] = gld<double2>(accg, ((uint32_t)(c - CREG) * M + (uint32_t)(q0 + u) * Tn + t) * 16u);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < GM; u++) {
                  if (c < CREG) accf[c < CREG ? c : 0][q0 + u] = cfma(w[u], x[q0 + u], accf[c < CREG ? c : 0][q0 + u]);
                  else gst<double2>(accg, ((uint32_t)(c - CREG) * M + (uint32_t)(q0 + u) * Tn + t) * 16u, cfma(w[u], x[q0 + u], old[u]));
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          STAMP(tM)
        }
      }
      // every gather of this CMUX is done (each thread reads coefficients other threads update)
      __syncthreads();
#pragma unroll
      for (int c = 0; c < KP1; c++) {
        double2 x[P];
        asm volatile("" : "+v"(t));
#pragma unroll
        for (int q = 0; q < P; q++)
          x[q] = (c < CREG) ? accf[c < CREG ? c : 0][q] : gld<double2>(accg, ((uint32_t)(c - CREG) * M + (uint32_t)q * Tn + t) * 16u);
        STAMP(tM)
        block_fft_inverse<B, PL, SHUF>(x, buf, ltw, gtw, t);
        STAMP(tI)
        asm volatile("" : "+v"(t));
        T* p = ct0 + (size_t)c * N;
#pragma unroll
        for (int q0 = 0; q0 < P; q0 += GU) {
          T lo[GU], hi[GU];
          double2 tw[GU];
#pragma unroll
          for (int u = 0; u < GU; u++) {
            const uint32_t j = (uint32_t)(q0 + u) * Tn + t;
            lo[u] = gld<T>(p, j * (uint32_t)sizeof(T));
            hi[u] = gld<T>(p, (j + M) * (uint32_t)sizeof(T));
            tw[u] = gld<double2>(twist, j * (uint32_t)sizeof(double2));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < GU; u++) {
            const uint32_t j = (uint32_t)(q0 + u) * Tn + t;
            const double2 v = cmul_conj(x[q0 + u], tw[u]);
            gst<T>(p, j * (uint32_t)sizeof(T), lo[u] + from_torus_faithful<T>(v.x * inv_m));
            gst<T>(p, (j + M) * (uint32_t)sizeof(T), hi[u] + from_torus_faithful<T>(v.y * inv_m));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
      STAMP(tU)
    }
#ifdef BLOCK_PROFILE
    if ((ct == 0 || ct == 100) && tid == 0)
      printf("ct %lu ticks per CMUX: gather %lu fwd %lu mac %lu inv %lu update %lu\n", (unsigned long)ct, (unsigned long)(tG / n),
             (unsigned long)(tF / n), (unsigned long)(tM / n), (unsigned long)(tI / n), (unsigned long)(tU / n));
#endif

    // sample extraction of coefficient 0
    T* out = a.ct_out + ct * a.out_stride;
    for (uint32_t idx = tid; idx < k * N; idx += Tn) {
      const uint32_t c = idx / N, j = idx - c * N;
      const T* p = ct0 + (size_t)c * N;
      out[idx] = (j == 0) ? p[0] : (T)(T(0) - p[N - j]);
    }
    if (tid == 0) out[(size_t)k * N] = ct0[(size_t)k * N];
    __syncthreads();
  }
}

int block_pl(uint32_t N) { return N >= 8192 ? BLOCK_BIG_PL : (N >= 1024 ? 3 : (N == 512 ? 2 : 1)); }

size_t block_lds_bytes(uint32_t N, uint32_t kp1, int torus_bytes, uint32_t level) {
  const uint32_t M = N / 2;
  size_t bytes = ((size_t)M + (M >> block_pl(N))) * sizeof(double2);
  if (block_acc_in_lds(torus_bytes, (int)ilog2(M)))
    bytes += (size_t)kp1 * N * torus_bytes;
  return bytes;
}

template <typename T, int B, int PL, int KP1, int LV>
int block_launch_levels(const PbsLaunch<T>& a, unsigned grid, hipStream_t stream) {
  const size_t lds = block_lds_bytes(a.N, KP1, (int)sizeof(T), LV);
  if (lds > 64 * 1024)
    CAMD_HIP(hipFuncSetAttribute((const void*)block_pbs_kernel<T, B, PL, KP1, LV>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((block_pbs_kernel<T, B, PL, KP1, LV>), dim3(grid), dim3(1u << (B - PL)), lds, stream, a);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

// LMASK: bit (L-1) set for every level count this (N, k) pair is instantiated for
template <typename T, int B, int PL, int KP1, int LMASK>
int block_launch_instance(const PbsLaunch<T>& a, unsigned grid, hipStream_t stream) {
  if constexpr (LMASK & 1) if (a.level == 1) return block_launch_levels<T, B, PL, KP1, 1>(a, grid, stream);
  if constexpr (LMASK & 2) if (a.level == 2) return block_launch_levels<T, B, PL, KP1, 2>(a, grid, stream);
  if constexpr (LMASK & 4) if (a.level == 3) return block_launch_levels<T, B, PL, KP1, 3>(a, grid, stream);
  return fail(CAMD_EINVAL, "the register-blocked kernel has no instance for this level count");
}

template <typename T, int B, int PL, bool SHUF>
int block_fourier_layout(const T* bsk_std, double2* fbsk, const double2* tables, size_t polys, hipStream_t stream) {
  const size_t lds = block_lds_bytes(2u << B, 0, 8, 1);  // the conversion kernel keeps nothing but the image in LDS
  if (lds > 64 * 1024)
    CAMD_HIP(hipFuncSetAttribute((const void*)block_bsk_fourier_kernel<T, B, PL, SHUF>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((block_bsk_fourier_kernel<T, B, PL, SHUF>), dim3((unsigned)polys), dim3(1u << (B - PL)), lds,
                     stream, bsk_std, fbsk, tables);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}
// the spectrum order of the key is the one the bootstrap instance for this level count works in
template <typename T, int B, int PL>
int block_fourier_instance(const T* bsk_std, double2* fbsk, const double2* tables, size_t polys, uint32_t levels,
                           hipStream_t stream) {
  if (block_shuffles((int)sizeof(T), B, (int)levels))
    return block_fourier_layout<T, B, PL, true>(bsk_std, fbsk, tables, polys, stream);
  return block_fourier_layout<T, B, PL, false>(bsk_std, fbsk, tables, polys, stream);
}

}  // namespace

// instances: the (N, k) pairs of the reference's parameter sets
// points per thread as a power of two, lane renumbering on or off, and the compile-time
// configuration of the N >= 8192 instances all move spectrum values inside the key
uint64_t block_layout_tag(const camd_params& p) {
  uint32_t B = 0;
  while ((2u << B) < p.polynomial_size) B++;
  const uint64_t shuf = block_shuffles((int)(p.torus_bits / 8), (int)B, (int)p.pbs_level) ? 1 : 0;
  constexpr uint64_t REVISION = 2;
  return (REVISION << 32) | ((uint64_t)block_pl(p.polynomial_size) << 16) | (shuf << 8) | ((uint64_t)BLOCK_BIG_PL << 4) |
         (uint64_t)BLOCK_BIG_CREG;
}

bool block_supported(const camd_params& p) {
  const uint32_t N = p.polynomial_size, k = p.glwe_dimension;
  const uint32_t L = p.pbs_level;
  if (p.torus_bits == 64) {
    switch (N) {
      case 256: return (k == 5 || k == 6) && L == 1;
      case 512: return k == 3 && L == 1;
      case 1024: return k == 2 && L == 1;
      case 2048: return k == 1 && L == 1;
      case 4096: case 8192: return k == 1 && (L == 1 || L == 2);
      case 16384: return k == 1 && (L == 2 || L == 3);
      default: return false;
    }
  }
  return (N == 512 && k == 2 && L == 2) || (N == 1024 && k == 1 && L == 3);
}

size_t block_scratch_bytes(const camd_params& p, unsigned* grid_out, size_t count) {
  const uint32_t N = p.polynomial_size, kp1 = p.glwe_dimension + 1;
  const int Bits = (int)ilog2(N / 2);
  const int creg = block_reg_columns(Bits, (int)kp1);
  const size_t slab = (size_t)kp1 * N * (p.torus_bits / 8) + (p.pbs_level > 1 ? (size_t)kp1 * N * sizeof(uint32_t) : 0) +
                      (size_t)(kp1 - creg) * (N / 2) * sizeof(double2);
  const size_t threads = (N / 2) >> block_pl(N);
  size_t per_cu = (size_t)block_waves(block_pl(N), creg, Bits) * 256 / threads;  // register budget of the instance
  const size_t by_lds = (160 * 1024) / block_lds_bytes(N, kp1, (int)(p.torus_bits / 8), p.pbs_level);
  if (per_cu > by_lds) per_cu = by_lds;
  if (per_cu < 1) per_cu = 1;
  size_t grid = 256 * per_cu;
  if (grid > count) grid = count;
  if (grid < 1) grid = 1;
  *grid_out = (unsigned)grid;
  return slab * grid;
}

template <typename T>
int block_bsk_to_fourier(const T* bsk_std, double2* fbsk, const double2* tables, uint32_t N, size_t polys,
                         uint32_t levels, hipStream_t stream) {
  switch (N) {
    case 256: return block_fourier_instance<T, 7, 1>(bsk_std, fbsk, tables, polys, levels, stream);
    case 512: return block_fourier_instance<T, 8, 2>(bsk_std, fbsk, tables, polys, levels, stream);
    case 1024: return block_fourier_instance<T, 9, 3>(bsk_std, fbsk, tables, polys, levels, stream);
    case 2048: return block_fourier_instance<T, 10, 3>(bsk_std, fbsk, tables, polys, levels, stream);
    case 4096: return block_fourier_instance<T, 11, 3>(bsk_std, fbsk, tables, polys, levels, stream);
    case 8192: return block_fourier_instance<T, 12, BLOCK_BIG_PL>(bsk_std, fbsk, tables, polys, levels, stream);
    case 16384: return block_fourier_instance<T, 13, BLOCK_BIG_PL>(bsk_std, fbsk, tables, polys, levels, stream);
  }
  return fail(CAMD_EINVAL, "the register-blocked kernel has no instance for this polynomial size");
}

template <typename T>
int block_launch_pbs(const PbsLaunch<T>& a, unsigned grid, hipStream_t stream) {
  if (a.count == 0) return CAMD_OK;
  const uint32_t N = a.N, kp1 = a.k + 1;
  if constexpr (sizeof(T) == 8) {
    if (N == 256 && kp1 == 6) return block_launch_instance<T, 7, 1, 6, 1>(a, grid, stream);
    if (N == 256 && kp1 == 7) return block_launch_instance<T, 7, 1, 7, 1>(a, grid, stream);
    if (N == 512 && kp1 == 4) return block_launch_instance<T, 8, 2, 4, 1>(a, grid, stream);
    if (N == 1024 && kp1 == 3) return block_launch_instance<T, 9, 3, 3, 1>(a, grid, stream);
    if (N == 2048 && kp1 == 2) return block_launch_instance<T, 10, 3, 2, 1>(a, grid, stream);
    if (N == 4096 && kp1 == 2) return block_launch_instance<T, 11, 3, 2, 3>(a, grid, stream);
    if (N == 8192 && kp1 == 2) return block_launch_instance<T, 12, BLOCK_BIG_PL, 2, 3>(a, grid, stream);
    if (N == 16384 && kp1 == 2) return block_launch_instance<T, 13, BLOCK_BIG_PL, 2, 6>(a, grid, stream);
  } else {
    if (N == 512 && kp1 == 3) return block_launch_instance<T, 8, 2, 3, 2>(a, grid, stream);
    if (N == 1024 && kp1 == 2) return block_launch_instance<T, 9, 3, 2, 4>(a, grid, stream);
  }
  return fail(CAMD_EINVAL, "the register-blocked kernel has no instance for this parameter set");
}

template int block_bsk_to_fourier<uint64_t>(const uint64_t*, double2*, const double2*, uint32_t, size_t, uint32_t,
                                             hipStream_t);
template int block_bsk_to_fourier<uint32_t>(const uint32_t*, double2*, const double2*, uint32_t, size_t, uint32_t,
                                             hipStream_t);
template int block_launch_pbs<uint64_t>(const PbsLaunch<uint64_t>&, unsigned, hipStream_t);
template int block_launch_pbs<uint32_t>(const PbsLaunch<uint32_t>&, unsigned, hipStream_t);

}  // namespace camd
