// Programmable bootstrap, wave-per-ciphertext family (N = 1024): the schedule with TWO waves per
// SIMD.
//
// Replaces FftEngine::discard_bootstrap_lwe_ciphertext as the reference calls it at
// concrete-shortint/src/engine/server_side/mod.rs:148-153, for the PARAM_MESSAGE_*_CARRY_* sets
// with N = 1024, k = 2 and one decomposition level, like wave_pbs_kernel_v2 in pbs_wave.hip.
//
// Why a second schedule. A wave that is alone on its SIMD issues one instruction every four
// cycles whatever the instruction is, so the integer, conversion and LDS instructions of a CMUX
// each cost as much as an f64 fused multiply-add, and nothing runs while the wave waits. With two
// waves on a SIMD the f64 pipe takes one wave's arithmetic while the other issues its LDS
// traffic, its 32-bit instructions (half the pipe time of an f64 one) or waits for key values.
// Eight ciphertexts have to fit one CU for that, which is what the layout below is sized for:
//   * The accumulator is kept to 32 bits per coefficient in LDS (12 KB per ciphertext instead of
//     24): the high dwords of the torus words. The gadget digits already come from the high
//     dwords only; each CMUX's increment is rounded at 2^-32 of the torus, a variance of
//     2^-64/12 per coefficient and CMUX against the 2^-60 or so that the f64 transforms leave,
//     and five orders of magnitude below the output noise of the parameter set. The low dwords
//     of the output ciphertext are zero; look-up tables are used to 2^-32.
//   * The lane exchanges of a transform go through ONE padded plane (4.5 KB per wave), the real
//     parts first, then the imaginary parts: LDS operations of a wave execute in order, so the
//     second use of the plane cannot overtake the first.
//   * No key row is parked in registers (a wave has 256 of them here, not 512). The 72 KiB of a
//     CMUX's GGSW come from the L2 during its multiply-accumulate, two of the eight frequency
//     groups ahead of the products that use them, and the Fourier sums overwrite the spectra
//     they were computed from: 96 registers of state instead of 192. Waiting for them is what
//     the second wave of the SIMD is there to cover.
//   * 512-thread workgroups, one per CU (19 KB of folded twiddle constants + 8 x 16.5 KB), eight
//     independent ciphertexts; persistent waves with the per-XCD rendezvous of the other
//     schedules, so that a key row is fetched into an XCD's L2 about once per round.
#include "lane_swap.hpp"
#include "wave1024.hpp"

namespace camd {

namespace {

constexpr int W8_WAVES = 8;   // ciphertexts per workgroup: two waves per SIMD

template <int K>
constexpr size_t w8_slab_bytes() {
  return (size_t)(K + 1) * 1024 * sizeof(uint32_t) + PLANE * sizeof(double);
}
template <int K>
constexpr size_t w8_lds_bytes() {
  return TWS_COUNT * sizeof(double2) + W8_WAVES * w8_slab_bytes<K>() + W8_WAVES * sizeof(uint32_t);   // + progress words
}

// The exchange of a transform that trades the register index with lane bits 5:3, in registers:
// x[q] of lane (h, lo) becomes what x[h] of lane (q, lo) was. One v_permlane32_swap per register
// pair and dword for lane bit 5, one v_permlane16_swap for bit 4, row shifts for bit 3: about 80
// vector instructions (32-bit moves, half the pipe time of an f64 instruction, issued beside the
// other wave of the SIMD) instead of 16 LDS stores and 16 LDS loads on a pipe the whole CU shares.
__device__ __forceinline__ void transpose_lane_bits_5_3(double2 (&x)[8]) {
#ifdef W8_TIMING_NO_SHUFFLES   // timing-only build: wrong results
  return;
#endif
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

// ---- the transforms of wave1024.hpp with ONE exchange plane ----
// Per stream and exchange: real parts out, real parts back, imaginary parts out, imaginary parts
// back. The values read back first wait in registers that the real parts just left.
template <int S, typename Hook = NoHook>
__device__ __forceinline__ void fft512_forward_multi1(double2 (&x)[S][8], int lane, const double2* tab, double* pl,
                                                      Hook hook = Hook()) {
  const int hi = lane >> 3, lo = lane & 7;
  {
    const Tw8 k = twist_tw8();
#pragma unroll
    for (int s = 0; s < S; s++) dft8_tw<false, false>(x[s], k);
  }
  {
    const Tw8 k = load_tw8<true>(tab + TWS_A + lane, 64);
#pragma unroll
    for (int s = 0; s < S; s++) transpose_lane_bits_5_3(x[s]);
#pragma unroll
    for (int s = 0; s < S; s++) dft8_tw<false, true>(x[s], k);
  }
  {
    const Tw8 k = load_tw8<false>(tab + TWS_B + lo, 8);
#pragma unroll
    for (int s = 0; s < S; s++) {
#ifdef W8_TIMING_NO_LDS_EXCHANGE   // timing-only build: wrong results
      continue;
#endif
      double re[8];
#pragma unroll
      for (int b = 0; b < 8; b++) pl[hi * ROW + b * 8 + (lo ^ b)] = x[s][b].x;
#pragma unroll
      for (int q = 0; q < 8; q++) re[q] = pl[hi * ROW + lo * 8 + (q ^ lo)];
#pragma unroll
      for (int b = 0; b < 8; b++) pl[hi * ROW + b * 8 + (lo ^ b)] = x[s][b].y;
#pragma unroll
      for (int q = 0; q < 8; q++) x[s][q] = make_double2(re[q], pl[hi * ROW + lo * 8 + (q ^ lo)]);
    }
    hook(0);
#pragma unroll
    for (int s = 0; s < S; s++) dft8_tw<false, false>(x[s], k);
  }
}

template <int S>
__device__ __forceinline__ void fft512_inverse_multi1(double2 (&x)[S][8], int lane, const double2* tab, double* pl) {
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int s = 0; s < S; s++) dft8<true>(x[s]);
  {
    const Tw8 k = load_tw8<false>(tab + TWS_B + lo, 8);
#pragma unroll
    for (int s = 0; s < S; s++) {
#ifdef W8_TIMING_NO_LDS_EXCHANGE   // timing-only build: wrong results
      continue;
#endif
      double re[8];
#pragma unroll
      for (int q = 0; q < 8; q++) pl[hi * ROW + lo * 8 + (q ^ lo)] = x[s][q].x;
#pragma unroll
      for (int b = 0; b < 8; b++) re[b] = pl[hi * ROW + b * 8 + (lo ^ b)];
#pragma unroll
      for (int q = 0; q < 8; q++) pl[hi * ROW + lo * 8 + (q ^ lo)] = x[s][q].y;
#pragma unroll
      for (int b = 0; b < 8; b++) x[s][b] = make_double2(re[b], pl[hi * ROW + b * 8 + (lo ^ b)]);
    }
#pragma unroll
    for (int s = 0; s < S; s++) dft8_tw<true, false>(x[s], k);
  }
  {
    const Tw8 k = load_tw8<true>(tab + TWS_C + lane, 64);
#pragma unroll
    for (int s = 0; s < S; s++) transpose_lane_bits_5_3(x[s]);
#pragma unroll
    for (int s = 0; s < S; s++) dft8_tw<true, true>(x[s], k);
  }
}

template <typename T, int K, int BL>
__global__ __launch_bounds__(512, 2) void wave8_pbs_kernel(PbsLaunch<T> a) {
  static_assert(sizeof(T) == 8 && BL <= 31, "64-bit torus whose digit lies in the high dword");
  constexpr int N = 1024, M = 512, LOGN = 10, R = K + 1, NS = R * R;
  extern __shared__ __align__(16) char smem[];
  double2* tab = (double2*)smem;
  stage_folded_tables(tab, a.tables, 512);
  __syncthreads();

  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* slab = smem + TWS_COUNT * sizeof(double2) + (size_t)wave * w8_slab_bytes<K>();
  uint32_t* ct0 = (uint32_t*)slab;                                   // [R][N] high dwords of the accumulator
  double* pl = (double*)(slab + (size_t)R * N * sizeof(uint32_t));   // the exchange plane
  int lane = lane0;
  // Progress of each wave (iterations done in its current round), for the priority rule below.
  // Waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the four SIMDs in turn).
#ifdef W8_PROGRESS_FLAT
  volatile uint32_t* progress = (volatile uint32_t*)(smem + TWS_COUNT * sizeof(double2) + W8_WAVES * w8_slab_bytes<K>());
#else
  uint32_t* progress = (uint32_t*)(smem + TWS_COUNT * sizeof(double2) + W8_WAVES * w8_slab_bytes<K>());
#endif

  // the Fourier key through a buffer descriptor (at most 4 GiB: the N = 1024 sets take 57 MB)
  const __amdgpu_buffer_rsrc_t key_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.fbsk, 0, (int)((size_t)a.n * NS * M * sizeof(double2)), 0x00020000);
  XcdRendezvous rendezvous;
  if (a.scratch) rendezvous.enrol(a.scratch, lane);
  const size_t per_round = (size_t)gridDim.x * W8_WAVES;
  uint32_t round = 0;
  for (size_t ct = (size_t)blockIdx.x * W8_WAVES + wave; ct < a.count; ct += per_round, round++) {
    const T* in = a.ct_in + ct * a.in_stride;
    const uint32_t n = a.n;
    const uint32_t lut_id = checked_lut_id(a, ct);
    const T* lut = a.luts + (size_t)lut_id * (R * N);
    {
      // ct0 <- lut * X^{-b_hat}, high dwords
      const uint32_t b_hat = modswitch<T>(in[n], LOGN);
      const uint32_t d0 = (2 * N - b_hat) & (2 * N - 1);
      for (int idx = lane; idx < R * N; idx += 64) {
        const int c = idx >> LOGN, j = idx & (N - 1);
        ct0[idx] = (uint32_t)(rotated_coeff<T>(lut + (size_t)c * N, j, d0, N) >> 32);
      }
    }

    if (lane == 0) progress[wave] = 0;
    // iterations whose mask element is zero are skipped, as the reference skips them
    uint32_t i = 0;
    T a_cur = wave_uniform(in[0]);
    while (i < n && a_cur == 0) {
      i++;
      a_cur = wave_uniform(in[i]);   // in[n] is the body: readable
    }

#ifdef CAMD_PROFILE
    const uint64_t c_begin = __builtin_amdgcn_s_memtime(), r_begin = __builtin_amdgcn_s_memrealtime();
    uint64_t tG = 0, tF = 0, tM = 0, tI = 0, tR = 0, t0, t1;
#define STAMP(acc) t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1;
#else
#define STAMP(acc)
#endif
    while (i < n) {
#ifdef CAMD_PROFILE
      t0 = __builtin_amdgcn_s_memtime();
#endif
      // The lane number is made opaque once per CMUX: every address that depends on it (some
      // forty LDS and key offsets) is then recomputed here, a few integer instructions each,
      // instead of being hoisted out of the loop into registers the loop does not have.
      lane = lane0;
      asm volatile("" : "+v"(lane));
      // The SIMD's arbiter favours the older of its two waves on every contended issue slot, so
      // left alone one ciphertext of each pair finishes a third earlier than the other and the
      // SIMD then runs a lone wave until the round ends. Each wave publishes its iteration count
      // and takes the higher priority while it is behind its partner: the pair stays in step.
      {
        // plain LDS accesses between compiler fences (a volatile pointer would lose its address
        // space and turn them into flat accesses that drain every counter)
        asm volatile("" ::: "memory");
        if (lane == 0) progress[wave] = i;
        const uint32_t other = __builtin_amdgcn_readfirstlane(progress[wave ^ 4]);
        asm volatile("" ::: "memory");
        if (other > i)
          __builtin_amdgcn_s_setprio(2);
        else
          __builtin_amdgcn_s_setprio(0);
      }
      const uint32_t a_hat = modswitch<T>(a_cur, LOGN) & (2 * N - 1);
      // this iteration's GGSW, [row][col][register g][lane]: byte offset of its first half-slab group
      const uint32_t key_off = i * (uint32_t)(NS * M * sizeof(double2));
      i++;

      // gather rotated - own (high dwords), one signed digit each
      double2 x[R][8];
      {
#ifndef W8_GATHER_AHEAD
#define W8_GATHER_AHEAD 1
#endif
        constexpr int GA = W8_GATHER_AHEAD;   // units of half a polynomial requested ahead of the arithmetic
        uint32_t own0[GA + 1][4], own1[GA + 1][4], rot0[GA + 1][4], rot1[GA + 1][4];
        auto issue = [&](int u) {
          const uint32_t* p = ct0 + (u >> 1) * N;
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const uint32_t j0 = 64 * (4 * (u & 1) + t) + lane, j1 = j0 + M;
            own0[u % (GA + 1)][t] = p[j0];
            own1[u % (GA + 1)][t] = p[j1];
            rot0[u % (GA + 1)][t] = p[(j0 - a_hat) & (N - 1)];
            rot1[u % (GA + 1)][t] = p[(j1 - a_hat) & (N - 1)];
          }
        };
        auto consume = [&](int u) {
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const int q = 4 * (u & 1) + t;
            const uint32_t j0 = 64 * q + lane, j1 = j0 + M;
            const int32_t g0 = digit_of_rotated_difference32<BL, LOGN>(rot0[u % (GA + 1)][t], own0[u % (GA + 1)][t], j0 - a_hat);
            const int32_t g1 = digit_of_rotated_difference32<BL, LOGN>(rot1[u % (GA + 1)][t], own1[u % (GA + 1)][t], j1 - a_hat);
            x[u >> 1][q] = make_double2((double)g0, (double)g1);   // the twist is in the transform
          }
        };
#pragma unroll
        for (int u = 0; u < GA; u++) issue(u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2 * R; u++) {
          if (u + GA < 2 * R) issue(u + GA);
          __builtin_amdgcn_sched_barrier(0);
          consume(u);
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      STAMP(tG)
      // key values of frequency group g: NS loads of 16 bytes per lane (1 KiB per wave each)
      double2 kq[3][NS];
      // (buffer loads: one descriptor for the whole key in scalar registers, the slab in the
      // per-lane offset, the iteration and the half of the slab in the scalar offset, the group
      // in the instruction's immediate: no 64-bit address registers)
      auto load_group = [&](int g) {
#pragma unroll
        for (int sl = 0; sl < NS; sl++) {
          const auto w = __builtin_amdgcn_raw_buffer_load_b128(
              key_rsrc, (int)((uint32_t)lane * 16u + (g & 3) * 1024),
              (int)(key_off + (uint32_t)sl * (uint32_t)(M * sizeof(double2)) + (g >> 2) * 4096), 0);
          kq[g % 3][sl] = as_double2(__builtin_bit_cast(v4u, w));
        }
      };
      // groups 0 and 1 are requested ahead of the last forward pass, whose butterflies cover them
      fft512_forward_multi1<R>(x, lane, tab, pl, [&](int) { load_group(0); });
      __builtin_amdgcn_sched_barrier(0);
      load_group(1);
      __builtin_amdgcn_sched_barrier(0);

      STAMP(tF)
      // x[c][g] <- sum_r GGSW[r][c][g] * x[r][g]: the sums of a group replace its spectra
#pragma unroll
      for (int g = 0; g < 8; g++) {
        if (g + 2 < 8) load_group(g + 2);
        __builtin_amdgcn_sched_barrier(0);
        double2 v[R];
#pragma unroll
        for (int c = 0; c < R; c++) {
          v[c] = cmul(kq[g % 3][c], x[0][g]);
#pragma unroll
          for (int r = 1; r < R; r++) v[c] = cfma(kq[g % 3][r * R + c], x[r][g], v[c]);
        }
#pragma unroll
        for (int c = 0; c < R; c++) x[c][g] = v[c];
        __builtin_amdgcn_sched_barrier(0);
      }

      STAMP(tM)
      // the next mask element is requested here, where registers are free again, and looked at
      // after the rounding (requested at the top of the CMUX it would have to be waited for and
      // spilled at once: it would be live across the multiply-accumulate)
      const T a_nxt = in[i];
      __builtin_amdgcn_sched_barrier(0);
      fft512_inverse_multi1<R>(x, lane, tab, pl);

      STAMP(tI)
      // untwist, round to 2^-32 of the torus, add into the accumulator
      {
#ifndef W8_ROUND_AHEAD
#define W8_ROUND_AHEAD 1
#endif
        constexpr int RA = W8_ROUND_AHEAD;   // columns whose old values are requested ahead
        uint32_t old0[RA + 1][8], old1[RA + 1][8];
        auto issue_old = [&](int c) {
          const uint32_t* pc = ct0 + c * N;
#pragma unroll
          for (int q = 0; q < 8; q++) {
            old0[c % (RA + 1)][q] = pc[64 * q + lane];
            old1[c % (RA + 1)][q] = pc[64 * q + lane + M];
          }
        };
#pragma unroll
        for (int c = 0; c < RA && c < R; c++) issue_old(c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < R; c++) {
          if (c + RA < R) issue_old(c + RA);
          __builtin_amdgcn_sched_barrier(0);
          uint32_t* pc = ct0 + c * N;
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const double2 v = untwist8(x[c][q], q);
            pc[64 * q + lane] = torus_accumulate(old0[c % (RA + 1)][q], v.x);
            pc[64 * q + lane + M] = torus_accumulate(old1[c % (RA + 1)][q], v.y);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      STAMP(tR)
      // next active iteration
      a_cur = wave_uniform(a_nxt);
      while (i < n && a_cur == 0) {
        i++;
        a_cur = wave_uniform(in[i]);
      }
    }

    __builtin_amdgcn_s_setprio(0);
    if (lane == 0) progress[wave] = n;   // done with this ciphertext: the partner keeps the higher priority until it is, too
#ifdef CAMD_PROFILE
    if ((ct == 0 || ct == 4 || ct == 5001 || ct == 5005 || ct == 9003) && lane == 0) {
      const uint64_t dc = __builtin_amdgcn_s_memtime() - c_begin, dr = __builtin_amdgcn_s_memrealtime() - r_begin;
      printf("ct %lu cycles per iteration: gather %lu fwd %lu mac %lu inv %lu round %lu | total memtime %lu realtime(100MHz) %lu\n",
             (unsigned long)ct, (unsigned long)(tG / n), (unsigned long)(tF / n), (unsigned long)(tM / n), (unsigned long)(tI / n),
             (unsigned long)(tR / n), (unsigned long)dc, (unsigned long)dr);
    }
#endif
    // sample extraction of coefficient 0; the low dwords are zero
    T* out = a.ct_out + ct * a.out_stride;
    for (int idx = lane; idx < K * N; idx += 64) {
      const int c = idx >> LOGN, j = idx & (N - 1);
      const uint32_t* p = ct0 + c * N;
      const uint32_t w = (j == 0) ? p[0] : (uint32_t)(0u - p[N - j]);
      out[idx] = (T)w << 32;
    }
    if (lane == 0) out[K * N] = (T)ct0[K * N] << 32;
    if ((size_t)(round + 1) * per_round < a.count) rendezvous.arrive_and_wait(round, lane);
  }
}

}  // namespace

template <typename T>
int wave8_launch_pbs(const PbsLaunch<T>& a, hipStream_t stream) {
  if constexpr (sizeof(T) == 8) {
    if (a.N == 1024 && a.k == 2 && a.level == 1 && a.base_log == 23) {
      constexpr size_t lds = w8_lds_bytes<2>();
      static_assert(lds <= 160 * 1024, "LDS budget exceeded");
      auto kern = wave8_pbs_kernel<T, 2, 23>;
      CAMD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int dev = 0, cus = 0;
      CAMD_HIP(hipGetDevice(&dev));
      CAMD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      const unsigned grid = (unsigned)((a.count + W8_WAVES - 1) / W8_WAVES);
      const unsigned persistent = cus > 0 && grid > (unsigned)cus ? (unsigned)cus : grid;   // LDS allows one workgroup per CU
      if (persistent < grid) {
        if (!a.scratch) return fail(CAMD_EINVAL, "wave kernel: arrival counters were not reserved");
        CAMD_HIP(hipMemsetAsync(a.scratch, 0, RENDEZVOUS_BYTES, stream));
      }
      hipLaunchKernelGGL(kern, dim3(persistent), dim3(64 * W8_WAVES), lds, stream, a);
      CAMD_HIP(hipGetLastError());
      return CAMD_OK;
    }
  }
  return fail(CAMD_EINVAL, "wave kernel (two waves per SIMD): no instance for this parameter set");
}

bool wave8_supported(uint32_t torus_bits, uint32_t N, uint32_t k, uint32_t level, uint32_t base_log) {
  return torus_bits == 64 && N == 1024 && k == 2 && level == 1 && base_log == 23;
}

template int wave8_launch_pbs<uint64_t>(const PbsLaunch<uint64_t>&, hipStream_t);
template int wave8_launch_pbs<uint32_t>(const PbsLaunch<uint32_t>&, hipStream_t);

}  // namespace camd
