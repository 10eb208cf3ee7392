// Programmable bootstrap, wave-per-ciphertext family (N = 1024).
//
// Replaces FftEngine::discard_bootstrap_lwe_ciphertext as the reference calls it at
// concrete-shortint/src/engine/server_side/mod.rs:148-153 (see pbs_generic.hip for the list of
// steps). This is the throughput kernel for the PARAM_MESSAGE_*_CARRY_* sets with N = 1024.
//
// Design for CDNA4
//   * ONE 64-lane wave owns one ciphertext for all n CMUX iterations; a 256-thread workgroup
//     is four independent ciphertexts, one per SIMD. A wave never waits for another wave:
//     the only workgroup barrier is after the twiddle tables are staged into LDS.
//   * The accumulator (k+1 polynomials) lives in that wave's LDS slab; rotation by X^a is an
//     index/sign permutation folded into the LDS read that feeds the decomposition.
//   * The N/2-point complex FFT is 8 x 8 x 8: each lane holds 8 points in registers, does a
//     radix-8 butterfly, and exchanges with the other lanes through a padded, bank-conflict-
//     free LDS image (two exchanges per transform). Forward is decimation in frequency,
//     inverse is decimation in time on the permuted spectrum, so no reordering pass exists;
//     the Fourier bootstrap key is stored in exactly the (register, lane) order the
//     multiply-accumulate consumes, which makes every key load a coalesced 1 KiB access.
//   * The 1/(N/2) of the inverse transform is folded into the key at conversion time.
//   * The twist of the negacyclic transform is split into eight constants and a per-lane factor
//     that rides on the first pass's twiddles: no twist table is read.
//   * f64 -> torus accumulation is four instructions (wave_common.hpp), the single-level digit of
//     the rotated difference four more, on the high dwords only.
//   * The default kernel (wave_pbs_kernel_v2, one decomposition level) runs the k+1 transforms of a
//     CMUX interleaved and keeps the key rows of the next CMUX in accumulation registers; the
//     first schedule (wave_pbs_kernel, any level count) and the k+1-waves-per-ciphertext variant
//     for small batches (wave_pbs_kernel_lat) are in this file too. The N = 512 instances are in
//     pbs_wave512.hip.
#include "wave1024.hpp"

namespace camd {

namespace {

template <typename T, int K>
constexpr size_t wave_slab_bytes() {
  return (size_t)(K + 1) * 1024 * sizeof(T) + 2 * PLANE * sizeof(double);
}
template <typename T, int K>
constexpr size_t wave_lds_bytes() {
  return TWS_COUNT * sizeof(double2) + WAVES * wave_slab_bytes<T, K>();
}

// ---- key conversion: one wave per polynomial, same transform, key pre-scaled by 1/512 ----
template <typename T>
__global__ __launch_bounds__(256) void wave_bsk_fourier_kernel(const T* __restrict__ bsk, double2* __restrict__ out,
                                                               const double2* __restrict__ tables, size_t polys) {
  using S = typename Torus<T>::Signed;
  extern __shared__ __align__(16) char smem[];
  double2* tab = (double2*)smem;
  for (int t = threadIdx.x; t < TAB_COUNT; t += 256) tab[t] = tables[t];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t q = (size_t)blockIdx.x * WAVES + wave;
  if (q >= polys) return;
  double* sre = (double*)(smem + TAB_COUNT * sizeof(double2)) + (size_t)wave * 2 * PLANE;
  double* sim = sre + PLANE;
  const T* poly = bsk + q * 1024;
  const double scale = (Torus<T>::BITS == 64) ? 0x1.0p-64 : 0x1.0p-32;
  double2 x[8];
#pragma unroll
  for (int a = 0; a < 8; a++) {
    const int j = 64 * a + lane;
    const double re = signed_to_double((S)poly[j]) * scale;
    const double im = signed_to_double((S)poly[j + 512]) * scale;
    x[a] = cmul(make_double2(re, im), tab[TAB_TWIST + j]);
  }
  fft512_forward(x, lane, tab, sre, sim);
  double2* dst = out + q * 512;
#pragma unroll
  for (int g = 0; g < 8; g++) dst[g * 64 + lane] = make_double2(x[g].x * (1.0 / 512.0), x[g].y * (1.0 / 512.0));
}

// ---- the bootstrap ----
template <typename T, int K, int L, int BL>
__global__ __launch_bounds__(256, 1) void wave_pbs_kernel(PbsLaunch<T> a) {
  using S = typename Torus<T>::Signed;
  constexpr int N = 1024, M = 512, LOGN = 10;
  extern __shared__ __align__(16) char smem[];
  double2* tab = (double2*)smem;
  stage_folded_tables(tab, a.tables, 256);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t ct = (size_t)blockIdx.x * WAVES + wave;
  if (ct >= a.count) return;

  char* slab = smem + TWS_COUNT * sizeof(double2) + (size_t)wave * wave_slab_bytes<T, K>();
  T* ct0 = (T*)slab;
  double* sre = (double*)(slab + (size_t)(K + 1) * N * sizeof(T));
  double* sim = sre + PLANE;

  const T* in = a.ct_in + ct * a.in_stride;
  const uint32_t n = a.n;
  const uint32_t lut_id = checked_lut_id(a, ct);
  const T* lut = a.luts + (size_t)lut_id * ((K + 1) * N);

  // ct0 <- lut * X^{-b_hat}
  {
    const uint32_t b_hat = modswitch<T>(in[n], LOGN);
    const uint32_t d0 = (2 * N - b_hat) & (2 * N - 1);
    for (int idx = lane; idx < (K + 1) * N; idx += 64) {
      const int c = idx >> LOGN, j = idx & (N - 1);
      ct0[idx] = rotated_coeff<T>(lut + (size_t)c * N, j, d0, N);
    }
  }

  // The (iteration, row) nest is flattened into one loop of steps so that the key rows of the
  // next step are loaded at a single site: they are issued right after the multiply-accumulate
  // of the current step and land while the inverse transforms / the next decomposition and
  // forward transform run. Iterations whose mask element is zero are skipped, as the reference
  // does; their rows are never loaded.
  double2 bq[K + 1][8];
  auto load_rows = [&](uint32_t it, int r, int l) {
    const double2* brow = a.fbsk + ((((size_t)it * L + (l - 1)) * (K + 1) + r) * (K + 1)) * M + lane;
#pragma unroll
    for (int c = 0; c <= K; c++)
#pragma unroll
      for (int g = 0; g < 8; g++) {
        bq[c][g] = brow[(size_t)c * M + g * 64];
      }
  };
  double2 acc[K + 1][8];
#pragma unroll
  for (int c = 0; c <= K; c++)
#pragma unroll
    for (int g = 0; g < 8; g++) {
      acc[c][g] = make_double2(0.0, 0.0);
      bq[c][g] = make_double2(0.0, 0.0);
    }

  uint32_t it = 0;
  int r = 0;
  T a_cur = wave_uniform(in[0]);
  T a_nxt = in[n > 1 ? 1 : n];
  bool run = (a_cur != 0);
  if (run) load_rows(0, 0, L);

  while (it < n) {
    if (run) {
      const uint32_t a_hat = modswitch<T>(a_cur, LOGN) & (2 * N - 1);
      // gather rotated - own, start the decomposition
      const T* p = ct0 + r * N;
      uint32_t st0[8], st1[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t j0 = 64 * q + lane, j1 = j0 + M;
        const T x0 = rotated_coeff<T>(p, j0, a_hat, N) - p[j0];
        const T x1 = rotated_coeff<T>(p, j1, a_hat, N) - p[j1];
        st0[q] = digit_state32<T, BL, L>(x0);
        st1[q] = digit_state32<T, BL, L>(x1);
      }
#pragma unroll
      for (int l = L; l >= 1; l--) {
        double2 x[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int32_t g0 = digit_next32<BL, L>(st0[q]), g1 = digit_next32<BL, L>(st1[q]);
          x[q] = make_double2((double)g0, (double)g1);   // the twist is in the transform
        }
        fft512_forward_tw<false>(x, lane, tab, sre, sim);
#pragma unroll
        for (int c = 0; c <= K; c++) {
#pragma unroll
          for (int g = 0; g < 8; g++) acc[c][g] = cfma(bq[c][g], x[g], acc[c][g]);
        }
        if (l > 1) load_rows(it, r, l - 1);
      }
    }

    // which step comes next, and is it executed?
    const bool last_row = (!run) || (r == K);
    const bool finish = run && (r == K);
    uint32_t it2 = it;
    int r2 = r + 1;
    T a2 = a_cur;
    bool run2 = run;
    if (last_row) {
      it2 = it + 1;
      r2 = 0;
      a2 = wave_uniform(a_nxt);
      run2 = (a2 != 0) && (it2 < n);
      a_nxt = in[(it2 + 1 < n) ? it2 + 1 : n];
    }
    if (run2) load_rows(it2, r2, L);

    if (finish) {
#pragma unroll
      for (int c = 0; c <= K; c++) {
        double2 z[8];
#pragma unroll
        for (int g = 0; g < 8; g++) {
          z[g] = acc[c][g];
          acc[c][g] = make_double2(0.0, 0.0);
        }
        fft512_inverse_tw<false>(z, lane, tab, sre, sim);
        T* pc = ct0 + c * N;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const uint32_t j0 = 64 * q + lane;
          const double2 v = untwist8(z[q], q);
          pc[j0] = torus_accumulate(pc[j0], v.x);
          pc[j0 + M] = torus_accumulate(pc[j0 + M], v.y);
        }
      }
    }
    it = it2;
    r = r2;
    a_cur = a2;
    run = run2;
  }

  // sample extraction of coefficient 0
  T* out = a.ct_out + ct * a.out_stride;
  for (int idx = lane; idx < K * N; idx += 64) {
    const int c = idx >> LOGN, j = idx & (N - 1);
    const T* p = ct0 + c * N;
    out[idx] = (j == 0) ? p[0] : (T)(T(0) - p[N - j]);
  }
  if (lane == 0) out[K * N] = ct0[K * N];
}


// ---- the bootstrap, second schedule (single decomposition level) ----
// Per CMUX: gather the k+1 rotated differences, run the k+1 forward transforms interleaved,
// multiply-accumulate against the whole GGSW (held in registers, loaded during the previous
// iteration's inverse transforms), run the k+1 inverse transforms interleaved, round and add.
// a wave moves on once all but 1/8 of its XCD's waves have arrived: waiting for the very last
// straggler costs 1.5 % of the run, this costs nothing and keeps the fetch traffic at its minimum

template <typename T, int K, int BL>
__global__ __launch_bounds__(256, 1) void wave_pbs_kernel_v2(PbsLaunch<T> a) {
  constexpr int N = 1024, M = 512, LOGN = 10, L = 1, R = K + 1;
  constexpr int START_STAGGER = 34;   // s_sleep units (64 cycles) between the starts of a CU's waves
  extern __shared__ __align__(16) char smem[];
  double2* tab = (double2*)smem;
  stage_folded_tables(tab, a.tables, 256);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* slab = smem + TWS_COUNT * sizeof(double2) + (size_t)wave * wave_slab_bytes<T, K>();
  T* ct0 = (T*)slab;
  double* sre = (double*)(slab + (size_t)R * N * sizeof(T));
  double* sim = sre + PLANE;

  // Persistent waves: the grid is sized to the device (one workgroup per CU) and every wave
  // walks its ciphertexts with a grid stride. All waves then start each of their ciphertexts at
  // about the same time, so the waves of an XCD read a key row within the same short window and
  // it is fetched into that XCD's L2 once; with one workgroup per four ciphertexts the start
  // times spread out as the launch progresses and every row was fetched about five times.
  // Waves that share an L2 (workgroups are dealt round-robin to the 8 XCDs) restart together:
  // after each ciphertext a wave reports to its XCD's counter and waits until all of them have.
  // Without this the waves drift apart by more key rows than the 4 MB L2 holds within a few
  // ciphertexts, and each row is fetched from the fabric many times per XCD.
  XcdRendezvous rendezvous;
  if (a.scratch) rendezvous.enrol(a.scratch, lane);
  const size_t per_round = (size_t)gridDim.x * WAVES;
  uint32_t round = 0;
  for (size_t ct = (size_t)blockIdx.x * WAVES + wave; ct < a.count; ct += per_round, round++) {
  const T* in = a.ct_in + ct * a.in_stride;
  const uint32_t n = a.n;
  const uint32_t lut_id = checked_lut_id(a, ct);
  const T* lut = a.luts + (size_t)lut_id * (R * N);
  {
    const uint32_t b_hat = modswitch<T>(in[n], LOGN);
    const uint32_t d0 = (2 * N - b_hat) & (2 * N - 1);
    for (int idx = lane; idx < R * N; idx += 64) {
      const int c = idx >> LOGN, j = idx & (N - 1);
      ct0[idx] = rotated_coeff<T>(lut + (size_t)c * N, j, d0, N);
    }
  }

  // first iteration whose mask element is not zero (the reference skips the others)
  uint32_t i = 0;
  T a_cur = wave_uniform(in[0]);
  while (i < n && a_cur == 0) {
    i++;
    a_cur = wave_uniform(in[i]);  // in[n] is the body: readable
  }

  // The four waves of a CU run the same program and share its LDS; they leave the rendezvous (and
  // the start of the launch) together. Wave w starts its iterations w eighths of an iteration
  // late (34 x 64 cycles each), so that the store-heavy exchanges of one wave fall beside the
  // arithmetic of the others from the first iteration on instead of after the waves have drifted
  // apart by themselves: 90.9 -> 90.1 ms per 2^14 (17, 68 and 100 x 64 cycles: 90.0-90.2). The
  // delay has to sit here, behind the table rotation and the scan for the first mask element
  // (ahead of them it gains nothing: their loads realign the waves), and ahead of the zeroing of
  // the parked registers, so that the compiler has no reason to move one of those afterwards.
  for (int k = 0; k < wave; k++) __builtin_amdgcn_s_sleep(START_STAGGER);

  // The GGSW of one iteration is R*R slabs of 8 registers (one 8 KiB slab per (row, col)).
  // NA of them are parked in the accumulation registers (AGPRs): their loads are issued one
  // iteration ahead and write the AGPRs directly, so they cost no vector registers while in
  // flight across the inverse transforms, the rounding, the next gather and the next forward
  // transforms. The remaining slabs do not fit the 256 AGPRs; they are loaded into vector
  // registers at the top of their own iteration. The loads are inline assembly (the compiler
  // cannot be told "load into an AGPR"), so the wait for them is explicit as well.
  constexpr int NS = R * R;                     // slabs per iteration
  constexpr int NA = NS * 32 <= 224 ? NS : 7;   // slabs that live in AGPRs
  v4u bqa[NA][8];
#pragma unroll
  for (int sl = 0; sl < NA; sl++)
#pragma unroll
    for (int g = 0; g < 8; g++) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(bqa[sl][g].x));
  const uint32_t lane_off = (uint32_t)lane * 16u;

  // The loop body runs once as a pure "load" pass (work = false) so that the key loads have a
  // single site; afterwards every pass transforms iteration `cur` and loads iteration `i`.
  bool work = false;
  uint32_t a_hat = 0, cur = 0;
#ifdef CAMD_PROFILE
  const uint64_t c_begin = __builtin_amdgcn_s_memtime(), r_begin = __builtin_amdgcn_s_memrealtime();
  uint64_t tG = 0, tF = 0, tM = 0, tL = 0, tI = 0, tR = 0, t0, t1;
#define STAMP(acc) t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1;
#else
#define STAMP(acc)
#endif
  for (;;) {
#ifdef CAMD_PROFILE
    t0 = __builtin_amdgcn_s_memtime();
#endif
    double2 x[R][8];
    v4u bqv[NS - NA > 0 ? NS - NA : 1][8];
    if (work) {
      // gather rotated - own, decompose (one level), twist. The LDS reads are issued in units
      // of half a polynomial, one unit ahead of the arithmetic that consumes them, and the
      // order is pinned so that the latency of a unit hides behind the previous unit's work.
      T own0[2][4], own1[2][4], rot0[2][4], rot1[2][4];
      auto issue = [&](int u) {
        const T* p = ct0 + (u >> 1) * N;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const uint32_t j0 = 64 * (4 * (u & 1) + t) + lane, j1 = j0 + M;
          own0[u & 1][t] = p[j0];
          own1[u & 1][t] = p[j1];
          rot0[u & 1][t] = p[(j0 - a_hat) & (N - 1)];
          rot1[u & 1][t] = p[(j1 - a_hat) & (N - 1)];
        }
      };
      auto consume = [&](int u) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int q = 4 * (u & 1) + t;
          const uint32_t j0 = 64 * q + lane, j1 = j0 + M;
          const int32_t g0 = digit_of_rotated_difference<T, BL, LOGN>(rot0[u & 1][t], own0[u & 1][t], j0 - a_hat);
          const int32_t g1 = digit_of_rotated_difference<T, BL, LOGN>(rot1[u & 1][t], own1[u & 1][t], j1 - a_hat);
          x[u >> 1][q] = make_double2((double)g0, (double)g1);   // the twist is in the transform
        }
      };
      issue(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2 * R; u++) {
        if (u + 1 < 2 * R) issue(u + 1);
        __builtin_amdgcn_sched_barrier(0);
        consume(u);
        __builtin_amdgcn_sched_barrier(0);
      }
      STAMP(tG)
      // the slabs that do not fit the AGPRs: requested now, needed after the forward transforms
#pragma unroll
      for (int sl = NA; sl < NS; sl++) {
        const char* base = (const char*)(a.fbsk + (size_t)cur * NS * M) + (size_t)sl * (M * sizeof(double2)) + 4096;
#pragma unroll
        for (int g = 0; g < 8; g++)
          asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"
                       : "=v"(bqv[sl - NA][g])
                       : "v"(lane_off), "s"(base), "n"(g * 1024 - 4096)
                       : "memory");
      }
      fft512_forward_multi<R>(x, lane, tab, sre, sim);
      STAMP(tF)
      // acc[c] = sum_r GGSW[r][c] * F[r], written over x column by column is not possible
      // (every column needs every row), so the sums go to a second array
    }
    double2 acc[R][8];
    if (!work) {
      // load-only first pass: the inverse transforms and the rounding below run on zeros (they
      // add nothing to the accumulator), which keeps a single site for the key loads
#pragma unroll
      for (int c = 0; c < R; c++)
#pragma unroll
        for (int g = 0; g < 8; g++) {
          // each zero is written by its own asm statement: as plain constants the compiler clears
          // the 48 registers ahead of the branch, in every iteration
          asm volatile("v_mov_b64 %0, 0" : "=v"(acc[c][g].x));
          asm volatile("v_mov_b64 %0, 0" : "=v"(acc[c][g].y));
        }
    } else {
      // every key load of this iteration must have landed before the products
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int sl = 0; sl < NA; sl++)
        asm volatile("" : "+a"(bqa[sl][0]), "+a"(bqa[sl][1]), "+a"(bqa[sl][2]), "+a"(bqa[sl][3]),
                          "+a"(bqa[sl][4]), "+a"(bqa[sl][5]), "+a"(bqa[sl][6]), "+a"(bqa[sl][7]));
#pragma unroll
      for (int sl = NA; sl < NS; sl++)
        asm volatile("" : "+v"(bqv[sl - NA][0]), "+v"(bqv[sl - NA][1]), "+v"(bqv[sl - NA][2]), "+v"(bqv[sl - NA][3]),
                          "+v"(bqv[sl - NA][4]), "+v"(bqv[sl - NA][5]), "+v"(bqv[sl - NA][6]), "+v"(bqv[sl - NA][7]));
      auto key = [&](int r, int c, int g) -> double2 {
        const int sl = r * R + c;
        return as_double2(sl < NA ? bqa[sl < NA ? sl : 0][g] : bqv[sl >= NA ? sl - NA : 0][g]);
      };
#pragma unroll
      for (int c = 0; c < R; c++)
#pragma unroll
        for (int g = 0; g < 8; g++) {
          double2 v = cmul(key(0, c, g), x[0][g]);
#pragma unroll
          for (int r = 1; r < R; r++) v = cfma(key(r, c, g), x[r][g], v);
          acc[c][g] = v;
        }
    }

    STAMP(tM)
    // key rows of the next active iteration (index i), if any
    const bool more = i < n;
    // Key rows of the next active iteration (index i). Each slab is eight 1 KiB requests; they
    // are spread over the inverse transforms and the rounding (one slab per slot) instead of
    // being issued in one burst, so that the four waves of a CU do not queue 288 KiB on the
    // load path at the same moment.
    const char* ggsw = (const char*)(a.fbsk + (size_t)(more ? i : 0) * NS * M);  // wave-uniform
    auto load_slab = [&](int sl) {
      if (more) {
        // scalar base in the middle of the slab, eight pieces around it
        const char* base = ggsw + (size_t)sl * (M * sizeof(double2)) + 4096;
#pragma unroll
        for (int g = 0; g < 8; g++)
          asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"
                       : "=a"(bqa[sl][g])
                       : "v"(lane_off), "s"(base), "n"(g * 1024 - 4096)
                       : "memory");
      }
    };
    // slot k (0..5) takes slabs [SLOT_LO(k), SLOT_LO(k+1))
    auto load_slot = [&](int k) {
#pragma unroll
      for (int sl = (k * NA) / 6; sl < ((k + 1) * NA) / 6; sl++) load_slab(sl);
    };
    const uint32_t next_i = i;
    const uint32_t next_hat = modswitch<T>(a_cur, LOGN) & (2 * N - 1);
    // scan ahead for the iteration after that one
    if (more) {
      i++;
      a_cur = wave_uniform(in[i]);
      while (i < n && a_cur == 0) {
        i++;
        a_cur = wave_uniform(in[i]);
      }
    }

    STAMP(tL)
    {
      fft512_inverse_multi<R>(acc, lane, tab, sre, sim, load_slot);
      STAMP(tI)
      // untwist, round, add into the accumulator: old values are read one column ahead
      T old0[2][8], old1[2][8];
      auto issue_old = [&](int c) {
        const T* pc = ct0 + c * N;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          old0[c & 1][q] = pc[64 * q + lane];
          old1[c & 1][q] = pc[64 * q + lane + M];
        }
      };
      issue_old(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < R; c++) {
        if (c + 1 < R) issue_old(c + 1);
        if (c < 3) load_slot(3 + c);
        __builtin_amdgcn_sched_barrier(0);
        T* pc = ct0 + c * N;
        if constexpr (sizeof(T) == 8) {
          // one coefficient pair behind: the multiply-adds of pair q-1 sit between the arithmetic
          // of pair q (two of them back to back cost an idle slot as well)
          TorusSum prev0, prev1;
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const double2 v = untwist8(acc[c][q], q);
            const TorusSum cur0 = torus_accumulate_begin(old0[c & 1][q], v.x);
            if (q > 0) pc[64 * (q - 1) + lane] = torus_accumulate_end(prev0);
            const TorusSum cur1 = torus_accumulate_begin(old1[c & 1][q], v.y);
            if (q > 0) pc[64 * (q - 1) + lane + M] = torus_accumulate_end(prev1);
            prev0 = cur0;
            prev1 = cur1;
          }
          pc[64 * 7 + lane] = torus_accumulate_end(prev0);
          __builtin_amdgcn_sched_barrier(0);
          pc[64 * 7 + lane + M] = torus_accumulate_end(prev1);
        } else {
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const uint32_t j0 = 64 * q + lane;
            const double2 v = untwist8(acc[c][q], q);
            pc[j0] = torus_accumulate(old0[c & 1][q], v.x);
            pc[j0 + M] = torus_accumulate(old1[c & 1][q], v.y);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(tR)
    if (!more) break;
    work = true;
    a_hat = next_hat;
    cur = next_i;
  }
#ifdef CAMD_PROFILE
  if ((ct == 0 || ct == 5000 || ct == 9001) && lane == 0) {
    const uint64_t dc = __builtin_amdgcn_s_memtime() - c_begin, dr = __builtin_amdgcn_s_memrealtime() - r_begin;
    printf("ct %lu cycles per iteration: gather %lu fwd %lu mac %lu load %lu inv %lu round %lu | total memtime %lu realtime(100MHz) %lu\n", (unsigned long)ct,
           (unsigned long)(tG / n), (unsigned long)(tF / n), (unsigned long)(tM / n), (unsigned long)(tL / n),
           (unsigned long)(tI / n), (unsigned long)(tR / n), (unsigned long)dc, (unsigned long)dr);
  }
#endif

  T* out = a.ct_out + ct * a.out_stride;
  for (int idx = lane; idx < K * N; idx += 64) {
    const int c = idx >> LOGN, j = idx & (N - 1);
    const T* p = ct0 + c * N;
    out[idx] = (j == 0) ? p[0] : (T)(T(0) - p[N - j]);
  }
  if (lane == 0) out[K * N] = ct0[K * N];
  // rendezvous, only when a further round follows (this one was then a full round); the
  // wait is bounded so that a wave can never be held forever (the order of work does not
  // affect results, only cache behaviour)
  if ((size_t)(round + 1) * per_round < a.count) rendezvous.arrive_and_wait(round, lane);
  }
}

// ---- the bootstrap, row-step schedule (several decomposition levels) ----
// The second schedule for a GGSW of more than one level: a step is one row r of the rotated
// difference. Its L digit polynomials are transformed interleaved (as the k+1 rows are above), the
// L (k+1) key polynomials of that row wait in accumulation registers (requested during the step
// before), and the Fourier sums of the k+1 columns stay in registers across the k+1 steps of a
// CMUX; after the last row they are transformed back, interleaved, and rounded into the
// accumulator. Persistent waves with the per-XCD rendezvous, as above. Products are accumulated
// rows outermost, levels from the finest up: the order of the first schedule and of the k+1-wave
// variant, hence the same ciphertext words.
template <typename T, int K, int L, int BL>
__global__ __launch_bounds__(256, 1) void wave_pbs_kernel_rows(PbsLaunch<T> a) {
  constexpr int N = 1024, M = 512, LOGN = 10, R = K + 1;
  constexpr int NS = L * R;   // key polynomials of a step: [level][column], all parked
  static_assert(NS * 32 <= 224, "a step's key rows must fit the accumulation registers");
  extern __shared__ __align__(16) char smem[];
  double2* tab = (double2*)smem;
  stage_folded_tables(tab, a.tables, 256);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* slab = smem + TWS_COUNT * sizeof(double2) + (size_t)wave * wave_slab_bytes<T, K>();
  T* ct0 = (T*)slab;
  double* sre = (double*)(slab + (size_t)R * N * sizeof(T));
  double* sim = sre + PLANE;

  XcdRendezvous rendezvous;
  if (a.scratch) rendezvous.enrol(a.scratch, lane);
  const size_t per_round = (size_t)gridDim.x * WAVES;
  uint32_t round = 0;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  for (size_t ct = (size_t)blockIdx.x * WAVES + wave; ct < a.count; ct += per_round, round++) {
    const T* in = a.ct_in + ct * a.in_stride;
    const uint32_t n = a.n;
    const uint32_t lut_id = checked_lut_id(a, ct);
    const T* lut = a.luts + (size_t)lut_id * (R * N);
    {
      const uint32_t b_hat = modswitch<T>(in[n], LOGN);
      const uint32_t d0 = (2 * N - b_hat) & (2 * N - 1);
      for (int idx = lane; idx < R * N; idx += 64) {
        const int c = idx >> LOGN, j = idx & (N - 1);
        ct0[idx] = rotated_coeff<T>(lut + (size_t)c * N, j, d0, N);
      }
    }
    // first iteration whose mask element is not zero (the reference skips the others)
    uint32_t i = 0;
    T a_cur = wave_uniform(in[0]);
    while (i < n && a_cur == 0) {
      i++;
      a_cur = wave_uniform(in[i]);  // in[n] is the body: readable
    }

    v4u bqa[NS][8];
#pragma unroll
    for (int sl = 0; sl < NS; sl++)
#pragma unroll
      for (int g = 0; g < 8; g++) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(bqa[sl][g].x));

    double2 acc[R][8];
#pragma unroll
    for (int c = 0; c < R; c++)
#pragma unroll
      for (int g = 0; g < 8; g++) acc[c][g] = make_double2(0.0, 0.0);

    // The loop body runs once as a pure "load" pass (work = false) so that the key loads have a
    // single site; afterwards every pass transforms step (cur, r) and loads the step after it.
    bool work = false;
    uint32_t a_hat = 0, cur = 0;
    int r = 0;
    for (;;) {
      if (work) {
        double2 x[L][8];
        {
          const T* p = ct0 + r * N;
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const uint32_t j0 = 64 * q + lane, j1 = j0 + M;
            const T x0 = rotated_coeff<T>(p, j0, a_hat, N) - p[j0];
            const T x1 = rotated_coeff<T>(p, j1, a_hat, N) - p[j1];
            uint32_t st0 = digit_state32<T, BL, L>(x0), st1 = digit_state32<T, BL, L>(x1);
#pragma unroll
            for (int l = L; l >= 1; l--) {
              const int32_t g0 = digit_next32<BL, L>(st0), g1 = digit_next32<BL, L>(st1);
              x[l - 1][q] = make_double2((double)g0, (double)g1);   // the twist is in the transform
            }
          }
        }
        fft512_forward_multi<L>(x, lane, tab, sre, sim);
        // every key load of this step must have landed before the products
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int sl = 0; sl < NS; sl++)
          asm volatile("" : "+a"(bqa[sl][0]), "+a"(bqa[sl][1]), "+a"(bqa[sl][2]), "+a"(bqa[sl][3]),
                            "+a"(bqa[sl][4]), "+a"(bqa[sl][5]), "+a"(bqa[sl][6]), "+a"(bqa[sl][7]));
#pragma unroll
        for (int c = 0; c < R; c++)
#pragma unroll
          for (int g = 0; g < 8; g++) {
            double2 v = acc[c][g];
#pragma unroll
            for (int l = L; l >= 1; l--) v = cfma(as_double2(bqa[(l - 1) * R + c][g]), x[l - 1][g], v);
            acc[c][g] = v;
          }
      }

      // the step after this one: the next row of the same iteration, or row 0 of the next
      // iteration whose mask element is not zero
      const bool last_row = !work || r == K;
      const bool more = last_row ? (i < n) : true;
      const uint32_t next_it = last_row ? i : cur;
      const int next_r = last_row ? 0 : r + 1;
      const uint32_t next_hat = last_row ? (modswitch<T>(a_cur, LOGN) & (2 * N - 1)) : a_hat;
      if (more) {
        // [it][level][row][col] polynomials of M complex values; scalar base in the middle of each
        const char* row0 = (const char*)(a.fbsk + (((size_t)next_it * L * R + (size_t)next_r) * R) * M) + 4096;
#pragma unroll
        for (int l = 1; l <= L; l++)
#pragma unroll
          for (int c = 0; c < R; c++) {
            const char* base = row0 + ((size_t)(l - 1) * R * R + c) * (M * sizeof(double2));
#pragma unroll
            for (int g = 0; g < 8; g++)
              asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"
                           : "=a"(bqa[(l - 1) * R + c][g])
                           : "v"(lane_off), "s"(base), "n"(g * 1024 - 4096)
                           : "memory");
          }
      }
      if (last_row && more) {   // scan ahead for the iteration after that one
        i++;
        a_cur = wave_uniform(in[i]);
        while (i < n && a_cur == 0) {
          i++;
          a_cur = wave_uniform(in[i]);
        }
      }

      if (work && r == K) {
        fft512_inverse_multi<R>(acc, lane, tab, sre, sim);
#pragma unroll
        for (int c = 0; c < R; c++) {
          T* pc = ct0 + c * N;
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const uint32_t j0 = 64 * q + lane;
            const double2 v = untwist8(acc[c][q], q);
            pc[j0] = torus_accumulate(pc[j0], v.x);
            pc[j0 + M] = torus_accumulate(pc[j0 + M], v.y);
            acc[c][q] = make_double2(0.0, 0.0);
          }
        }
      }
      if (!more) break;
      work = true;
      cur = next_it;
      r = next_r;
      a_hat = next_hat;
    }

    T* out = a.ct_out + ct * a.out_stride;
    for (int idx = lane; idx < K * N; idx += 64) {
      const int c = idx >> LOGN, j = idx & (N - 1);
      const T* p = ct0 + c * N;
      out[idx] = (j == 0) ? p[0] : (T)(T(0) - p[N - j]);
    }
    if (lane == 0) out[K * N] = ct0[K * N];
    if ((size_t)(round + 1) * per_round < a.count) rendezvous.arrive_and_wait(round, lane);
  }
}

// per-wave LDS of the k+1-wave variant: exchange image + one staging row for the rotation
template <typename T, int K>
constexpr size_t lat_slab_bytes() {
  return 2 * PLANE * sizeof(double) + 1024 * sizeof(T);
}

// ---- the bootstrap, latency schedule: k+1 waves per ciphertext ----
// A batch smaller than the chip leaves most SIMDs idle while each lone wave walks its n CMUXes
// at one instruction per four cycles. Here the k+1 waves of a workgroup share one ciphertext:
// wave c owns polynomial c of the accumulator (in registers), rotates,
// decomposes and forward-transforms it, and publishes the spectrum in LDS; after one barrier it
// multiplies all k+1 spectra with column c of the key row (requested at the top of the
// iteration, so the loads fly behind the forward transform), transforms back and adds into its
// polynomial. One transform each way per wave and CMUX instead of k+1. The spectra alternate
// between two images, so one barrier per CMUX is enough: a wave can only overwrite an image after
// the barrier of the following CMUX, which every reader of that image has reached by then.
// Same products in the same order as the first one-wave schedule, hence the same ciphertext words.
template <typename T, int K, int L>
constexpr size_t wavelat_lds_bytes() {
  return TWS_COUNT * sizeof(double2) + (K + 1) * lat_slab_bytes<T, K>() + 2 * (size_t)(K + 1) * L * 512 * sizeof(double2);
}

template <typename T, int K, int L, int BL>
__global__ __launch_bounds__(64 * (K + 1)) void wave_pbs_kernel_lat(PbsLaunch<T> a) {
  constexpr int N = 1024, M = 512, LOGN = 10, R = K + 1, S = R * L, NS = S * R;
  extern __shared__ __align__(16) char smem[];
  double2* tab = (double2*)smem;
  stage_folded_tables(tab, a.tables, 64 * R);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // = the polynomial this wave owns
  char* slab = smem + TWS_COUNT * sizeof(double2) + (size_t)wave * lat_slab_bytes<T, K>();
  double* sre = (double*)slab;
  double* sim = sre + PLANE;
  T* stage = (T*)(slab + 2 * PLANE * sizeof(double));
  double2* spectra = (double2*)(smem + TWS_COUNT * sizeof(double2) + (size_t)R * lat_slab_bytes<T, K>());  // [2][S][8][64]
  const uint32_t n = a.n;
  uint32_t image = 0;

  for (size_t ct = blockIdx.x; ct < a.count; ct += gridDim.x) {
    const T* in = a.ct_in + ct * a.in_stride;
    const uint32_t lut_id = checked_lut_id(a, ct);
    const T* lut = a.luts + (size_t)lut_id * (R * N) + (size_t)wave * N;

    // acc[2q] = coefficient 64q + lane, acc[2q+1] = coefficient 64q + lane + N/2 of the own polynomial
    T acc[16];
    {
      const uint32_t b_hat = modswitch<T>(in[n], LOGN);
      const uint32_t d0 = (2 * N - b_hat) & (2 * N - 1);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        acc[2 * q] = rotated_coeff<T>(lut, 64 * q + lane, d0, N);
        acc[2 * q + 1] = rotated_coeff<T>(lut, 64 * q + lane + M, d0, N);
      }
    }

    // One CMUX: column `wave` of key row `it` (stream (level l, row r) at index (l - 1) R + r) is
    // requested first and flies behind the rotation and the forward transform.
    auto cmux = [&](uint32_t it, T a_val) {
      const uint32_t a_hat = modswitch<T>(a_val, LOGN) & (2 * N - 1);
      const double2* key = a.fbsk + (size_t)it * NS * M + (size_t)wave * M + lane;
      double2 kk[S][8];
#pragma unroll
      for (int st = 0; st < S; st++)
#pragma unroll
        for (int g = 0; g < 8; g++) kk[st][g] = key[(size_t)(st * R) * M + g * 64];

      // rotation by X^a_hat through the staging row, then the digits of (rotated - own)
      uint32_t st0[8], st1[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        stage[64 * q + lane] = acc[2 * q];
        stage[64 * q + lane + M] = acc[2 * q + 1];
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t j0 = 64 * q + lane, j1 = j0 + M;
        const T v0 = stage[(j0 - a_hat) & (N - 1)], v1 = stage[(j1 - a_hat) & (N - 1)];
        const T r0 = ((j0 - a_hat) & N) ? (T)(T(0) - v0) : v0;
        const T r1 = ((j1 - a_hat) & N) ? (T)(T(0) - v1) : v1;
        st0[q] = digit_state32<T, BL, L>((T)(r0 - acc[2 * q]));
        st1[q] = digit_state32<T, BL, L>((T)(r1 - acc[2 * q + 1]));
      }
      double2* img = spectra + (size_t)image * (S * M);
      double2 x[8];
#pragma unroll
      for (int l = L; l >= 1; l--) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int32_t d0 = digit_next32<BL, L>(st0[q]), d1 = digit_next32<BL, L>(st1[q]);
          x[q] = make_double2((double)d0, (double)d1);   // the twist is in the transform
        }
        fft512_forward_tw<true>(x, lane, tab, sre, sim);
#pragma unroll
        for (int g = 0; g < 8; g++) img[(((l - 1) * R + wave) * 8 + g) * 64 + lane] = x[g];
      }
      __syncthreads();
      image ^= 1u;

      // rows outermost, levels from the finest up: the order of the one-wave schedules
#pragma unroll
      for (int g = 0; g < 8; g++) {
        double2 v = cmul(kk[(L - 1) * R][g], img[(((L - 1) * R) * 8 + g) * 64 + lane]);
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
          for (int l = L; l >= 1; l--) {
            if (r == 0 && l == L) continue;
            const int st = (l - 1) * R + r;
            v = cfma(kk[st][g], img[(st * 8 + g) * 64 + lane], v);
          }
        x[g] = v;
      }
      fft512_inverse_tw<true>(x, lane, tab, sre, sim);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const double2 v = untwist8(x[q], q);
        acc[2 * q] = torus_accumulate(acc[2 * q], v.x);
        acc[2 * q + 1] = torus_accumulate(acc[2 * q + 1], v.y);
      }
    };
    // Iterations whose mask element is zero are skipped, as the reference does; every wave of the
    // group sees the same values. The two loop forms compile to different schedules, and each
    // instance takes the one that measured faster (single level: 2.8 against 3.1 ms per 2_2
    // bootstrap; three levels: 4.2 against 5.3 ms per gate).
    if constexpr (L == 1) {
      T a_next = wave_uniform(in[0]);
      for (uint32_t i = 0; i < n; i++) {
        const T a_cur = a_next;
        a_next = wave_uniform(in[i + 1]);  // in[n] is the body: readable
        if (a_cur == 0) continue;
        cmux(i, a_cur);
      }
    } else {
      uint32_t i = 0;
      T a_cur = wave_uniform(in[0]);
      while (i < n && a_cur == 0) {
        i++;
        a_cur = wave_uniform(in[i]);
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
    }

    // sample extraction of coefficient 0: out[c N + (N - j) mod N] = (j == 0) ? p[j] : -p[j]
    T* out = a.ct_out + ct * a.out_stride;
    if (wave < K) {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t j0 = 64 * q + lane, j1 = j0 + M;
        out[(size_t)wave * N + ((N - j0) & (N - 1))] = (j0 == 0) ? acc[2 * q] : (T)(T(0) - acc[2 * q]);
        out[(size_t)wave * N + (N - j1)] = (T)(T(0) - acc[2 * q + 1]);
      }
    } else if (lane == 0) {
      out[K * N] = acc[0];
    }
  }
}


template <typename T, int K, int L, int BL>
int launch_instance(const PbsLaunch<T>& a, hipStream_t stream) {
  constexpr size_t lds = wave_lds_bytes<T, K>();
  static_assert(lds <= 160 * 1024, "LDS budget exceeded");
  const unsigned grid = (unsigned)((a.count + WAVES - 1) / WAVES);
  static const int schedule = []() {
    // 3 (default where it has an instance): two waves per SIMD, pbs_wave8.hip; 2: one wave per
    // SIMD with the key rows parked in registers; 1: the first schedule
    const char* e = getenv("CAMD_WAVE_SCHEDULE");
    return e ? atoi(e) : 3;
  }();
  // fewer ciphertexts than CUs: k+1 waves share each one
  {
    int dev = 0, cus = 0;
    CAMD_HIP(hipGetDevice(&dev));
    CAMD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    constexpr int lat_fit = (int)((160 * 1024) / wavelat_lds_bytes<T, K, L>());
    if (a.count <= wave_latency_limit(cus, lat_fit)) {
      constexpr size_t ldsl = wavelat_lds_bytes<T, K, L>();
      static_assert(ldsl <= 160 * 1024, "LDS budget exceeded");
      auto kernl = wave_pbs_kernel_lat<T, K, L, BL>;
      CAMD_HIP(hipFuncSetAttribute((const void*)kernl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsl));
      hipLaunchKernelGGL(kernl, dim3((unsigned)a.count), dim3(64 * (K + 1)), ldsl, stream, a);
      CAMD_HIP(hipGetLastError());
      return CAMD_OK;
    }
  }
  if constexpr (L == 1 && sizeof(T) == 8) {
    if (schedule == 3 && wave8_supported(64, 1024, K, L, BL)) return wave8_launch_pbs<T>(a, stream);
  }
  if constexpr (L == 1) {
    if (schedule == 2 || schedule == 3) {
      auto kern2 = wave_pbs_kernel_v2<T, K, BL>;
      CAMD_HIP(hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int dev = 0, cus = 0;
      CAMD_HIP(hipGetDevice(&dev));
      CAMD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      const unsigned persistent = cus > 0 && grid > (unsigned)cus ? (unsigned)cus : grid;  // LDS allows one workgroup per CU
      if (persistent < grid) {
        if (!a.scratch) return fail(CAMD_EINVAL, "wave kernel: arrival counters were not reserved");
        CAMD_HIP(hipMemsetAsync(a.scratch, 0, RENDEZVOUS_BYTES, stream));
      }
      hipLaunchKernelGGL(kern2, dim3(persistent), dim3(256), lds, stream, a);
      CAMD_HIP(hipGetLastError());
      return CAMD_OK;
    }
  }
  if constexpr (L > 1) {
    if (schedule == 2 || schedule == 3) {
      auto kernr = wave_pbs_kernel_rows<T, K, L, BL>;
      CAMD_HIP(hipFuncSetAttribute((const void*)kernr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int dev = 0, cus = 0;
      CAMD_HIP(hipGetDevice(&dev));
      CAMD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      const unsigned persistent = cus > 0 && grid > (unsigned)cus ? (unsigned)cus : grid;
      if (persistent < grid) {
        if (!a.scratch) return fail(CAMD_EINVAL, "wave kernel: arrival counters were not reserved");
        CAMD_HIP(hipMemsetAsync(a.scratch, 0, RENDEZVOUS_BYTES, stream));
      }
      hipLaunchKernelGGL(kernr, dim3(persistent), dim3(256), lds, stream, a);
      CAMD_HIP(hipGetLastError());
      return CAMD_OK;
    }
  }
  auto kern = wave_pbs_kernel<T, K, L, BL>;
  CAMD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

}  // namespace

bool wave_scratch_is_fixed(const camd_params& p) { return !wave16k_supported(p); }
size_t wave_scratch_bytes(const camd_params& p, size_t count) {
  if (wave16k_supported(p)) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
    return wave16k_scratch_bytes(p.polynomial_size, nullptr, count, cus);
  }
  return RENDEZVOUS_BYTES;
}

// the boolean TFHE-lib parameters (32-bit torus, N = 1024, k = 1, three levels of base 2^7): first
// schedule for large batches, the k+1-wave variant for small ones
static bool tfhe_lib_instance(const camd_params& p) {
  return p.torus_bits == 32 && p.polynomial_size == 1024 && p.glwe_dimension == 1 && p.pbs_level == 3 && p.pbs_base_log == 7;
}

// [register g][lane] with frequency 64 g + 8 (lane & 7) + (lane >> 3) at N = 1024 and the radix-4
// digit order at N = 512, values pre-scaled by 1/(N/2): fixed by the transforms of these files
uint64_t wave_layout_tag(const camd_params& p) {
  constexpr uint64_t REVISION = 1;
  return (REVISION << 32) | p.polynomial_size;
}

bool wave_supported(const camd_params& p) {
  if (p.polynomial_size == 1024 && p.torus_bits == 64)
    return p.glwe_dimension == 2 && p.pbs_level == 1 && p.pbs_base_log == 23;
  return wave512_supported(p) || tfhe_lib_instance(p) || wave16k_supported(p);
}

// every set with an instance gets it by default; the bit-exact kernels stay selectable
// (CAMD_WAVE_16K=0 keeps the register-blocked kernel as the default of the N = 16384 sets)
bool wave_default(const camd_params& p) {
  static const bool big = []() {
    const char* e = getenv("CAMD_WAVE_16K");
    return e ? atoi(e) != 0 : true;
  }();
  // three levels of 2^11 at N = 16384: so little output noise that rounding the accumulator at 2^-32 would
  // triple it; that instance is for callers that ask for it
  if (wave16k_supported(p)) return big && p.pbs_level < 3;
  return wave_supported(p);
}

size_t wave_tables_count(uint32_t N) {
  return N == 512 ? wave512_tables_count() : N >= 2048 ? wave16k_tables_count() : (size_t)TAB_TOTAL;
}

void wave_fill_tables(uint32_t N, double2* t) {
  if (N == 512) {
    wave512_fill_tables(t);
    return;
  }
  if (N >= 2048) {
    wave16k_fill_tables(N, t);
    return;
  }
  for (int j = 0; j < 512; j++) {
    const double ang = M_PI * (double)j / (double)N;
    t[TAB_TWIST + j] = make_double2(std::cos(ang), std::sin(ang));
  }
  for (int b = 1; b < 8; b++)
    for (int m = 0; m < 64; m++) {
      const double ang = -2.0 * M_PI * (double)(m * b) / 512.0;
      t[TAB_TW0 + (b - 1) * 64 + m] = make_double2(std::cos(ang), std::sin(ang));
    }
  for (int b = 1; b < 8; b++)
    for (int m = 0; m < 8; m++) {
      const double ang = -2.0 * M_PI * (double)(m * b) / 64.0;
      t[TAB_TW1 + (b - 1) * 8 + m] = make_double2(std::cos(ang), std::sin(ang));
    }
  // folded constants of the twiddled passes (see dft8_tw): th[q] is the angle of input q
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
    for (int q = 0; q < 8; q++) th[q] = M_PI * (double)((8 * q + lo) * (1 - 4 * hi)) / (double)N;
    fold(t + TAB_TWS + TWS_A + lane, 64, th, true);
    for (int b = 0; b < 8; b++) th[b] = M_PI * (double)(lane * (1 - 4 * b)) / (double)N;
    fold(t + TAB_TWS + TWS_C + lane, 64, th, true);
  }
  for (int lo = 0; lo < 8; lo++) {
    double th[8];
    for (int q = 0; q < 8; q++) th[q] = -2.0 * M_PI * (double)(lo * q) / 64.0;
    fold(t + TAB_TWS + TWS_B + lo, 8, th, false);
  }
}

template <typename T>
int wave_bsk_to_fourier(const T* bsk_std, double2* fbsk, const double2* tables, const camd_params& p,
                        hipStream_t stream) {
  if (!wave_supported(p)) return fail(CAMD_EINVAL, "wave kernel: unsupported parameter set");
  const size_t polys = (size_t)p.lwe_dimension * p.pbs_level * (p.glwe_dimension + 1) * (p.glwe_dimension + 1);
  if (wave512_supported(p)) {
    if ((int)sizeof(T) * 8 != (int)p.torus_bits) return fail(CAMD_EINVAL, "wave kernel: torus width mismatch");
    return wave512_bsk_to_fourier<T>(bsk_std, fbsk, tables, polys, stream);
  }
  if (wave16k_supported(p)) {
    if constexpr (sizeof(T) == 8) return wave16k_bsk_to_fourier<T>(bsk_std, fbsk, tables, p.polynomial_size, polys, stream);
    return fail(CAMD_EINVAL, "wave kernel: torus width mismatch");
  }
  const size_t lds = TAB_COUNT * sizeof(double2) + WAVES * 2 * PLANE * sizeof(double);
  hipLaunchKernelGGL(wave_bsk_fourier_kernel<T>, dim3((unsigned)((polys + WAVES - 1) / WAVES)), dim3(256), lds, stream,
                     bsk_std, fbsk, tables, polys);
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

template <typename T>
int wave_launch_pbs(const PbsLaunch<T>& a, hipStream_t stream) {
  if (a.count == 0) return CAMD_OK;
  if (a.N == 512) return wave512_launch_pbs<T>(a, stream);
  if (a.N >= 2048) return wave16k_launch_pbs<T>(a, stream);
  if constexpr (sizeof(T) == 8) {
    if (a.N == 1024 && a.k == 2 && a.level == 1 && a.base_log == 23) return launch_instance<T, 2, 1, 23>(a, stream);
  } else {
    if (a.N == 1024 && a.k == 1 && a.level == 3 && a.base_log == 7) return launch_instance<T, 1, 3, 7>(a, stream);
  }
  return fail(CAMD_EINVAL, "wave kernel: no instance for this parameter set");
}

template int wave_bsk_to_fourier<uint64_t>(const uint64_t*, double2*, const double2*, const camd_params&, hipStream_t);
template int wave_bsk_to_fourier<uint32_t>(const uint32_t*, double2*, const double2*, const camd_params&, hipStream_t);
template int wave_launch_pbs<uint64_t>(const PbsLaunch<uint64_t>&, hipStream_t);
template int wave_launch_pbs<uint32_t>(const PbsLaunch<uint32_t>&, hipStream_t);

}  // namespace camd
