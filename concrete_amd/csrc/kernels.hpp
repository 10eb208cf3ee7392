// Launcher declarations shared by the C ABI layer and the kernel translation units.
#pragma once

#include <algorithm>

#include "common.hpp"

namespace camd {

// The input ciphertexts of a keyswitch or a bootstrap, as a linear combination that the kernel
// forms while it reads: row c is alpha * x[c] + beta * y[c] (wrapping), with body_add added to its
// last word (index `body`). A plain batch is alpha = 1, y = nullptr, body_add = 0. This is how the
// leveled operations the reference performs ahead of a bootstrap (the shift-and-add of
// concrete-shortint/src/engine/server_side/mod.rs:182-188, the gate combinations of
// concrete-boolean/src/engine/mod.rs:347-353) ride on the first kernel's loads.
template <typename T>
struct LweInput {
  const T* x;
  size_t x_stride;
  const T* y;                // nullptr: no second operand
  size_t y_stride;
  T alpha, beta, body_add;
  uint32_t body;             // index of the body word = the dimension of the input ciphertexts
  static LweInput plain(const T* x, size_t stride, uint32_t dimension) {
    return LweInput{x, stride, nullptr, 0, T(1), T(0), T(0), dimension};
  }
  bool is_plain() const { return y == nullptr && alpha == T(1) && body_add == T(0); }
  __device__ __forceinline__ T at(size_t c, uint32_t j) const {
    T v = alpha * x[c * x_stride + j];
    if (y) v += beta * y[c * y_stride + j];
    return j == body ? (T)(v + body_add) : v;
  }
  LweInput rows_from(size_t first) const {
    LweInput r = *this;
    r.x += first * x_stride;
    if (r.y) r.y += first * y_stride;
    return r;
  }
};

// One row of an LweInput, indexed like the array it stands for.
template <typename T>
struct LweInputRow {
  const T* x;
  const T* y;
  T alpha, beta, body_add;
  uint32_t body;
  __device__ __forceinline__ T operator[](uint32_t j) const {
    T v = alpha * x[j];
    if (y) v += beta * y[j];
    return j == body ? (T)(v + body_add) : v;
  }
};

// Everything a bootstrap launch needs. Pointers are device memory.
template <typename T>
struct PbsLaunch {
  const T* ct_in;            // [count][in_stride], n+1 words used
  size_t in_stride;
  // Second operand and coefficients of an LweInput whose x is ct_in. Only the generic and block
  // families look at these (through input_row); the wave family is handed plain batches.
  const T* y_in;
  size_t y_stride;
  T alpha, beta, body_add;
  T* ct_out;                 // [count][out_stride], kN+1 words written
  size_t out_stride;
  const uint32_t* lut_ids;   // [count] or nullptr (lut 0)
  const T* luts;             // [lut_count][(k+1)*N]
  uint32_t lut_count;
  size_t count;
  const double2* fbsk;       // Fourier bootstrap key in the layout of the kernel family
  const double2* tables;     // twiddle tables of the kernel family
  void* scratch;             // per-workgroup scratch (generic and block families), arrival counters (wave family)
  uint32_t* status;          // launch status word in host-visible memory (bit 0: a look-up table id was out of range)
  uint32_t n, k, N, level, base_log;
};

constexpr uint32_t STATUS_BAD_LUT_ID = 1u;

template <typename T>
__device__ __forceinline__ LweInputRow<T> input_row(const PbsLaunch<T>& a, size_t ct) {
  return LweInputRow<T>{a.ct_in + ct * a.in_stride, a.y_in ? a.y_in + ct * a.y_stride : nullptr, a.alpha, a.beta,
                        a.body_add, a.n};
}

// Look-up table of ciphertext `ct`. The host-pointer entry points reject an id that is out of range
// before anything is launched; a device-pointer call cannot look at its ids without a round trip,
// so the kernel checks: a bad id raises STATUS_BAD_LUT_ID in the key's status word (the next call
// on the key reports it as CAMD_EINVAL) and the ciphertext is bootstrapped with table 0 so that
// every access stays in bounds.
template <typename T>
__device__ __forceinline__ uint32_t checked_lut_id(const PbsLaunch<T>& a, size_t ct) {
  const uint32_t id = a.lut_ids ? a.lut_ids[ct] : 0u;
  if (id < a.lut_count) return id;
  *(volatile uint32_t*)a.status = STATUS_BAD_LUT_ID;
  return 0u;
}

// ---- rendezvous of persistent waves that share an L2 (bootstrap kernels) ----
constexpr uint32_t WAVE_QUORUM_SLACK = 8;
constexpr uint32_t WAVE_QUORUM_GRACE = 128;   // polls (about 0.5 us each) granted to the stragglers once the quorum is there
constexpr int SYNC_GROUPS = 16;         // the hardware's XCC id is a 4-bit field (MI355X has 8 XCDs)
constexpr int SYNC_STRIDE = 32;         // counters 128 bytes apart
constexpr size_t RENDEZVOUS_BYTES = 2 * (size_t)SYNC_GROUPS * SYNC_STRIDE * sizeof(uint32_t);

// Rendezvous of the persistent waves that share an L2. Nothing is assumed about how workgroups are
// dealt to XCDs or how many XCDs there are: every wave reads the id of the XCD it runs on from the
// hardware register, enrols in that XCD's member count when it starts, and between ciphertexts
// reports to that XCD's arrival counter and waits for all members; once all but
// 1/WAVE_QUORUM_SLACK of them are there, the rest get a short grace period and are then left
// behind. Waiting for the quorum alone let waves that are a fraction of a percent slower fall
// further behind with every ciphertext until they streamed the key through the L2 on their own
// (measured on the N = 512 instance: four times the fetch traffic after 32 ciphertexts); waiting
// for every wave without a bound would tie the kernel to its slowest one. The first wait comes a whole bootstrap (milliseconds) after the launch, when every
// workgroup of the device-sized grid has long enrolled; should a member count still be short, or a
// wave lag, the wait is bounded and only cache behaviour is affected, never results.
// Both counter arrays live in the key's scratch buffer and are cleared by every launch.
struct XcdRendezvous {
  uint32_t* arrivals;
  uint32_t* members;
  __device__ __forceinline__ void enrol(void* scratch, int lane) {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4): register 20, offset 0, width 4
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & (SYNC_GROUPS - 1);
    arrivals = (uint32_t*)scratch + xcc * SYNC_STRIDE;
    members = arrivals + SYNC_GROUPS * SYNC_STRIDE;
    if (lane == 0) __hip_atomic_fetch_add(members, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // after ciphertext number `round` (0-based) of this wave
  __device__ __forceinline__ void arrive_and_wait(uint32_t round, int lane) {
    if (lane == 0) {
      __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t m = __hip_atomic_load(members, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t all = (round + 1) * m, quorum = all - m / WAVE_QUORUM_SLACK;
      uint32_t grace = WAVE_QUORUM_GRACE;
      for (uint32_t spin = 0; spin < (1u << 16); spin++) {
        const uint32_t seen = __hip_atomic_load(arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen >= all) break;
        if (seen >= quorum && grace-- == 0) break;
        __builtin_amdgcn_s_sleep(16);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
};
// ---- keyswitch (keyswitch.hip) ----
template <typename T>
int launch_keyswitch(const LweInput<T>& in, T* ct_out, size_t out_stride, const T* ksk,
                     uint32_t in_dim, uint32_t out_dim, uint32_t base_log, uint32_t level, size_t count,
                     hipStream_t stream);

// ---- keyswitch on the matrix cores (keyswitch_mfma.hip): same result, for batches ----
// limbs: the key split into balanced byte limbs (keyswitch_mfma_prepare, once per key);
// digits: scratch of keyswitch_mfma_digit_bytes(count) bytes
bool keyswitch_mfma_supported(uint32_t in_dim, uint32_t base_log, uint32_t level);
size_t keyswitch_mfma_limb_bytes(uint32_t in_dim, uint32_t out_dim, uint32_t level, int torus_bytes);
size_t keyswitch_mfma_digit_bytes(uint32_t in_dim, uint32_t level, size_t count);
template <typename T>
int keyswitch_mfma_prepare(const T* ksk, int8_t* limbs, uint32_t in_dim, uint32_t out_dim, uint32_t level, hipStream_t stream);
template <typename T>
int launch_keyswitch_mfma(const LweInput<T>& in, T* ct_out, size_t out_stride, const int8_t* limbs,
                          int8_t* digits, uint32_t in_dim, uint32_t out_dim, uint32_t base_log, uint32_t level, size_t count,
                          hipStream_t stream);

// ---- measurement helper (probe.hip): sustained v_fma_f64 rate of the current device ----
int measure_fp64_fma_rate(hipStream_t stream, double* tflops);

// ---- leveled operations on device ciphertexts (linear.hip) ----
template <typename T>
int launch_lwe_axpby(T* out, size_t out_stride, const T* x, size_t x_stride, T alpha, const T* y, size_t y_stride, T beta,
                     T body_add, size_t words, size_t count, hipStream_t stream);

// ---- workgroup-per-ciphertext bootstrap, any parameter set (pbs_generic.hip) ----
// tables: twist[M] then stage twiddles W_M^j for j < M/2 (host fills them in double precision)
size_t generic_tables_count(uint32_t N);                    // number of double2 entries
void generic_fill_tables(uint32_t N, double2* host_tables);
size_t generic_scratch_bytes(uint32_t k, uint32_t N, int torus_bytes, unsigned* grid_out, size_t count);
template <typename T>
int generic_bsk_to_fourier(const T* bsk_std, double2* fbsk, const double2* tables, uint32_t N, size_t polys,
                           hipStream_t stream);
template <typename T>
int generic_launch_pbs(const PbsLaunch<T>& a, hipStream_t stream);

// ---- layout of the Fourier bootstrap key, per kernel family ----
// Every compile-time choice that decides where a spectrum value lies in the key blob (points per
// thread, lane renumbering, register/lane placement) plus a revision number that is raised with
// any other change of the order. A blob is only valid for a library whose tag equals the one it
// was exported under (camd_fourier_layout_id).
uint64_t generic_layout_tag(const camd_params& p);
uint64_t block_layout_tag(const camd_params& p);
uint64_t wave_layout_tag(const camd_params& p);

// ---- register-blocked workgroup-per-ciphertext bootstrap (pbs_block.hip) ----
// uses the tables of the generic family; results are bit-identical to it
bool block_supported(const camd_params& p);
size_t block_scratch_bytes(const camd_params& p, unsigned* grid_out, size_t count);
template <typename T>
int block_bsk_to_fourier(const T* bsk_std, double2* fbsk, const double2* tables, uint32_t N, size_t polys,
                         uint32_t levels, hipStream_t stream);
template <typename T>
int block_launch_pbs(const PbsLaunch<T>& a, unsigned grid, hipStream_t stream);

// ---- wave-per-ciphertext bootstrap (pbs_wave.hip): N = 1024 (64-bit, k = 2, one level; 32-bit, k = 1,
// three levels) and N = 512 (64-bit, k = 3, one level; 32-bit, k = 2, two levels) ----
bool wave_supported(const camd_params& p);   // an instance exists (CAMD_PBS_WAVE may be requested)
bool wave_default(const camd_params& p);     // ... and CAMD_PBS_AUTO selects it
size_t wave_tables_count(uint32_t N);
// the N = 512 instances (pbs_wave512.hip), reached through the wave_* functions
bool wave512_supported(const camd_params& p);
size_t wave512_tables_count();
void wave512_fill_tables(double2* t);
template <typename T>
int wave512_bsk_to_fourier(const T* bsk_std, double2* fbsk, const double2* tables, size_t polys, hipStream_t stream);
template <typename T>
int wave512_launch_pbs(const PbsLaunch<T>& a, hipStream_t stream);
// scratch of a launch of `count` ciphertexts: the arrival counters of the persistent waves (cleared by
// every launch) or, for the N = 16384 instances, one accumulator slab per resident workgroup
size_t wave_scratch_bytes(const camd_params& p, size_t count);
bool wave_scratch_is_fixed(const camd_params& p);   // the scratch does not grow with the batch
// the N = 16384 instances (pbs_wave16k.hip), reached through the wave_* functions
bool wave16k_supported(const camd_params& p);
size_t wave16k_tables_count();
void wave16k_fill_tables(uint32_t N, double2* t);
size_t wave16k_scratch_bytes(uint32_t N, unsigned* grid_out, size_t count, int cus);
template <typename T>
int wave16k_bsk_to_fourier(const T* bsk_std, double2* fbsk, const double2* tables, uint32_t N, size_t polys, hipStream_t stream);
template <typename T>
int wave16k_launch_pbs(const PbsLaunch<T>& a, hipStream_t stream);
// the schedule with two waves per SIMD and a 32-bit accumulator (pbs_wave8.hip), reached through wave_launch_pbs
bool wave8_supported(uint32_t torus_bits, uint32_t N, uint32_t k, uint32_t level, uint32_t base_log);
template <typename T>
int wave8_launch_pbs(const PbsLaunch<T>& a, hipStream_t stream);
void wave_fill_tables(uint32_t N, double2* host_tables);
template <typename T>
int wave_bsk_to_fourier(const T* bsk_std, double2* fbsk, const double2* tables, const camd_params& p,
                        hipStream_t stream);
template <typename T>
int wave_launch_pbs(const PbsLaunch<T>& a, hipStream_t stream);

}  // namespace camd
