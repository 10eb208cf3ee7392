// C ABI of the backend: contexts, server keys, look-up tables, batch entry points.
// The header (include/concrete_amd.h) documents which reference interface each entry replaces.
#include <cstdint>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <initializer_list>
#include <mutex>
#include <new>
#include <vector>

#include "kernels.hpp"

namespace camd {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

}  // namespace camd

using namespace camd;

// Scratch of the entry point that allocates nothing (camd_ks_pbs_batch_device_noalloc): a
// file-scope device array, so it exists from the moment the code object is loaded on a GPU and
// costs no hipMalloc. Every GPU the library runs on holds its own copy, at its own address, until
// the process ends. Layout: [arrival / member counters of the wave kernels | keyswitch output].
constexpr size_t NOALLOC_COUNTER_BYTES = 4096;
constexpr size_t NOALLOC_MID_BYTES = (size_t)16 << 20;
constexpr size_t NOALLOC_BYTES = NOALLOC_COUNTER_BYTES + NOALLOC_MID_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

struct camd_context {
  int device = 0;
  hipStream_t stream = nullptr;    // kernels of the host-pointer entry points
  hipStream_t copy_in = nullptr;   // their host -> device copies
  hipStream_t copy_out = nullptr;  // their device -> host copies
  // address of `scratch` on THIS context's GPU (hipGetSymbolAddress at context creation). Host
  // code must use this pointer, never the array's name, which is only a host-side stand-in.
  unsigned char* noalloc_scratch = nullptr;
};

namespace {

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return CAMD_OK;
    if (ptr) {
      CAMD_HIP(hipDeviceSynchronize());
      CAMD_HIP(hipFree(ptr));
      ptr = nullptr;
      bytes = 0;
    }
    CAMD_HIP(hipMalloc(&ptr, need));
    bytes = need;
    return CAMD_OK;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
};

struct Timing {
  bool enabled = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pbs, ks;
  void clear() {
    for (auto& e : pbs) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& e : ks) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    pbs.clear();
    ks.clear();
  }
};

enum class Op { KsPbs, PbsKs, Pbs, Ks };

// A linear combination formed ahead of the first kernel of an operation: that kernel's input rows
// are alpha * x + beta * y, body_add added to the body (kernels.hpp: LweInput). Strides in words.
struct Combination {
  size_t x_stride;
  const void* y;
  size_t y_stride;
  uint64_t alpha, beta, body_add;
};

// One blocked caller of a host-pointer entry point.
struct HostRequest {
  Op op;
  const void* in;
  const uint32_t* ids;
  void* out;
  size_t count;
  int rc = CAMD_OK;
  std::string err;
  bool done = false;
};

}  // namespace

struct camd_server_key {
  camd_context* ctx = nullptr;
  camd_params p{};
  int variant = CAMD_PBS_GENERIC;
  size_t word = 8;          // bytes per torus word
  // one blob: [fourier bsk | ksk]
  void* blob = nullptr;
  size_t fbsk_bytes = 0, ksk_bytes = 0;
  double2* tables = nullptr;
  // look-up tables
  DeviceBuffer luts;
  uint32_t lut_count = 0;
  // scratch
  DeviceBuffer mid;         // keyswitch output (shortint order) or bootstrap output (boolean order)
  DeviceBuffer lin;         // a combination written out, for the bootstrap kernels that read plain batches only
  DeviceBuffer pbs_scratch;
  DeviceBuffer ks_limbs;    // keyswitch key as balanced byte limbs (matrix-core keyswitch), derived from the blob
  DeviceBuffer ks_digits;   // digits of a batch (matrix-core keyswitch)
  DeviceBuffer stage_in, stage_out, stage_ids;
  std::mutex mu;            // guards scratch, look-up tables, timing and the launch order of both kinds of entry point
  Timing timing;
  // launch status word in pinned host memory, written by the kernels (kernels.hpp: STATUS_*)
  uint32_t* status = nullptr;
  // Device-pointer calls return while their kernels are still running on the caller's stream; they
  // share `mid` and `pbs_scratch` with every other call on the key. The event marks the end of the
  // last such call: a later call on another stream (the host-pointer entry points use the
  // context's own) makes its stream wait for it before it touches the shared buffers.
  hipEvent_t device_done = nullptr;
  hipStream_t device_stream = nullptr;
  bool device_pending = false;
  // host callers that arrive while a batch is on the GPU wait here and leave together, as one batch
  std::mutex queue_mu;
  std::condition_variable queue_cv;
  std::deque<HostRequest*> queue;
  bool serving = false;

  const double2* fbsk() const { return (const double2*)blob; }
  const void* ksk() const { return (const char*)blob + fbsk_bytes; }
  size_t big() const { return (size_t)p.glwe_dimension * p.polynomial_size; }
  size_t glwe_words() const { return (size_t)(p.glwe_dimension + 1) * p.polynomial_size; }
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int validate_params(const camd_params& p) {
  if (p.torus_bits != 32 && p.torus_bits != 64) return fail(CAMD_EINVAL, "torus_bits must be 32 or 64");
  const uint32_t N = p.polynomial_size;
  if (N < 256 || N > 16384 || (N & (N - 1))) return fail(CAMD_EINVAL, "polynomial_size must be a power of two in [256, 16384]");
  if (p.lwe_dimension == 0 || p.glwe_dimension == 0 || p.glwe_dimension > 8) return fail(CAMD_EINVAL, "bad lwe/glwe dimension");
  if (p.pbs_level == 0 || p.pbs_base_log == 0 || p.pbs_base_log * p.pbs_level >= p.torus_bits || p.pbs_base_log > 31)
    return fail(CAMD_EINVAL, "bad bootstrap decomposition (need 0 < base_log*level < torus_bits, base_log <= 31)");
  if (p.ks_level == 0 || p.ks_level > 8 || p.ks_base_log == 0 || p.ks_base_log * p.ks_level >= p.torus_bits || p.ks_base_log > 31)
    return fail(CAMD_EINVAL, "bad keyswitch decomposition (need 0 < base_log*level < torus_bits, level <= 8)");
  return CAMD_OK;
}

size_t fbsk_bytes_of(const camd_params& p) {
  return (size_t)p.lwe_dimension * p.pbs_level * (p.glwe_dimension + 1) * (p.glwe_dimension + 1) * (p.polynomial_size / 2) *
         sizeof(double2);
}
size_t ksk_bytes_of(const camd_params& p) {
  return (size_t)p.glwe_dimension * p.polynomial_size * p.ks_level * (p.lwe_dimension + 1) * (p.torus_bits / 8);
}

int pick_variant(const camd_params& p, int requested, int* out) {
  if (requested == CAMD_PBS_GENERIC) { *out = CAMD_PBS_GENERIC; return CAMD_OK; }
  if (requested == CAMD_PBS_WAVE) {
    if (!wave_supported(p)) return fail(CAMD_EINVAL, "the wave-per-polynomial kernel has no instance for this parameter set");
    *out = CAMD_PBS_WAVE;
    return CAMD_OK;
  }
  if (requested == CAMD_PBS_BLOCK) {
    if (!block_supported(p)) return fail(CAMD_EINVAL, "the register-blocked kernel has no instance for this parameter set");
    *out = CAMD_PBS_BLOCK;
    return CAMD_OK;
  }
  if (requested != CAMD_PBS_AUTO) return fail(CAMD_EINVAL, "unknown pbs_variant");
  // AUTO takes the wave kernel wherever it has an instance (its results agree with the bit-exact
  // families at the decryption level, not bit for bit)
  *out = wave_default(p) ? CAMD_PBS_WAVE : block_supported(p) ? CAMD_PBS_BLOCK : CAMD_PBS_GENERIC;
  return CAMD_OK;
}

// allocates blob + tables; leaves the Fourier key uninitialised
int key_alloc(camd_context* ctx, const camd_params& p, int variant, camd_server_key** out) {
  camd_server_key* key = new (std::nothrow) camd_server_key();
  if (!key) return fail(CAMD_ENOMEM, "out of host memory");
  key->ctx = ctx;
  key->p = p;
  key->variant = variant;
  key->word = p.torus_bits / 8;
  key->fbsk_bytes = fbsk_bytes_of(p);
  key->ksk_bytes = ksk_bytes_of(p);
  hipError_t e = hipMalloc(&key->blob, key->fbsk_bytes + key->ksk_bytes);
  if (e != hipSuccess) {
    delete key;
    return fail(CAMD_ENOMEM, std::string("hipMalloc(key blob): ") + hipGetErrorString(e));
  }
  e = hipHostMalloc((void**)&key->status, 64, hipHostMallocMapped);
  if (e == hipSuccess) {
    *key->status = 0;
    e = hipEventCreateWithFlags(&key->device_done, hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    (void)hipFree(key->blob);
    if (key->status) (void)hipHostFree(key->status);
    delete key;
    return fail(CAMD_EHIP, std::string("status word / event: ") + hipGetErrorString(e));
  }
  const size_t tcount = (variant == CAMD_PBS_WAVE) ? wave_tables_count(p.polynomial_size) : generic_tables_count(p.polynomial_size);
  std::vector<double2> host(tcount);
  if (variant == CAMD_PBS_WAVE) wave_fill_tables(p.polynomial_size, host.data());
  else generic_fill_tables(p.polynomial_size, host.data());
  e = hipMalloc((void**)&key->tables, tcount * sizeof(double2));
  if (e == hipSuccess) e = hipMemcpy(key->tables, host.data(), tcount * sizeof(double2), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(key->blob);
    if (key->tables) (void)hipFree(key->tables);
    (void)hipHostFree(key->status);
    (void)hipEventDestroy(key->device_done);
    delete key;
    return fail(CAMD_EHIP, std::string("twiddle table upload: ") + hipGetErrorString(e));
  }
  *out = key;
  return CAMD_OK;
}

int key_derive(camd_server_key* key, hipStream_t s);

template <typename T>
int key_fill(camd_server_key* key, const void* bsk_host, const void* ksk_host) {
  const camd_params& p = key->p;
  hipStream_t s = key->ctx->stream;
  const size_t polys = (size_t)p.lwe_dimension * p.pbs_level * (p.glwe_dimension + 1) * (p.glwe_dimension + 1);
  const size_t std_bytes = polys * p.polynomial_size * sizeof(T);
  void* d_std = nullptr;
  CAMD_HIP(hipMalloc(&d_std, std_bytes));
  hipError_t e = hipMemcpyAsync(d_std, bsk_host, std_bytes, hipMemcpyHostToDevice, s);
  int rc = CAMD_OK;
  if (e != hipSuccess) rc = fail(CAMD_EHIP, std::string("bsk upload: ") + hipGetErrorString(e));
  if (rc == CAMD_OK) {
    if (key->variant == CAMD_PBS_WAVE) rc = wave_bsk_to_fourier<T>((const T*)d_std, (double2*)key->blob, key->tables, p, s);
    else if (key->variant == CAMD_PBS_BLOCK)
      rc = block_bsk_to_fourier<T>((const T*)d_std, (double2*)key->blob, key->tables, p.polynomial_size, polys,
                                   p.pbs_level, s);
    else rc = generic_bsk_to_fourier<T>((const T*)d_std, (double2*)key->blob, key->tables, p.polynomial_size, polys, s);
  }
  if (rc == CAMD_OK) {
    e = hipMemcpyAsync((char*)key->blob + key->fbsk_bytes, ksk_host, key->ksk_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = fail(CAMD_EHIP, std::string("ksk upload: ") + hipGetErrorString(e));
    if (rc == CAMD_OK) rc = key_derive(key, s);
  } else {
    (void)hipStreamSynchronize(s);
  }
  (void)hipFree(d_std);
  return rc;
}

// Which keyswitch kernel serves a batch. The matrix-core form is faster for every batch size
// measured (2_2: 0.036 against 0.073 ms for one ciphertext, 0.91 against 6.2 ms for 2^14) except
// the smallest batches of the N = 16384 sets, whose 98304 key rows the f64-FMA kernel spreads over
// the chip slightly better (0.24 against 0.30 ms below 32 ciphertexts). CAMD_KS_MFMA_MIN overrides
// the smallest batch that takes the matrix cores (0 = never).
constexpr size_t KS_MFMA_MAX_COUNT = 65535u * 16u;   // launch_keyswitch_mfma's grid limit
bool ks_use_mfma(const camd_server_key* key, size_t count) {
  if (!key->ks_limbs.ptr) return false;
  const char* e = std::getenv("CAMD_KS_MFMA_MIN");
  if (e && *e) {
    const long v = std::strtol(e, nullptr, 10);
    return v > 0 && count >= (size_t)v;
  }
  if (count > KS_MFMA_MAX_COUNT) return false;   // beyond one launch of the matrix-core kernel: the f64-FMA form has no limit
  const size_t rows = key->big() * key->p.ks_level;
  return rows <= 32768 || count >= 32;
}

// the keyswitch key of the blob, split into balanced byte limbs for the matrix cores
int key_derive(camd_server_key* key, hipStream_t s) {
  const camd_params& p = key->p;
  const uint32_t big = (uint32_t)key->big();
  if (!keyswitch_mfma_supported(big, p.ks_base_log, p.ks_level)) return CAMD_OK;
  int rc = key->ks_limbs.ensure(keyswitch_mfma_limb_bytes(big, p.lwe_dimension, p.ks_level, (int)key->word));
  if (rc != CAMD_OK) return rc;
  rc = p.torus_bits == 64
           ? keyswitch_mfma_prepare<uint64_t>((const uint64_t*)key->ksk(), (int8_t*)key->ks_limbs.ptr, big, p.lwe_dimension, p.ks_level, s)
           : keyswitch_mfma_prepare<uint32_t>((const uint32_t*)key->ksk(), (int8_t*)key->ks_limbs.ptr, big, p.lwe_dimension, p.ks_level, s);
  if (rc != CAMD_OK) return rc;
  CAMD_HIP(hipStreamSynchronize(s));
  return CAMD_OK;
}

void key_free(camd_server_key* key) {
  if (!key) return;
  key->timing.clear();
  if (key->blob) (void)hipFree(key->blob);
  if (key->tables) (void)hipFree(key->tables);
  if (key->status) (void)hipHostFree(key->status);
  if (key->device_done) (void)hipEventDestroy(key->device_done);
  key->luts.release();
  key->mid.release();
  key->lin.release();
  key->pbs_scratch.release();
  key->ks_limbs.release();
  key->ks_digits.release();
  key->stage_in.release();
  key->stage_out.release();
  key->stage_ids.release();
  delete key;
}

int reserve_locked(camd_server_key* key, size_t count) {
  const camd_params& p = key->p;
  const size_t mid_words = std::max<size_t>(key->big() + 1, p.lwe_dimension + 1);
  int rc = key->mid.ensure(count * mid_words * key->word);
  if (rc != CAMD_OK) return rc;
  if (ks_use_mfma(key, count)) {
    // Digit scratch of the matrix-core keyswitch (count x kN x level bytes). If the device cannot
    // spare it the call still succeeds: run_ks sees the short buffer and takes the f64-FMA kernel,
    // which needs no scratch.
    if (key->ks_digits.ensure(keyswitch_mfma_digit_bytes((uint32_t)key->big(), p.ks_level, count)) != CAMD_OK)
      (void)hipGetLastError();   // the failed allocation must not surface as the next launch's error
  }
  if (key->variant == CAMD_PBS_GENERIC) {
    unsigned grid = 1;
    const size_t bytes = generic_scratch_bytes(p.glwe_dimension, p.polynomial_size, (int)key->word, &grid, count);
    rc = key->pbs_scratch.ensure(bytes);
  } else if (key->variant == CAMD_PBS_BLOCK) {
    unsigned grid = 1;
    rc = key->pbs_scratch.ensure(block_scratch_bytes(p, &grid, count));
  } else if (key->variant == CAMD_PBS_WAVE) {
    rc = key->pbs_scratch.ensure(wave_scratch_bytes(p, count));
  }
  return rc;
}

struct TimedRegion {
  std::vector<std::pair<hipEvent_t, hipEvent_t>>* list = nullptr;
  hipStream_t s = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  TimedRegion(camd_server_key* key, bool is_pbs, hipStream_t stream) {
    if (!key->timing.enabled) return;
    list = is_pbs ? &key->timing.pbs : &key->timing.ks;
    s = stream;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { list = nullptr; return; }
    (void)hipEventRecord(a, s);
  }
  ~TimedRegion() {
    if (!list) return;
    (void)hipEventRecord(b, s);
    list->push_back({a, b});
  }
};

template <typename T>
LweInput<T> input_of(const void* x, size_t dense_stride, uint32_t dimension, const Combination* c) {
  if (!c) return LweInput<T>::plain((const T*)x, dense_stride, dimension);
  return LweInput<T>{(const T*)x, c->x_stride, (const T*)c->y, c->y_stride, (T)c->alpha, (T)c->beta, (T)c->body_add, dimension};
}

// The bootstrap kernels that form an LweInput while they read (kernels.hpp: input_row): the
// workgroup-per-ciphertext families. For the wave family run_pbs writes the combination out first:
// its kernels fetch the next mask element as one scalar a whole iteration ahead, and forming it
// from two operands inside that loop cost the N = 512 instances 4 % (32-bit, two levels) to 39 %
// (64-bit, one level) of their time, against 0.05 % for the pass that writes (n + 1) words per
// ciphertext.
bool pbs_family_combines(const camd_server_key* key) { return key->variant != CAMD_PBS_WAVE; }

template <typename T>
int run_pbs(camd_server_key* key, LweInput<T> in, const uint32_t* ids, T* out, size_t out_stride,
            size_t count, hipStream_t s, void* scratch_override = nullptr) {
  if (key->lut_count == 0) return fail(CAMD_EINVAL, "no look-up table uploaded for this key");
  const camd_params& p = key->p;
  if (!in.is_plain() && !pbs_family_combines(key)) {
    if (scratch_override) return fail(CAMD_EINVAL, "internal: a combination on the path that allocates nothing");
    const size_t words = (size_t)p.lwe_dimension + 1;
    int rc = key->lin.ensure(count * words * sizeof(T));
    if (rc != CAMD_OK) return rc;
    rc = launch_lwe_axpby<T>((T*)key->lin.ptr, words, in.x, in.x_stride, in.alpha, in.y, in.y_stride, in.beta, in.body_add,
                             words, count, s);
    if (rc != CAMD_OK) return rc;
    in = LweInput<T>::plain((const T*)key->lin.ptr, words, p.lwe_dimension);
  }
  PbsLaunch<T> a{};
  a.ct_in = in.x; a.in_stride = in.x_stride; a.ct_out = out; a.out_stride = out_stride;
  a.y_in = in.y; a.y_stride = in.y_stride; a.alpha = in.alpha; a.beta = in.beta; a.body_add = in.body_add;
  a.lut_ids = ids; a.luts = (const T*)key->luts.ptr; a.lut_count = key->lut_count; a.count = count;
  a.fbsk = key->fbsk(); a.tables = key->tables; a.status = key->status;
  a.scratch = scratch_override ? scratch_override : key->pbs_scratch.ptr;
  a.n = p.lwe_dimension; a.k = p.glwe_dimension; a.N = p.polynomial_size; a.level = p.pbs_level; a.base_log = p.pbs_base_log;
  if (scratch_override) return wave_launch_pbs<T>(a, s);   // the caller checked the family; no events are created here
  TimedRegion t(key, true, s);
  if (key->variant == CAMD_PBS_WAVE) return wave_launch_pbs<T>(a, s);
  if (key->variant == CAMD_PBS_BLOCK) {
    unsigned grid = 1;
    block_scratch_bytes(p, &grid, count);
    return block_launch_pbs<T>(a, grid, s);
  }
  return generic_launch_pbs<T>(a, s);
}

template <typename T>
int run_ks(camd_server_key* key, const LweInput<T>& in, T* out, size_t out_stride, size_t count, hipStream_t s,
           bool timed = true) {
  const camd_params& p = key->p;
  if (!timed)   // the entry point that allocates nothing: the kernel that needs no digit scratch
    return launch_keyswitch<T>(in, out, out_stride, (const T*)key->ksk(), (uint32_t)key->big(), p.lwe_dimension,
                               p.ks_base_log, p.ks_level, count, s);
  TimedRegion t(key, false, s);
  if (ks_use_mfma(key, count) && key->ks_digits.bytes >= keyswitch_mfma_digit_bytes((uint32_t)key->big(), p.ks_level, count))
    return launch_keyswitch_mfma<T>(in, out, out_stride, (const int8_t*)key->ks_limbs.ptr, (int8_t*)key->ks_digits.ptr,
                                    (uint32_t)key->big(), p.lwe_dimension, p.ks_base_log, p.ks_level, count, s);
  return launch_keyswitch<T>(in, out, out_stride, (const T*)key->ksk(), (uint32_t)key->big(), p.lwe_dimension,
                             p.ks_base_log, p.ks_level, count, s);
}

size_t in_words(const camd_server_key* k, Op op) {
  return (op == Op::KsPbs || op == Op::Ks) ? k->big() + 1 : (size_t)k->p.lwe_dimension + 1;
}
size_t out_words(const camd_server_key* k, Op op) {
  return (op == Op::KsPbs || op == Op::Pbs) ? k->big() + 1 : (size_t)k->p.lwe_dimension + 1;
}

template <typename T>
int run_op_device(camd_server_key* key, Op op, const void* in, const uint32_t* ids, void* out, size_t count, hipStream_t s,
                  const Combination* comb = nullptr) {
  int rc = reserve_locked(key, count);
  if (rc != CAMD_OK) return rc;
  const uint32_t big = (uint32_t)key->big(), small = key->p.lwe_dimension;
  const size_t big1 = (size_t)big + 1, small1 = (size_t)small + 1;
  const LweInput<T> mid_small = LweInput<T>::plain((const T*)key->mid.ptr, small1, small);
  const LweInput<T> mid_big = LweInput<T>::plain((const T*)key->mid.ptr, big1, big);
  switch (op) {
    case Op::KsPbs:
      rc = run_ks<T>(key, input_of<T>(in, big1, big, comb), (T*)key->mid.ptr, small1, count, s);
      if (rc == CAMD_OK) rc = run_pbs<T>(key, mid_small, ids, (T*)out, big1, count, s);
      break;
    case Op::PbsKs:
      rc = run_pbs<T>(key, input_of<T>(in, small1, small, comb), ids, (T*)key->mid.ptr, big1, count, s);
      if (rc == CAMD_OK) rc = run_ks<T>(key, mid_big, (T*)out, small1, count, s);
      break;
    case Op::Pbs:
      rc = run_pbs<T>(key, input_of<T>(in, small1, small, comb), ids, (T*)out, big1, count, s);
      break;
    case Op::Ks:
      rc = run_ks<T>(key, input_of<T>(in, big1, big, comb), (T*)out, small1, count, s);
      break;
  }
  return rc;
}

// A bad look-up table id seen by a kernel of an earlier device-pointer call (kernels.hpp:
// checked_lut_id). The word is in host memory; it is certain to be up to date once the stream of
// that call has been synchronised, and is looked at by every later call on the key.
int report_status(camd_server_key* key) {
  if (!key->status || *(volatile uint32_t*)key->status == 0) return CAMD_OK;
  *(volatile uint32_t*)key->status = 0;
  return fail(CAMD_EINVAL,
              "an earlier device-pointer call on this key had a look-up table id out of range; those ciphertexts were "
              "bootstrapped with table 0");
}

// makes `s` wait for the kernels of the last device-pointer call if they run on another stream
int order_after_device_calls(camd_server_key* key, hipStream_t s) {
  if (key->device_pending && key->device_stream != s) CAMD_HIP(hipStreamWaitEvent(s, key->device_done, 0));
  return CAMD_OK;
}

int dispatch_device(camd_server_key* key, Op op, const void* in, const uint32_t* ids, void* out, size_t count, void* stream,
                    const Combination* comb = nullptr) {
  if (!key || (!in && count) || (!out && count)) return fail(CAMD_EINVAL, "null argument");
  if (comb && count) {
    const size_t words = in_words(key, op);
    if (comb->x_stride < words || (comb->y && comb->y_stride < words)) return fail(CAMD_EINVAL, "row stride shorter than the ciphertext");
  }
  std::lock_guard<std::mutex> lock(key->mu);
  int rc = report_status(key);
  if (rc != CAMD_OK) return rc;
  if (count == 0) return CAMD_OK;
  DeviceGuard g(key->ctx->device);
  if (!g.ok) return fail(CAMD_EHIP, "hipSetDevice failed");
  hipStream_t s = (hipStream_t)stream;
  rc = order_after_device_calls(key, s);
  if (rc != CAMD_OK) return rc;
  rc = key->p.torus_bits == 64 ? run_op_device<uint64_t>(key, op, in, ids, out, count, s, comb)
                               : run_op_device<uint32_t>(key, op, in, ids, out, count, s, comb);
  // recorded even after a failed launch sequence: whatever was enqueued must be waited for
  CAMD_HIP(hipEventRecord(key->device_done, s));
  key->device_stream = s;
  key->device_pending = true;
  return rc;
}

// Ciphertexts per slice of a host-buffer batch: the copies of one slice run beside the kernels
// of its neighbours. CAMD_HOST_CHUNK overrides it (0 = never slice).
size_t host_chunk() {
  const char* e = std::getenv("CAMD_HOST_CHUNK");
  if (!e || !*e) return 2048;
  const long v = std::strtol(e, nullptr, 10);
  return v <= 0 ? SIZE_MAX : (size_t)v;
}

struct EventList {
  std::vector<hipEvent_t> ev;
  hipError_t add(hipEvent_t* out) {
    hipError_t e = hipEventCreateWithFlags(out, hipEventDisableTiming);
    if (e == hipSuccess) ev.push_back(*out);
    return e;
  }
  ~EventList() {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }
};

template <typename T>
int run_op_host(camd_server_key* key, Op op, const void* in, const uint32_t* d_ids, void* out, size_t count) {
  camd_context* ctx = key->ctx;
  const size_t iw = in_words(key, op) * key->word, ow = out_words(key, op) * key->word;
  const size_t chunk = host_chunk();
  // Slice boundaries. The copy of the first slice in and of the last slice out have nothing to
  // hide behind, so for a batch of at least four slices those two are half a slice; a short tail
  // is folded into the slice before it.
  std::vector<size_t> bounds{0};
  if (count / 2 < chunk) {
    bounds.push_back(count);
  } else if (count >= 4 * chunk && chunk >= 2) {
    bounds.push_back(chunk / 2);
    while (count - bounds.back() >= chunk + chunk / 2 + chunk / 2) bounds.push_back(bounds.back() + chunk);
    // what is left is between one and two slices: the last half slice takes the end of it
    if (count - bounds.back() > chunk / 2) bounds.push_back(count - chunk / 2);
    bounds.push_back(count);
  } else {
    for (size_t i = 1; i < count / chunk; i++) bounds.push_back(i * chunk);
    bounds.push_back(count);
  }
  const size_t slices = bounds.size() - 1;
  EventList events;
  hipEvent_t computed_prev = nullptr;
  size_t prev_first = 0, prev_n = 0;
  int rc = CAMD_OK;
  hipError_t e = hipSuccess;
  for (size_t i = 0; i < slices && rc == CAMD_OK && e == hipSuccess; i++) {
    const size_t first = bounds[i], n = bounds[i + 1] - bounds[i];
    hipEvent_t arrived = nullptr, computed = nullptr;
    e = events.add(&arrived);
    if (e == hipSuccess) e = events.add(&computed);
    if (e == hipSuccess)
      e = hipMemcpyAsync((char*)key->stage_in.ptr + first * iw, (const char*)in + first * iw, n * iw, hipMemcpyHostToDevice,
                         ctx->copy_in);
    if (e == hipSuccess) e = hipEventRecord(arrived, ctx->copy_in);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, arrived, 0);
    if (e != hipSuccess) break;
    rc = run_op_device<T>(key, op, (char*)key->stage_in.ptr + first * iw, d_ids ? d_ids + first : nullptr,
                          (char*)key->stage_out.ptr + first * ow, n, ctx->stream);
    if (rc != CAMD_OK) break;
    e = hipEventRecord(computed, ctx->stream);
    // the previous slice goes home while this one computes
    if (e == hipSuccess && computed_prev) {
      e = hipStreamWaitEvent(ctx->copy_out, computed_prev, 0);
      if (e == hipSuccess)
        e = hipMemcpyAsync((char*)out + prev_first * ow, (char*)key->stage_out.ptr + prev_first * ow, prev_n * ow,
                           hipMemcpyDeviceToHost, ctx->copy_out);
    }
    computed_prev = computed; prev_first = first; prev_n = n;
  }
  if (rc == CAMD_OK && e == hipSuccess) {
    e = hipStreamWaitEvent(ctx->copy_out, computed_prev, 0);
    if (e == hipSuccess)
      e = hipMemcpyAsync((char*)out + prev_first * ow, (char*)key->stage_out.ptr + prev_first * ow, prev_n * ow,
                         hipMemcpyDeviceToHost, ctx->copy_out);
  }
  // nothing of this call may be in flight when it returns, whatever happened
  const hipError_t e1 = hipStreamSynchronize(ctx->copy_in);
  const hipError_t e2 = hipStreamSynchronize(ctx->stream);
  const hipError_t e3 = hipStreamSynchronize(ctx->copy_out);
  if (rc != CAMD_OK) return rc;
  for (hipError_t x : {e, e1, e2, e3})
    if (x != hipSuccess) return fail(CAMD_EHIP, std::string("host-buffer batch: ") + hipGetErrorString(x));
  return CAMD_OK;
}

// One caller's batch, alone on the GPU.
int serve_one(camd_server_key* key, const HostRequest& r) {
  const size_t ib = r.count * in_words(key, r.op) * key->word, ob = r.count * out_words(key, r.op) * key->word;
  int rc = key->stage_in.ensure(ib);
  if (rc == CAMD_OK) rc = key->stage_out.ensure(ob);
  if (rc == CAMD_OK && r.ids) rc = key->stage_ids.ensure(r.count * sizeof(uint32_t));
  // scratch is sized once, for the whole batch, before any slice is in flight
  if (rc == CAMD_OK) rc = reserve_locked(key, r.count);
  if (rc != CAMD_OK) return rc;
  if (r.ids) CAMD_HIP(hipMemcpyAsync(key->stage_ids.ptr, r.ids, r.count * sizeof(uint32_t), hipMemcpyHostToDevice, key->ctx->stream));
  const uint32_t* d_ids = r.ids ? (const uint32_t*)key->stage_ids.ptr : nullptr;
  return key->p.torus_bits == 64 ? run_op_host<uint64_t>(key, r.op, r.in, d_ids, r.out, r.count)
                                 : run_op_host<uint32_t>(key, r.op, r.in, d_ids, r.out, r.count);
}

// Several callers' batches of one operation as one launch: inputs side by side in the staging
// buffer, outputs copied back to each caller's buffer.
int serve_many(camd_server_key* key, Op op, const std::vector<HostRequest*>& group) {
  const size_t iw = in_words(key, op) * key->word, ow = out_words(key, op) * key->word;
  size_t total = 0;
  bool any_ids = false;
  for (const HostRequest* r : group) {
    total += r->count;
    any_ids = any_ids || r->ids;
  }
  int rc = key->stage_in.ensure(total * iw);
  if (rc == CAMD_OK) rc = key->stage_out.ensure(total * ow);
  if (rc == CAMD_OK && any_ids) rc = key->stage_ids.ensure(total * sizeof(uint32_t));
  if (rc == CAMD_OK) rc = reserve_locked(key, total);
  if (rc != CAMD_OK) return rc;
  hipStream_t s = key->ctx->stream;
  hipError_t e = hipSuccess;
  size_t first = 0;
  for (const HostRequest* r : group) {
    if (e == hipSuccess) e = hipMemcpyAsync((char*)key->stage_in.ptr + first * iw, r->in, r->count * iw, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && any_ids) {
      uint32_t* d = (uint32_t*)key->stage_ids.ptr + first;   // no ids = table 0 for every ciphertext
      e = r->ids ? hipMemcpyAsync(d, r->ids, r->count * sizeof(uint32_t), hipMemcpyHostToDevice, s)
                 : hipMemsetAsync(d, 0, r->count * sizeof(uint32_t), s);
    }
    first += r->count;
  }
  if (e == hipSuccess) {
    const uint32_t* d_ids = any_ids ? (const uint32_t*)key->stage_ids.ptr : nullptr;
    rc = key->p.torus_bits == 64 ? run_op_device<uint64_t>(key, op, key->stage_in.ptr, d_ids, key->stage_out.ptr, total, s)
                                 : run_op_device<uint32_t>(key, op, key->stage_in.ptr, d_ids, key->stage_out.ptr, total, s);
  }
  first = 0;
  for (const HostRequest* r : group) {
    if (e == hipSuccess && rc == CAMD_OK)
      e = hipMemcpyAsync(r->out, (char*)key->stage_out.ptr + first * ow, r->count * ow, hipMemcpyDeviceToHost, s);
    first += r->count;
  }
  const hipError_t e1 = hipStreamSynchronize(s);
  if (rc != CAMD_OK) return rc;
  for (hipError_t x : {e, e1})
    if (x != hipSuccess) return fail(CAMD_EHIP, std::string("host-buffer batch: ") + hipGetErrorString(x));
  return CAMD_OK;
}

// Runs the requests of one operation that were waiting; fills in rc / err of each.
void serve(camd_server_key* key, Op op, std::vector<HostRequest*>& group) {
  std::lock_guard<std::mutex> lock(key->mu);
  DeviceGuard g(key->ctx->device);
  // kernels of an earlier device-pointer call may still be running on the caller's stream with
  // the shared scratch: the context's stream waits for them; this call ends synchronised, so the
  // reverse order needs nothing
  int order_rc = g.ok ? order_after_device_calls(key, key->ctx->stream) : CAMD_OK;
  if (order_rc == CAMD_OK && g.ok && key->device_pending) {
    // the status word of those calls is final once their stream has drained
    if (hipEventSynchronize(key->device_done) == hipSuccess) key->device_pending = false;
    order_rc = report_status(key);
  }
  const std::string order_err = g_last_error;
  std::vector<HostRequest*> good;
  for (HostRequest* r : group) {
    if (!g.ok) {
      r->rc = CAMD_EHIP;
      r->err = "hipSetDevice failed";
      continue;
    }
    if (order_rc != CAMD_OK) {
      r->rc = order_rc;
      r->err = order_err;
      continue;
    }
    bool ids_ok = true;
    if (r->ids)
      for (size_t i = 0; i < r->count && ids_ok; i++) ids_ok = r->ids[i] < key->lut_count;
    if (!ids_ok) {
      r->rc = CAMD_EINVAL;
      r->err = "lut id out of range";
      continue;
    }
    good.push_back(r);
  }
  if (good.empty()) return;
  const int rc = good.size() == 1 ? serve_one(key, *good[0]) : serve_many(key, op, good);
  for (HostRequest* r : good) {
    r->rc = rc;
    if (rc != CAMD_OK) r->err = g_last_error;
  }
}

// The reference calls this path one ciphertext at a time, from many threads (rayon in
// concrete-integer's *_parallelized operations). A call that finds the key busy queues its
// request; whoever is serving takes everything queued for one operation as ONE batch when the
// GPU comes free, so concurrent single-ciphertext calls cost one launch, not one launch each.
// A lone caller is served at once and alone, as before.
int dispatch_host(camd_server_key* key, Op op, const void* in, const uint32_t* ids, void* out, size_t count) {
  if (!key || (!in && count) || (!out && count)) return fail(CAMD_EINVAL, "null argument");
  if (count == 0) return CAMD_OK;
  HostRequest req{op, in, ids, out, count};
  std::unique_lock<std::mutex> q(key->queue_mu);
  key->queue.push_back(&req);
  while (!req.done && key->serving) key->queue_cv.wait(q);
  if (!req.done) {
    // nobody is serving: this caller does, until its own request is done
    key->serving = true;
    while (!req.done) {
      std::vector<HostRequest*> group;
      const Op group_op = key->queue.front()->op;
      for (auto it = key->queue.begin(); it != key->queue.end();) {
        if ((*it)->op == group_op) {
          group.push_back(*it);
          it = key->queue.erase(it);
        } else {
          ++it;
        }
      }
      q.unlock();
      serve(key, group_op, group);
      q.lock();
      for (HostRequest* r : group) r->done = true;
      key->queue_cv.notify_all();
    }
    // whoever still waits (their requests are in the queue) elects the next server
    key->serving = false;
    key->queue_cv.notify_all();
  }
  q.unlock();
  if (req.rc != CAMD_OK) set_error(req.err);
  return req.rc;
}

}  // namespace

extern "C" {

const char* camd_last_error(void) { return g_last_error.c_str(); }
const char* camd_version(void) { return "concrete_amd 0.1 (gfx950)"; }

int camd_context_create(int device_id, camd_context** out) {
  if (!out) return fail(CAMD_EINVAL, "null out pointer");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) return fail(CAMD_ENODEV, "no HIP device is visible");
  if (device_id < 0 || device_id >= ndev) return fail(CAMD_EINVAL, "device id out of range");
  hipDeviceProp_t prop;
  CAMD_HIP(hipGetDeviceProperties(&prop, device_id));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(CAMD_ENODEV, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
  DeviceGuard g(device_id);
  if (!g.ok) return fail(CAMD_EHIP, "hipSetDevice failed");
  camd_context* ctx = new (std::nothrow) camd_context();
  if (!ctx) return fail(CAMD_ENOMEM, "out of host memory");
  ctx->device = device_id;
  // address of the fixed scratch on this GPU (the device is current: DeviceGuard above)
  e = hipGetSymbolAddress((void**)&ctx->noalloc_scratch, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !ctx->noalloc_scratch) {
    delete ctx;
    return fail(CAMD_EHIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
  }
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking);
  if (e != hipSuccess) {
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_in) (void)hipStreamDestroy(ctx->copy_in);
    delete ctx;
    return fail(CAMD_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  }
  *out = ctx;
  return CAMD_OK;
}

int camd_context_destroy(camd_context* ctx) {
  if (!ctx) return CAMD_OK;
  DeviceGuard g(ctx->device);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->copy_in) (void)hipStreamDestroy(ctx->copy_in);
  if (ctx->copy_out) (void)hipStreamDestroy(ctx->copy_out);
  delete ctx;
  return CAMD_OK;
}

int camd_context_device(const camd_context* ctx, int* device_id) {
  if (!ctx || !device_id) return fail(CAMD_EINVAL, "null argument");
  *device_id = ctx->device;
  return CAMD_OK;
}

int camd_context_synchronize(camd_context* ctx) {
  if (!ctx) return fail(CAMD_EINVAL, "null context");
  DeviceGuard g(ctx->device);
  CAMD_HIP(hipDeviceSynchronize());
  return CAMD_OK;
}

int camd_server_key_create(camd_context* ctx, const camd_params* params, const void* bsk_standard, const void* ksk,
                           int pbs_variant, camd_server_key** out) {
  if (!ctx || !params || !bsk_standard || !ksk || !out) return fail(CAMD_EINVAL, "null argument");
  *out = nullptr;
  int rc = validate_params(*params);
  if (rc != CAMD_OK) return rc;
  int variant = CAMD_PBS_GENERIC;
  rc = pick_variant(*params, pbs_variant, &variant);
  if (rc != CAMD_OK) return rc;
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(CAMD_EHIP, "hipSetDevice failed");
  camd_server_key* key = nullptr;
  rc = key_alloc(ctx, *params, variant, &key);
  if (rc != CAMD_OK) return rc;
  rc = params->torus_bits == 64 ? key_fill<uint64_t>(key, bsk_standard, ksk) : key_fill<uint32_t>(key, bsk_standard, ksk);
  if (rc != CAMD_OK) {
    key_free(key);
    return rc;
  }
  *out = key;
  return CAMD_OK;
}

int camd_server_key_destroy(camd_server_key* key) {
  if (!key) return CAMD_OK;
  DeviceGuard g(key->ctx->device);
  (void)hipDeviceSynchronize();
  key_free(key);
  return CAMD_OK;
}

int camd_server_key_params(const camd_server_key* key, camd_params* out) {
  if (!key || !out) return fail(CAMD_EINVAL, "null argument");
  *out = key->p;
  return CAMD_OK;
}

int camd_server_key_variant(const camd_server_key* key, int* variant) {
  if (!key || !variant) return fail(CAMD_EINVAL, "null argument");
  *variant = key->variant;
  return CAMD_OK;
}

int camd_server_key_blob_size(const camd_server_key* key, size_t* bytes) {
  if (!key || !bytes) return fail(CAMD_EINVAL, "null argument");
  *bytes = key->fbsk_bytes + key->ksk_bytes;
  return CAMD_OK;
}

int camd_server_key_export_blob(const camd_server_key* key, void* device_dst, void* stream) {
  if (!key || !device_dst) return fail(CAMD_EINVAL, "null argument");
  DeviceGuard g(key->ctx->device);
  CAMD_HIP(hipMemcpyAsync(device_dst, key->blob, key->fbsk_bytes + key->ksk_bytes, hipMemcpyDefault, (hipStream_t)stream));
  return CAMD_OK;
}

int camd_fourier_layout_id(const camd_params* params, int pbs_variant, uint64_t* id) {
  if (!params || !id) return fail(CAMD_EINVAL, "null argument");
  int rc = validate_params(*params);
  if (rc != CAMD_OK) return rc;
  int variant = CAMD_PBS_GENERIC;
  rc = pick_variant(*params, pbs_variant, &variant);
  if (rc != CAMD_OK) return rc;
  const uint64_t tag = variant == CAMD_PBS_WAVE ? wave_layout_tag(*params)
                       : variant == CAMD_PBS_BLOCK ? block_layout_tag(*params) : generic_layout_tag(*params);
  // FNV-1a over (family, shape of a key polynomial set, family tag)
  const uint64_t words[] = {(uint64_t)variant, params->torus_bits, params->polynomial_size, params->glwe_dimension,
                            params->pbs_level, tag};
  uint64_t h = 0xcbf29ce484222325ull;
  for (uint64_t w : words)
    for (int b = 0; b < 8; b++) {
      h ^= (w >> (8 * b)) & 0xff;
      h *= 0x100000001b3ull;
    }
  *id = h;
  return CAMD_OK;
}

int camd_server_key_import_blob(camd_context* ctx, const camd_params* params, int pbs_variant, const void* device_src,
                                size_t bytes, void* stream, camd_server_key** out) {
  if (!ctx || !params || !device_src || !out) return fail(CAMD_EINVAL, "null argument");
  *out = nullptr;
  int rc = validate_params(*params);
  if (rc != CAMD_OK) return rc;
  int variant = CAMD_PBS_GENERIC;
  rc = pick_variant(*params, pbs_variant, &variant);
  if (rc != CAMD_OK) return rc;
  if (bytes != fbsk_bytes_of(*params) + ksk_bytes_of(*params)) return fail(CAMD_EINVAL, "blob size does not match the parameter set");
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(CAMD_EHIP, "hipSetDevice failed");
  camd_server_key* key = nullptr;
  rc = key_alloc(ctx, *params, variant, &key);
  if (rc != CAMD_OK) return rc;
  hipError_t e = hipMemcpyAsync(key->blob, device_src, bytes, hipMemcpyDefault, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) {
    key_free(key);
    return fail(CAMD_EHIP, std::string("blob copy: ") + hipGetErrorString(e));
  }
  rc = key_derive(key, (hipStream_t)stream);
  if (rc != CAMD_OK) {
    key_free(key);
    return rc;
  }
  *out = key;
  return CAMD_OK;
}

int camd_lut_upload(camd_server_key* key, const void* glwe_host, uint32_t* lut_id) {
  if (!key || !glwe_host || !lut_id) return fail(CAMD_EINVAL, "null argument");
  std::lock_guard<std::mutex> lock(key->mu);
  DeviceGuard g(key->ctx->device);
  const size_t one = key->glwe_words() * key->word;
  const size_t need = (size_t)(key->lut_count + 1) * one;
  if (need > key->luts.bytes) {
    // grow geometrically, keeping the tables already uploaded
    size_t cap = key->luts.bytes ? key->luts.bytes * 2 : one * 16;
    while (cap < need) cap *= 2;
    void* fresh = nullptr;
    CAMD_HIP(hipDeviceSynchronize());
    CAMD_HIP(hipMalloc(&fresh, cap));
    if (key->lut_count) {
      hipError_t e = hipMemcpy(fresh, key->luts.ptr, (size_t)key->lut_count * one, hipMemcpyDeviceToDevice);
      if (e != hipSuccess) {
        (void)hipFree(fresh);
        return fail(CAMD_EHIP, std::string("lut copy: ") + hipGetErrorString(e));
      }
    }
    key->luts.release();
    key->luts.ptr = fresh;
    key->luts.bytes = cap;
  }
  CAMD_HIP(hipMemcpy((char*)key->luts.ptr + (size_t)key->lut_count * one, glwe_host, one, hipMemcpyHostToDevice));
  *lut_id = key->lut_count++;
  return CAMD_OK;
}

int camd_lut_count(const camd_server_key* key, uint32_t* count) {
  if (!key || !count) return fail(CAMD_EINVAL, "null argument");
  *count = key->lut_count;
  return CAMD_OK;
}

int camd_ks_pbs_batch(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids, void* ct_out, size_t count) {
  return dispatch_host(key, Op::KsPbs, ct_in, lut_ids, ct_out, count);
}
int camd_pbs_ks_batch(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids, void* ct_out, size_t count) {
  return dispatch_host(key, Op::PbsKs, ct_in, lut_ids, ct_out, count);
}
int camd_pbs_batch(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids, void* ct_out, size_t count) {
  return dispatch_host(key, Op::Pbs, ct_in, lut_ids, ct_out, count);
}
int camd_ks_batch(camd_server_key* key, const void* ct_in, void* ct_out, size_t count) {
  return dispatch_host(key, Op::Ks, ct_in, nullptr, ct_out, count);
}

int camd_server_key_check(camd_server_key* key, void* stream) {
  if (!key) return fail(CAMD_EINVAL, "null key");
  std::lock_guard<std::mutex> lock(key->mu);
  DeviceGuard g(key->ctx->device);
  if (!g.ok) return fail(CAMD_EHIP, "hipSetDevice failed");
  CAMD_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (key->device_pending && key->device_stream == (hipStream_t)stream) key->device_pending = false;
  return report_status(key);
}

int camd_reserve(camd_server_key* key, size_t count) {
  if (!key) return fail(CAMD_EINVAL, "null key");
  std::lock_guard<std::mutex> lock(key->mu);
  DeviceGuard g(key->ctx->device);
  return reserve_locked(key, count);
}
int camd_ks_pbs_batch_device(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids, void* ct_out, size_t count,
                             void* stream) {
  return dispatch_device(key, Op::KsPbs, ct_in, lut_ids, ct_out, count, stream);
}

int camd_ks_pbs_batch_device_noalloc(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids, void* ct_out,
                                     size_t count, void* stream) {
  if (!key || (!ct_in && count) || (!ct_out && count)) return fail(CAMD_EINVAL, "null argument");
  if (key->variant != CAMD_PBS_WAVE || !wave_scratch_is_fixed(key->p))
    return fail(CAMD_EINVAL, "the no-allocation entry point serves keys of the wave kernel family with N <= 1024 only (the "
                             "other kernels need per-ciphertext scratch that grows with the batch)");
  const uint32_t small = key->p.lwe_dimension, big = (uint32_t)key->big();
  const size_t small1 = (size_t)small + 1, big1 = (size_t)big + 1;
  const size_t max_count = NOALLOC_MID_BYTES / (small1 * key->word);
  if (count > max_count)
    return fail(CAMD_EINVAL, "batch too large for the no-allocation entry point: its fixed scratch holds the keyswitch "
                             "output of at most " + std::to_string(max_count) + " ciphertexts of this parameter set, " +
                             std::to_string(count) + " were given");
  std::lock_guard<std::mutex> lock(key->mu);
  int rc = report_status(key);
  if (rc != CAMD_OK) return rc;
  if (count == 0) return CAMD_OK;
  DeviceGuard g(key->ctx->device);
  if (!g.ok) return fail(CAMD_EHIP, "hipSetDevice failed");
  unsigned char* base = key->ctx->noalloc_scratch;   // this GPU's copy of the array
  if (!base) return fail(CAMD_EHIP, "the fixed scratch of this GPU was not resolved at context creation");
  hipStream_t s = (hipStream_t)stream;
  rc = order_after_device_calls(key, s);
  if (rc != CAMD_OK) return rc;
  // The array keeps the previous call's contents. The counters are cleared by the launch itself
  // (hipMemsetAsync on `s`, inside wave_launch_pbs, whenever the kernel will use them); the keyswitch
  // writes every word of its output rows before the bootstrap reads them.
  void* counters = base;
  void* mid = base + NOALLOC_COUNTER_BYTES;
  if (key->p.torus_bits == 64) {
    rc = run_ks<uint64_t>(key, LweInput<uint64_t>::plain((const uint64_t*)ct_in, big1, big), (uint64_t*)mid, small1, count, s, false);
    if (rc == CAMD_OK)
      rc = run_pbs<uint64_t>(key, LweInput<uint64_t>::plain((const uint64_t*)mid, small1, small), lut_ids, (uint64_t*)ct_out, big1,
                             count, s, counters);
  } else {
    rc = run_ks<uint32_t>(key, LweInput<uint32_t>::plain((const uint32_t*)ct_in, big1, big), (uint32_t*)mid, small1, count, s, false);
    if (rc == CAMD_OK)
      rc = run_pbs<uint32_t>(key, LweInput<uint32_t>::plain((const uint32_t*)mid, small1, small), lut_ids, (uint32_t*)ct_out, big1,
                             count, s, counters);
  }
  CAMD_HIP(hipEventRecord(key->device_done, s));
  key->device_stream = s;
  key->device_pending = true;
  return rc;
}
int camd_pbs_ks_batch_device(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids, void* ct_out, size_t count,
                             void* stream) {
  return dispatch_device(key, Op::PbsKs, ct_in, lut_ids, ct_out, count, stream);
}
int camd_pbs_batch_device(camd_server_key* key, const void* ct_in, const uint32_t* lut_ids, void* ct_out, size_t count,
                          void* stream) {
  return dispatch_device(key, Op::Pbs, ct_in, lut_ids, ct_out, count, stream);
}
int camd_ks_batch_device(camd_server_key* key, const void* ct_in, void* ct_out, size_t count, void* stream) {
  return dispatch_device(key, Op::Ks, ct_in, nullptr, ct_out, count, stream);
}

int camd_axpby_ks_pbs_batch_device(camd_server_key* key, const void* x, size_t x_stride, uint64_t alpha, const void* y,
                                   size_t y_stride, uint64_t beta, uint64_t body_add, const uint32_t* lut_ids, void* ct_out,
                                   size_t count, void* stream) {
  const Combination c{x_stride, y, y_stride, alpha, beta, body_add};
  return dispatch_device(key, Op::KsPbs, x, lut_ids, ct_out, count, stream, &c);
}
int camd_axpby_pbs_ks_batch_device(camd_server_key* key, const void* x, size_t x_stride, uint64_t alpha, const void* y,
                                   size_t y_stride, uint64_t beta, uint64_t body_add, const uint32_t* lut_ids, void* ct_out,
                                   size_t count, void* stream) {
  const Combination c{x_stride, y, y_stride, alpha, beta, body_add};
  return dispatch_device(key, Op::PbsKs, x, lut_ids, ct_out, count, stream, &c);
}

int camd_lwe_axpby_device(camd_context* ctx, uint32_t torus_bits, void* out, size_t out_stride, const void* x,
                          size_t x_stride, uint64_t alpha, const void* y, size_t y_stride, uint64_t beta, uint64_t body_add,
                          size_t words, size_t count, void* stream) {
  if (!ctx || (count && words && (!out || !x))) return fail(CAMD_EINVAL, "null argument");
  if (torus_bits != 32 && torus_bits != 64) return fail(CAMD_EINVAL, "torus_bits must be 32 or 64");
  if (count && words && (out_stride < words || x_stride < words || (y && y_stride < words)))
    return fail(CAMD_EINVAL, "row stride shorter than the ciphertext");
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(CAMD_EHIP, "hipSetDevice failed");
  hipStream_t s = (hipStream_t)stream;
  if (torus_bits == 64)
    return launch_lwe_axpby<uint64_t>((uint64_t*)out, out_stride, (const uint64_t*)x, x_stride, alpha, (const uint64_t*)y,
                                      y_stride, beta, body_add, words, count, s);
  return launch_lwe_axpby<uint32_t>((uint32_t*)out, out_stride, (const uint32_t*)x, x_stride, (uint32_t)alpha,
                                    (const uint32_t*)y, y_stride, (uint32_t)beta, (uint32_t)body_add, words, count, s);
}

int camd_measure_fp64_fma_rate(camd_context* ctx, double* tflops) {
  if (!ctx || !tflops) return fail(CAMD_EINVAL, "null argument");
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(CAMD_EHIP, "hipSetDevice failed");
  return measure_fp64_fma_rate(ctx->stream, tflops);
}

int camd_timing_enable(camd_server_key* key, int enable) {
  if (!key) return fail(CAMD_EINVAL, "null key");
  DeviceGuard g(key->ctx->device);
  key->timing.clear();
  key->timing.enabled = enable != 0;
  return CAMD_OK;
}

int camd_timing_read(camd_server_key* key, double* pbs_ms_avg, uint64_t* pbs_launches, double* ks_ms_avg,
                     uint64_t* ks_launches) {
  if (!key) return fail(CAMD_EINVAL, "null key");
  DeviceGuard g(key->ctx->device);
  auto avg = [](std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, double* out_ms, uint64_t* out_n) -> int {
    double total = 0;
    uint64_t n = 0;
    for (auto& e : v) {
      CAMD_HIP(hipEventSynchronize(e.second));
      float ms = 0;
      CAMD_HIP(hipEventElapsedTime(&ms, e.first, e.second));
      total += ms;
      n++;
    }
    if (out_ms) *out_ms = n ? total / (double)n : 0.0;
    if (out_n) *out_n = n;
    return CAMD_OK;
  };
  int rc = avg(key->timing.pbs, pbs_ms_avg, pbs_launches);
  if (rc == CAMD_OK) rc = avg(key->timing.ks, ks_ms_avg, ks_launches);
  key->timing.clear();
  return rc;
}

}  // extern "C"
