// Host-side client operations declared in include/concrete_amd_client.h: secret keys,
// standard-domain server-key material, LWE encryption / phase, look-up tables.
// No GPU code. The arithmetic mirrors what the reference obtains from concrete-core's
// DefaultEngine at concrete-shortint/src/engine/client_side.rs:9-144 and
// concrete-shortint/src/engine/server_side/mod.rs:31-72.
#pragma clang fp contract(off)
#include <sys/random.h>

#include <cerrno>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/concrete_amd_client.h"

namespace camd {
int fail(int code, const std::string& msg);
}

namespace {

inline uint64_t splitmix64(uint64_t& x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

// ChaCha20 block function (RFC 8439 quarter rounds; 64-bit block counter, zero nonce): the
// secure mode of camd_rng. key = r.s, block number = r.counter, output = r.block.
inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
inline void chacha_quarter(uint32_t* x, int a, int b, int c, int d) {
  x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);
  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);
  x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);
  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
}
void chacha_refill(camd_rng& r) {
  uint32_t in[16], x[16];
  in[0] = 0x61707865u; in[1] = 0x3320646eu; in[2] = 0x79622d32u; in[3] = 0x6b206574u;
  std::memcpy(in + 4, r.s, 32);
  in[12] = (uint32_t)r.counter;
  in[13] = (uint32_t)(r.counter >> 32);
  in[14] = in[15] = 0;
  std::memcpy(x, in, sizeof(x));
  for (int round = 0; round < 10; round++) {
    chacha_quarter(x, 0, 4, 8, 12); chacha_quarter(x, 1, 5, 9, 13);
    chacha_quarter(x, 2, 6, 10, 14); chacha_quarter(x, 3, 7, 11, 15);
    chacha_quarter(x, 0, 5, 10, 15); chacha_quarter(x, 1, 6, 11, 12);
    chacha_quarter(x, 2, 7, 8, 13); chacha_quarter(x, 3, 4, 9, 14);
  }
  for (int i = 0; i < 16; i++) x[i] += in[i];
  std::memcpy(r.block, x, 64);
  r.counter++;
  r.have = 8;
}

inline uint64_t next(camd_rng& r) {
  if (r.secure) {
    if (r.have == 0) chacha_refill(r);
    return r.block[8 - r.have--];
  }
  uint64_t* s = r.s;
  const uint64_t result = rotl(s[1] * 5, 7) * 9;
  const uint64_t t = s[1] << 17;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl(s[3], 45);
  return result;
}
// Box-Muller, one deviate per call
inline double gauss(camd_rng& r) {
  const double u1 = ((double)(next(r) >> 11) + 1.0) * 0x1.0p-53;
  const double u2 = (double)(next(r) >> 11) * 0x1.0p-53;
  return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925286766559 * u2);
}
inline double round_half_away(double x) {
  const double t = std::trunc(x);
  const double d = x - t;
  return (std::fabs(d) >= 0.5) ? t + std::copysign(1.0, x) : t;
}

template <typename T>
struct W;
template <>
struct W<uint64_t> {
  using S = int64_t;
  static constexpr int BITS = 64;
  static constexpr double SCALE = 0x1.0p64;
};
template <>
struct W<uint32_t> {
  using S = int32_t;
  static constexpr int BITS = 32;
  static constexpr double SCALE = 0x1.0p32;
};

template <typename T>
inline T from_torus(double x) {
  using S = typename W<T>::S;
  double fr = x - round_half_away(x);
  fr *= W<T>::SCALE;
  fr = round_half_away(fr);
  S s;
  if (fr >= W<T>::SCALE * 0.5) s = (S)((T(1) << (W<T>::BITS - 1)) - 1);
  else if (fr <= -W<T>::SCALE * 0.5) s = (S)(T(1) << (W<T>::BITS - 1));
  else s = (S)fr;
  return (T)s;
}
template <typename T>
inline T random_torus(camd_rng& r) { return (T)(next(r) >> (64 - W<T>::BITS)); }
template <typename T>
inline T random_noise(camd_rng& r, double std_dev) { return from_torus<T>(std_dev * gauss(r)); }

template <typename T>
void lwe_encrypt_one(camd_rng& rng, const T* key, size_t dim, T pt, double std_dev, T* ct) {
  T acc = 0;
  for (size_t i = 0; i < dim; i++) {
    ct[i] = random_torus<T>(rng);
    acc += ct[i] * key[i];
  }
  ct[dim] = acc + random_noise<T>(rng, std_dev) + pt;
}

// A generator of the parent's kind for one worker: the deterministic kind is re-seeded from one
// output word (what the tests' reference sequence does), the secure kind gets a fresh 256-bit
// ChaCha20 key drawn from the parent's stream.
camd_rng derive(camd_rng& parent) {
  camd_rng sub;
  if (parent.secure) {
    std::memset(&sub, 0, sizeof(sub));
    for (int i = 0; i < 4; i++) sub.s[i] = next(parent);
    sub.secure = 1;
  } else {
    camd_rng_seed(&sub, next(parent));
  }
  return sub;
}

template <typename T>
void gen_ggsw(camd_rng sub, const camd_params& p, double std_dev, T bit, const T* glwe_key, T* out) {
  const size_t k = p.glwe_dimension, N = p.polynomial_size;
  for (uint32_t l = 1; l <= p.pbs_level; l++)
    for (size_t row = 0; row <= k; row++) {
      T* body = out + k * N;
      for (size_t j = 0; j < k * N; j++) out[j] = random_torus<T>(sub);
      for (size_t j = 0; j < N; j++) body[j] = random_noise<T>(sub, std_dev);
      for (size_t m = 0; m < k; m++) {
        const T* mask = out + m * N;
        const T* key = glwe_key + m * N;
        for (size_t d = 0; d < N; d++) {
          if (!key[d]) continue;
          for (size_t t = 0; t < N - d; t++) body[t + d] += mask[t];
          for (size_t t = N - d; t < N; t++) body[t + d - N] -= mask[t];
        }
      }
      out[row * N] += bit * (T(1) << (W<T>::BITS - p.pbs_base_log * l));
      out += (k + 1) * N;
    }
}

template <typename T>
void gen_bsk(camd_rng& rng, const camd_params& p, double std_dev, const T* lwe_key, const T* glwe_key, T* bsk,
             int threads) {
  const size_t n = p.lwe_dimension;
  const size_t per = (size_t)p.pbs_level * (p.glwe_dimension + 1) * (p.glwe_dimension + 1) * p.polynomial_size;
  std::vector<camd_rng> seeds(n);
  for (size_t i = 0; i < n; i++) seeds[i] = derive(rng);
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = (int)(n ? n : 1);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++)
    pool.emplace_back([&, t]() {
      for (size_t i = n * t / threads; i < n * (t + 1) / threads; i++)
        gen_ggsw<T>(seeds[i], p, std_dev, lwe_key[i], glwe_key, bsk + i * per);
    });
  for (auto& th : pool) th.join();
}

template <typename T>
void gen_ksk(camd_rng& rng, const T* in_key, size_t in_dim, const T* out_key, size_t out_dim, uint32_t base_log,
             uint32_t level, double std_dev, T* ksk) {
  for (size_t i = 0; i < in_dim; i++)
    for (uint32_t l = 1; l <= level; l++) {
      const T pt = in_key[i] * (T(1) << (W<T>::BITS - base_log * l));
      lwe_encrypt_one<T>(rng, out_key, out_dim, pt, std_dev, ksk + (i * level + (l - 1)) * (out_dim + 1));
    }
}

bool bad_bits(uint32_t b) { return b != 32 && b != 64; }

}  // namespace

extern "C" {

void camd_rng_seed(camd_rng* rng, uint64_t seed) {
  std::memset(rng, 0, sizeof(*rng));
  uint64_t x = seed;
  for (int i = 0; i < 4; i++) rng->s[i] = splitmix64(x);
}

int camd_rng_seed_secure(camd_rng* rng) {
  if (!rng) return camd::fail(CAMD_EINVAL, "rng_seed_secure: null generator");
  std::memset(rng, 0, sizeof(*rng));
  unsigned char* dst = (unsigned char*)rng->s;
  size_t got = 0;
  while (got < sizeof(rng->s)) {
    const ssize_t r = getrandom(dst + got, sizeof(rng->s) - got, 0);
    if (r < 0) {
      if (errno == EINTR) continue;
      return camd::fail(CAMD_EINVAL, std::string("rng_seed_secure: getrandom failed: ") + std::strerror(errno));
    }
    got += (size_t)r;
  }
  rng->secure = 1;
  return CAMD_OK;
}

int camd_client_gen_binary_key(camd_rng* rng, uint32_t torus_bits, size_t len, void* key) {
  if (!rng || !key || bad_bits(torus_bits)) return camd::fail(CAMD_EINVAL, "gen_binary_key: bad argument");
  for (size_t i = 0; i < len; i++) {
    const uint64_t bit = next(*rng) >> 63;
    if (torus_bits == 64) ((uint64_t*)key)[i] = bit;
    else ((uint32_t*)key)[i] = (uint32_t)bit;
  }
  return CAMD_OK;
}

int camd_client_gen_bootstrap_key(camd_rng* rng, const camd_params* p, double glwe_std_dev, const void* lwe_key,
                                  const void* glwe_key, void* bsk_out, int threads) {
  if (!rng || !p || !lwe_key || !glwe_key || !bsk_out || bad_bits(p->torus_bits))
    return camd::fail(CAMD_EINVAL, "gen_bootstrap_key: bad argument");
  if (p->pbs_base_log * p->pbs_level >= p->torus_bits) return camd::fail(CAMD_EINVAL, "gen_bootstrap_key: bad decomposition");
  if (p->torus_bits == 64)
    gen_bsk<uint64_t>(*rng, *p, glwe_std_dev, (const uint64_t*)lwe_key, (const uint64_t*)glwe_key, (uint64_t*)bsk_out, threads);
  else
    gen_bsk<uint32_t>(*rng, *p, glwe_std_dev, (const uint32_t*)lwe_key, (const uint32_t*)glwe_key, (uint32_t*)bsk_out, threads);
  return CAMD_OK;
}

int camd_client_gen_keyswitch_key(camd_rng* rng, uint32_t torus_bits, const void* in_key, size_t in_dim,
                                  const void* out_key, size_t out_dim, uint32_t base_log, uint32_t level,
                                  double lwe_std_dev, void* ksk_out) {
  if (!rng || !in_key || !out_key || !ksk_out || bad_bits(torus_bits) || base_log * level >= torus_bits)
    return camd::fail(CAMD_EINVAL, "gen_keyswitch_key: bad argument");
  if (torus_bits == 64)
    gen_ksk<uint64_t>(*rng, (const uint64_t*)in_key, in_dim, (const uint64_t*)out_key, out_dim, base_log, level, lwe_std_dev,
                      (uint64_t*)ksk_out);
  else
    gen_ksk<uint32_t>(*rng, (const uint32_t*)in_key, in_dim, (const uint32_t*)out_key, out_dim, base_log, level, lwe_std_dev,
                      (uint32_t*)ksk_out);
  return CAMD_OK;
}

int camd_client_lwe_encrypt(camd_rng* rng, uint32_t torus_bits, const void* key, size_t dim, const void* plaintexts,
                            size_t count, double std_dev, void* ct_out) {
  if (!rng || !key || (!plaintexts && count) || (!ct_out && count) || bad_bits(torus_bits))
    return camd::fail(CAMD_EINVAL, "lwe_encrypt: bad argument");
  for (size_t c = 0; c < count; c++) {
    if (torus_bits == 64)
      lwe_encrypt_one<uint64_t>(*rng, (const uint64_t*)key, dim, ((const uint64_t*)plaintexts)[c], std_dev,
                                (uint64_t*)ct_out + c * (dim + 1));
    else
      lwe_encrypt_one<uint32_t>(*rng, (const uint32_t*)key, dim, ((const uint32_t*)plaintexts)[c], std_dev,
                                (uint32_t*)ct_out + c * (dim + 1));
  }
  return CAMD_OK;
}

int camd_client_lwe_phase(uint32_t torus_bits, const void* key, size_t dim, const void* ct, size_t count,
                          void* phases_out) {
  if (!key || (!ct && count) || (!phases_out && count) || bad_bits(torus_bits))
    return camd::fail(CAMD_EINVAL, "lwe_phase: bad argument");
  for (size_t c = 0; c < count; c++) {
    if (torus_bits == 64) {
      const uint64_t* k = (const uint64_t*)key;
      const uint64_t* x = (const uint64_t*)ct + c * (dim + 1);
      uint64_t acc = 0;
      for (size_t i = 0; i < dim; i++) acc += x[i] * k[i];
      ((uint64_t*)phases_out)[c] = x[dim] - acc;
    } else {
      const uint32_t* k = (const uint32_t*)key;
      const uint32_t* x = (const uint32_t*)ct + c * (dim + 1);
      uint32_t acc = 0;
      for (size_t i = 0; i < dim; i++) acc += x[i] * k[i];
      ((uint32_t*)phases_out)[c] = x[dim] - acc;
    }
  }
  return CAMD_OK;
}

int camd_client_shortint_accumulator(uint32_t k, uint32_t N, uint32_t message_modulus, uint32_t carry_modulus,
                                     const uint64_t* f_values, uint64_t* glwe_out) {
  if (!f_values || !glwe_out || !message_modulus || !carry_modulus) return camd::fail(CAMD_EINVAL, "accumulator: bad argument");
  const size_t modulus_sup = (size_t)message_modulus * carry_modulus;
  if (modulus_sup > N || (N % modulus_sup)) return camd::fail(CAMD_EINVAL, "accumulator: message space does not divide N");
  const size_t box = N / modulus_sup, half = box / 2;
  const uint64_t delta = (1ULL << 63) / (uint64_t)modulus_sup;
  std::vector<uint64_t> tmp(N);
  for (size_t i = 0; i < modulus_sup; i++)
    for (size_t j = 0; j < box; j++) tmp[i * box + j] = f_values[i] * delta;
  for (size_t j = 0; j < half; j++) tmp[j] = (uint64_t)0 - tmp[j];
  std::memset(glwe_out, 0, (size_t)k * N * sizeof(uint64_t));
  for (size_t j = 0; j < N; j++) glwe_out[(size_t)k * N + j] = tmp[(j + half) % N];
  return CAMD_OK;
}

int camd_client_boolean_accumulator(uint32_t k, uint32_t N, uint32_t* glwe_out) {
  if (!glwe_out) return camd::fail(CAMD_EINVAL, "accumulator: bad argument");
  std::memset(glwe_out, 0, (size_t)k * N * sizeof(uint32_t));
  for (size_t j = 0; j < N; j++) glwe_out[(size_t)k * N + j] = 1u << 29;
  return CAMD_OK;
}

}  // extern "C"
