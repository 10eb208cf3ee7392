// Leveled (bootstrap-free) LWE operations on device-resident ciphertexts:
//   out[c][j] = alpha * x[c][j] + beta * y[c][j]   (wrapping), plus body_add on the last word.
//
// One entry covers what the reference does with concrete-core's fuse_add_lwe_ciphertext,
// fuse_opp_lwe_ciphertext, fuse_mul_lwe_ciphertext_cleartext and
// fuse_add_lwe_ciphertext_plaintext around the bootstrap
// (concrete-shortint/src/engine/server_side/add.rs:17-26, neg.rs:52-68, scalar_mul.rs:19-30,
// scalar_add.rs:19-32, mod.rs:182-188; concrete-boolean/src/engine/mod.rs:347-353), so that a
// chain of operations never leaves the GPU. Rows are addressed with a stride: a block of a
// radix integer batch is a strided view.
#include "kernels.hpp"

namespace camd {

namespace {

template <typename T>
__global__ __launch_bounds__(256) void lwe_axpby_kernel(T* __restrict__ out, size_t out_stride, const T* x, size_t x_stride,
                                                        T alpha, const T* y, size_t y_stride, T beta, T body_add,
                                                        uint32_t words) {
  const size_t c = blockIdx.y;
  T* o = out + c * out_stride;
  const T* xs = x + c * x_stride;
  const T* ys = y ? y + c * y_stride : nullptr;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < words; j += gridDim.x * blockDim.x) {
    T v = alpha * xs[j];
    if (ys) v += beta * ys[j];
    if (j == words - 1) v += body_add;
    o[j] = v;
  }
}

}  // namespace

template <typename T>
int launch_lwe_axpby(T* out, size_t out_stride, const T* x, size_t x_stride, T alpha, const T* y, size_t y_stride, T beta,
                     T body_add, size_t words, size_t count, hipStream_t stream) {
  if (count == 0 || words == 0) return CAMD_OK;
  if (words > 0xFFFFFFFFull) return fail(CAMD_EINVAL, "lwe_axpby: ciphertext too long");
  const unsigned bx = (unsigned)std::min<size_t>((words + 255) / 256, 64);
  for (size_t done = 0; done < count;) {  // gridDim.y is limited to 65535
    const size_t chunk = std::min<size_t>(count - done, 65535);
    hipLaunchKernelGGL(lwe_axpby_kernel<T>, dim3(bx, (unsigned)chunk), dim3(256), 0, stream, out + done * out_stride,
                       out_stride, x + done * x_stride, x_stride, alpha, y ? y + done * y_stride : nullptr, y_stride, beta,
                       body_add, (uint32_t)words);
    done += chunk;
  }
  CAMD_HIP(hipGetLastError());
  return CAMD_OK;
}

template int launch_lwe_axpby<uint64_t>(uint64_t*, size_t, const uint64_t*, size_t, uint64_t, const uint64_t*, size_t,
                                        uint64_t, uint64_t, size_t, size_t, hipStream_t);
template int launch_lwe_axpby<uint32_t>(uint32_t*, size_t, const uint32_t*, size_t, uint32_t, const uint32_t*, size_t,
                                        uint32_t, uint32_t, size_t, size_t, hipStream_t);

}  // namespace camd
