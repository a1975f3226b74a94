// Fast Walsh-Hadamard transform over the last dimension: y = (x . H_n) * scale, Sylvester (natural)
// order, n a power of two.  Drop-in for the reference's only native dependency, the un-vendored CUDA
// extension `fast_hadamard_transform.hadamard_transform(x, scale)` (call sites
// palu/model/modules/hadamard_utils.py:141,145,177) and equal to the in-tree butterfly `matmul_hadU`
// (:92-113).  Weight preparation is offline (the rotation is fused into VT / U / W_o),
// so this kernel is simple: one workgroup per row, the row lives in LDS as fp32, log2(n) butterfly
// stages; the first 3 stages of each 8-element run are done in registers.
#include "palu_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void fwht_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int n,
                                                   float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s = reinterpret_cast<float*>(smem_raw);
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  const T* xr = x + row * n;
  T* yr = y + row * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = (float)xr[i];
  __syncthreads();
  for (int h = 1; h < n; h <<= 1) {
    for (int idx = threadIdx.x; idx < (n >> 1); idx += blockDim.x) {
      const int i = ((idx / h) * (h << 1)) + (idx % h);
      const float a = s[i], b = s[i + h];
      s[i] = a + b;
      s[i + h] = a - b;
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) yr[i] = (T)(s[i] * scale);
}

}  // namespace

extern "C" int palu_hadamard_transform(const void* x, void* y, int64_t rows, int n, float scale, int dtype,
                                       palu_stream_t stream) {
  PALU_REQUIRE(x && y && rows >= 0 && n > 0, PALU_ERR_ARG, "hadamard_transform: bad arguments");
  PALU_REQUIRE((n & (n - 1)) == 0 && n <= 16384, PALU_ERR_UNSUPPORTED,
               "hadamard_transform: n must be a power of two <= 16384 (got %d)", n);
  PALU_REQUIRE(dtype == 0 || dtype == 1, PALU_ERR_ARG, "hadamard_transform: dtype 0 = fp16, 1 = fp32");
  if (rows == 0) return PALU_OK;
  PALU_REQUIRE(rows < (1ll << 31), PALU_ERR_UNSUPPORTED, "hadamard_transform: too many rows");
  {
    int rca = palu_func_max_lds(reinterpret_cast<const void*>(fwht_kernel<float>), 65536);
    if (!rca) rca = palu_func_max_lds(reinterpret_cast<const void*>(fwht_kernel<h16>), 65536);
    if (rca) return rca;
  }
  dim3 grid((unsigned)rows), block(n >= 512 ? 256 : (n >= 128 ? 64 : 64));
  if (dtype == 0)
    hipLaunchKernelGGL(fwht_kernel<h16>, grid, block, (size_t)n * 4, (hipStream_t)stream, (const h16*)x, (h16*)y, rows, n, scale);
  else
    hipLaunchKernelGGL(fwht_kernel<float>, grid, block, (size_t)n * 4, (hipStream_t)stream, (const float*)x, (float*)y, rows, n, scale);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}
