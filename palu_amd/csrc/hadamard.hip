// Fast Walsh-Hadamard transform over the last dimension: y = (x . H_n) * scale, Sylvester (natural)
// order, n a power of two.  Drop-in for the reference's only native dependency, the un-vendored CUDA
// extension `fast_hadamard_transform.hadamard_transform(x, scale)` (call sites
// palu/model/modules/hadamard_utils.py:141,145,177) and equal to the in-tree butterfly `matmul_hadU`
// (:92-113).  Weight preparation is offline (the rotation is fused into VT / U / W_o),
// so the kernels are small.  n <= 2048: fwht_wave_kernel -- a row (or 64 / n rows) per WAVE, every lane holds n / 64
// consecutive elements in fp32 registers: the stages below the lane's run are in-register butterflies, the six above it
// cross lanes (DPP / ds_bpermute through __shfl_xor, no LDS storage, no barrier); loads and stores are the lane's
// contiguous 2..64 bytes.  Stage order h = 1, 2, 4, ... and the fp32 (a + b, a - b) arithmetic are those of the LDS kernel
// and of matmul_hadU: bit-identical results.  Larger n: one workgroup per row, the row in LDS as fp32.
#include "palu_common.h"

namespace {

// E = elements per lane (n = 64 E for n >= 64; n < 64: E = 1 and 64 / n rows share the wave)
template <typename T, int E>
__global__ __launch_bounds__(256) void fwht_wave_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int n,
                                                        float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int rpw = n >= 64 ? 1 : 64 / n;                    // rows per wave
  const int64_t row = wave * rpw + (n >= 64 ? 0 : lane / n);
  const int col = n >= 64 ? lane * E : lane % n;
  const bool ok = row < rows;
  float v[E];
  {
    const T* src = x + (ok ? row : 0) * n + col;
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = ok ? (float)src[e] : 0.f;
  }
  // stages inside the lane's run
#pragma unroll
  for (int h = 1; h < E; h <<= 1) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      if ((i & h) == 0) {
        const float a = v[i], b = v[i + h];
        v[i] = a + b;
        v[i + h] = a - b;
      }
    }
  }
  // stages across lanes: element distance h = E d, lane distance d; the lane with bit d set holds the upper element
  const int nd = n >= 64 ? 64 : n;
  for (int d = 1; d < nd; d <<= 1) {
    const bool upper = (lane & d) != 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float o = __shfl_xor(v[e], d, 64);
      v[e] = upper ? o - v[e] : v[e] + o;
    }
  }
  if (ok) {
    T* dst = y + row * n + col;
#pragma unroll
    for (int e = 0; e < E; ++e) dst[e] = (T)(v[e] * scale);
  }
}

template <typename T, int E>
int launch_fwht_wave(const void* x, void* y, int64_t rows, int n, float scale, hipStream_t s) {
  const int rpw = n >= 64 ? 1 : 64 / n;
  const int64_t waves = (rows + rpw - 1) / rpw;
  hipLaunchKernelGGL((fwht_wave_kernel<T, E>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, (const T*)x, (T*)y, rows, n, scale);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

template <typename T>
int launch_fwht_wave_n(const void* x, void* y, int64_t rows, int n, float scale, hipStream_t s) {
  switch (n >= 64 ? n / 64 : 1) {
    case 1: return launch_fwht_wave<T, 1>(x, y, rows, n, scale, s);
    case 2: return launch_fwht_wave<T, 2>(x, y, rows, n, scale, s);
    case 4: return launch_fwht_wave<T, 4>(x, y, rows, n, scale, s);
    case 8: return launch_fwht_wave<T, 8>(x, y, rows, n, scale, s);
    case 16: return launch_fwht_wave<T, 16>(x, y, rows, n, scale, s);
    default: return launch_fwht_wave<T, 32>(x, y, rows, n, scale, s);
  }
}

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
  if (n <= 2048)
    return dtype == 0 ? launch_fwht_wave_n<h16>(x, y, rows, n, scale, (hipStream_t)stream)
                      : launch_fwht_wave_n<float>(x, y, rows, n, scale, (hipStream_t)stream);
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
