// What this box's HBM delivers to plain streaming kernels (context for the HBM-bound rows of DESIGN.md section 4):
// read-only reduction and float4 copy over 2 GiB, grid = 256 CUs x 8 workgroups of 256 threads.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm.hip -o gpurun_in/ubench_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ x, float* out, size_t n) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    f32x4 v = __builtin_nontemporal_load(x + i);
    acc += v[0] + v[1] + v[2] + v[3];
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ x, float4* __restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = x[i];
}

int main() {
  const size_t bytes = 2ull << 30, n = bytes / sizeof(float4);
  float4 *x, *y; float* out;
  hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&out, 4);
  hipMemset(x, 1, bytes); hipMemset(y, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int wgs = 4; wgs <= 16; wgs *= 2) {
      dim3 grid(256 * wgs), block(256);
      for (int r = 0; r < 2; ++r) { if (mode == 0) hipLaunchKernelGGL(k_read, grid, block, 0, 0, (const f32x4*)x, out, n); else hipLaunchKernelGGL(k_copy, grid, block, 0, 0, x, y, n); }
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) { if (mode == 0) hipLaunchKernelGGL(k_read, grid, block, 0, 0, (const f32x4*)x, out, n); else hipLaunchKernelGGL(k_copy, grid, block, 0, 0, x, y, n); }
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double gb = (mode == 0 ? 1.0 : 2.0) * bytes * 5 / (ms * 1e-3) * 1e-9;
      printf("%s, %2d workgroups per CU: %7.0f GB/s\n", mode == 0 ? "read 2 GiB (nt float4)" : "copy 2 GiB (read + write)", wgs, gb);
    }
  }
  return 0;
}
