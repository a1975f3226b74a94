// How fast do the CUs of an XCD get the SAME 128 KB (what the score kernels' prologue asks for: every workgroup of a latent
// group loads the group's B fragments, 32 workgroups per XCD at the same moment) compared with 128 KB of their own?
// 256 workgroups (one per CU: 160 KB of LDS each), WAVES waves each issuing its share as back-to-back 16-byte-per-lane loads,
// s_memtime at kernel start / after the last load is issued / after the last load has landed.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_l2_broadcast.hip -o gpurun_in/ubench_l2_broadcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int WAVES, int N>
__global__ __launch_bounds__(WAVES * 64) void k_bcast(const u32x4* __restrict__ buf, size_t region_vecs, int shared, int rot,
                                                     unsigned long long* stamps, unsigned* sink) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_readcyclecounter();
  const int g = blockIdx.x % 8, c = blockIdx.x / 8;
  const u32x4* base = buf + (shared ? (size_t)g : (size_t)blockIdx.x) * region_vecs + (size_t)w * N * 64 + lane;
  const int r0 = rot ? (c * 5) % N : 0;
  u32x4 r[N];
#pragma unroll
  for (int t = 0; t < N; ++t) r[t] = base[(size_t)((t + r0) % N) * 64];
  const unsigned long long t1 = __builtin_readcyclecounter();
  unsigned acc = 0;
#pragma unroll
  for (int t = 0; t < N; ++t) acc ^= r[t][0] ^ r[t][1] ^ r[t][2] ^ r[t][3];
  asm volatile("" : "+v"(acc));
  const unsigned long long t2 = __builtin_readcyclecounter();
  if (acc == 0x12345u) sink[0] = acc + smem[0];
  if (lane == 0) {
    stamps[(blockIdx.x * 8 + w) * 2] = t1 - t0;
    stamps[(blockIdx.x * 8 + w) * 2 + 1] = t2 - t0;
  }
}

__global__ __launch_bounds__(256) void k_flush(const u32x4* __restrict__ x, unsigned* sink, size_t n) {
  unsigned acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc ^= x[i][0];
  if (acc == 0x12345u) sink[0] = acc;
}

template <int WAVES, int N>
void run(const char* name, const u32x4* buf, const u32x4* big, size_t bign, unsigned long long* stamps, unsigned* sink, int shared, int rot, int flush) {
  const size_t region_vecs = (size_t)WAVES * N * 64;   // 128 KB
  hipFuncSetAttribute((const void*)k_bcast<WAVES, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  std::vector<unsigned long long> h(256 * 8 * 2);
  double si = 0, sd = 0, mx = 0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms_sum = 0;
  const int reps = 10;
  for (int it = 0; it < reps + 2; ++it) {
    if (flush) hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, 0, big, sink, bign);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_bcast<WAVES, N>), dim3(256), dim3(WAVES * 64), 160 * 1024 - 64, 0, buf, region_vecs, shared, rot, stamps, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    if (it < 2) continue;
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms_sum += ms;
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0, m = 0;
    for (int wg = 0; wg < 256; ++wg)
      for (int w = 0; w < WAVES; ++w) {
        a += h[(wg * 8 + w) * 2];
        b += h[(wg * 8 + w) * 2 + 1];
        m = std::max(m, (double)h[(wg * 8 + w) * 2 + 1]);
      }
    si += a / (256 * WAVES);
    sd += b / (256 * WAVES);
    mx += m;
  }
  printf("%-44s issued %7.0f  landed %7.0f  slowest wave %7.0f cycles   = %5.1f B/clk/CU   kernel %6.2f us\n", name, si / reps, sd / reps,
         mx / reps, 131072.0 / (sd / reps), ms_sum / reps * 1e3);
}

int main() {
  const size_t bytes = 256ull * 128 * 1024;   // a private 128 KB region per workgroup
  u32x4 *buf, *big;
  unsigned long long* stamps;
  unsigned* sink;
  const size_t bigbytes = 1ull << 30;
  hipMalloc(&buf, bytes);
  hipMalloc(&big, bigbytes);
  hipMalloc(&stamps, 256 * 8 * 2 * 8);
  hipMalloc(&sink, 64);
  hipMemset(buf, 1, bytes);
  hipMemset(big, 1, bigbytes);
  const size_t bign = bigbytes / 16;
  for (int flush = 0; flush < 2; ++flush) {
    printf("---- %s\n", flush ? "a 1 GB streaming read between launches (L2 and MALL turned over)" : "back to back (everything hot)");
    run<4, 32>("4 waves x 32 loads, shared by the XCD", buf, big, bign, stamps, sink, 1, 0, flush);
    run<4, 32>("4 waves x 32 loads, shared, staggered order", buf, big, bign, stamps, sink, 1, 1, flush);
    run<4, 32>("4 waves x 32 loads, private per workgroup", buf, big, bign, stamps, sink, 0, 0, flush);
    run<8, 16>("8 waves x 16 loads, shared by the XCD", buf, big, bign, stamps, sink, 1, 0, flush);
    run<8, 16>("8 waves x 16 loads, private per workgroup", buf, big, bign, stamps, sink, 0, 0, flush);
  }
  return 0;
}
