// Issue-rate microbenchmark behind the abx_rope design notes: how many VALU instructions fit in the
// shadow of one v_mfma_f32_32x32x16_f16 (32 cycles of matrix pipe) for 1 or 2 waves per SIMD, and what dependent
// VALU chains cost.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o gpurun_in/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// NV VALU ops per MFMA gap, chains rotate over NR registers (NR = 1: fully dependent), KIND: 0 v_fma_f32, 1 + s_nop 0,
// 2: + one ds_read_b128 per 4 gaps, 3: v_fma + s_waitcnt lgkmcnt(0)
template <int NV, int NR, int KIND, bool MFMA>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ char smem[];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  h16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = threadIdx.x * 0.01f + e;
  float m1 = 1.0001f, m2 = 0.0001f;
  asm volatile("" : "+v"(m1), "+v"(m2));
  unsigned ldsaddr = (threadIdx.x & 63) * 16;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 ld = {0, 0, 0, 0};
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if (MFMA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int r = (g * NV + q) % NR;
        if (KIND == 4) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(v[r]) : "v"(m1), "v"(m2));
        else if (KIND == 5) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(v[r]) : "v"(m1), "v"(m2));
        else if (KIND == 6) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(v[r]) : "v"(m1), "v"(m2));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(m1), "v"(m2));
      }
      if (KIND == 1) asm volatile("s_nop 0");
      if (KIND == 2 && (g & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(ldsaddr));
      if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  for (int e = 0; e < 8; ++e) s += v[e];
  s += __builtin_bit_cast(float, ld[0]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// Power wall: MFMA-only stream whose operands change every instruction (4 A and 4 B fragments of pseudo-random
// fp16 in [-2, 2)) -- what a real GEMM feeds the matrix pipe -- versus the constant-operand stream above.
template <bool RANDOM>
__global__ __launch_bounds__(512) void kpow(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  h16x8 a[4], b[4];
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      st = st * 1664525u + 1013904223u;
      float fa = RANDOM ? ((int)(st >> 8 & 0xFFFF) - 32768) * (1.f / 16384.f) : 1.0f;
      st = st * 1664525u + 1013904223u;
      float fb = RANDOM ? ((int)(st >> 8 & 0xFFFF) - 32768) * (1.f / 16384.f) : 0.5f;
      a[i][e] = (_Float16)fa; b[i][e] = (_Float16)fb;
    }
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a[g & 3]), "v"(b[(g >> 2) & 3]));
    if (RANDOM && (it & 15) == 15)     // keep the accumulators finite and toggling
      for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] *= 0.001f;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <bool RANDOM>
void run_pow(const char* name) {
  const int nwg = 256, iters = 4000, threads = 512;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, nwg * 512 * sizeof(float));
  hipMalloc(&cyc, nwg * 8 * sizeof(unsigned long long));
  auto kern = kpow<RANDOM>;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg * 8);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += (double)v;
  const double ticks = sum / h.size();
  const double flops = 10.0 * iters * 16.0 * 32768.0 * nwg * 8;
  printf("%-40s %.0f TFLOP/s  (%.2f ticks/ns, %.1f ticks per MFMA per SIMD)\n", name, flops / (ms * 1e-3) * 1e-12,
         ticks / (ms * 1e6 / 10), ticks / iters / 16.0 / 2.0);
  hipFree(out); hipFree(cyc);
}

template <int NV, int NR, int KIND, bool MFMA>
void run(const char* name, int threads) {
  const int nwg = 256, iters = 2000;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, nwg * 512 * sizeof(float));
  hipMalloc(&cyc, nwg * 8 * sizeof(unsigned long long));
  hipMemset(cyc, 0, nwg * 8 * sizeof(unsigned long long));
  auto kern = k<NV, NR, KIND, MFMA>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), 100 * 1024, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), 100 * 1024, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg * 8);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double sum = 0; int cnt = 0;
  for (int i = 0; i < nwg; ++i) for (int w = 0; w < threads / 64; ++w) { sum += (double)h[i * 8 + w]; ++cnt; }
  const double per_gap = sum / cnt / iters / 16.0;
  const double ns_gap = ms * 1e6 / 10 / iters / 16.0;   // wall time per gap (kernel ~= loop)
  printf("%-40s w/SIMD %d NV %2d NR %d : %5.1f ticks/gap (%.2f/instr)  %5.2f ns/gap  -> %.2f ticks/ns\n", name, threads / 256, NV, NR, per_gap,
         per_gap / (NV + (MFMA ? 1 : 0) + (KIND == 1 || KIND == 3 ? 1 : 0)), ns_gap, per_gap / ns_gap);
  fflush(stdout);
  hipFree(out); hipFree(cyc);
}

#define RUN(NV, NR, KIND, MFMA, name) run<NV, NR, KIND, MFMA>(name, 256); run<NV, NR, KIND, MFMA>(name, 512);
int main() {
  run_pow<false>("mfma-only, constant operands (2 w/SIMD)");
  run_pow<true>("mfma-only, random operands (2 w/SIMD)");
  RUN(0, 1, 0, true, "mfma only");
  RUN(4, 8, 0, false, "valu only, independent");
  RUN(4, 1, 0, false, "valu only, fully dependent");
  RUN(4, 2, 0, false, "valu only, distance 2");
  RUN(8, 8, 4, false, "v_dot2c_f32_f16 only, independent");
  RUN(8, 8, 5, false, "v_dot2_f32_f16 only, independent");
  RUN(8, 8, 6, false, "v_perm_b32 only, independent");
  RUN(8, 8, 0, false, "v_fma_f32 only, 8 independent");
  RUN(2, 8, 0, true, "mfma + 2 indep valu");
  RUN(4, 8, 0, true, "mfma + 4 indep valu");
  RUN(5, 8, 0, true, "mfma + 5 indep valu");
  RUN(6, 8, 0, true, "mfma + 6 indep valu");
  RUN(7, 8, 0, true, "mfma + 7 indep valu");
  RUN(8, 8, 0, true, "mfma + 8 indep valu");
  RUN(10, 8, 0, true, "mfma + 10 indep valu");
  RUN(4, 1, 0, true, "mfma + 4 dependent valu");
  RUN(4, 2, 0, true, "mfma + 4 valu distance 2");
  RUN(8, 1, 0, true, "mfma + 8 dependent valu");
  RUN(8, 2, 0, true, "mfma + 8 valu distance 2");
  RUN(8, 4, 0, true, "mfma + 8 valu distance 4");
  RUN(4, 8, 1, true, "mfma + 4 indep valu + s_nop");
  RUN(4, 8, 2, true, "mfma + 4 indep valu + ds_read/4gaps");
  RUN(4, 8, 3, true, "mfma + 4 indep valu + s_waitcnt");
  return 0;
}
