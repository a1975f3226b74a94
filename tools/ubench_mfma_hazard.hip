// ubench_mfma_hazard: how many wait states does a VALU / v_permlane32_swap read of a fresh v_mfma_f32_32x32x16_f16 result
// need on MI355X when the matrix pipe of the SIMD is shared with a second wave?  (Background: profiles/
// r03_shared_b_permlane_hazard.txt -- the shared-B score kernel's drain read such a result 12 wait states after the MFMA,
// which is what hipcc pads, and got half-written accumulators in ~30 % of cold launches.)
//
// 512 threads = 2 waves per SIMD.  Every wave repeats: a phase jitter (s_sleep), a burst of BURST chained MFMAs on a
// private accumulator (the "main loop" of the neighbour), then ONE C = 0 MFMA whose result register 0 / 1 is read K wait
// states later by (0) v_mov_b32 or (1) v_permlane32_swap_b32, and again ~140 wait states later (the reference).  The
// kernel counts lanes whose early read differs from the late one, per 16-lane group.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_mfma_hazard.hip -o ubench_mfma_hazard && ./ubench_mfma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h16;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// PRIO: 0 = no wave priorities; 1 = waves 4-7 (the younger wave of every SIMD) at s_setprio 1 throughout; 2 = waves 4-7 raise
// the priority for their burst and drop it in front of the probed MFMA (the score kernels' former per-tile pattern: raised
// for the first half of a tile); 3 = waves 4-7 raise it in front of the probed MFMA and drop it after the early read.
template <int K, int CONSUMER, int BURST, int PRIO = 0>
__global__ __launch_bounds__(512) void hazard_kernel(unsigned* bad, int iters, unsigned seed) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  h16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (h16)(((lane & 31) < 4) ? 1.0f : 0.0f);                    // rows 0..3 of A are ones
    b[e] = (h16)(float)(((lane * 8 + e) * 5 + 3) % 7 - 3);            // column n = lane & 31, k = 8 (lane >> 5) + e
  }
  f32x16 acc2;
  for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
  unsigned nbad = 0, rng = seed * 2654435761u + blockIdx.x * 977u + w * 131u;
  float junk = (float)lane;
  if (PRIO == 1 && w >= 4) __builtin_amdgcn_s_setprio(1);
  for (int it = 0; it < iters; ++it) {
    rng = rng * 1664525u + 1013904223u;
    // phase jitter between the two waves of a SIMD (wave-uniform)
    const int j = __builtin_amdgcn_readfirstlane((rng >> 24) & 7);
    for (int q = 0; q < j; ++q) __builtin_amdgcn_s_sleep(1);
    // the neighbour's main loop: a burst of dependent MFMAs on another accumulator
    if (PRIO == 2 && w >= 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < BURST; ++i) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
    if (PRIO == 2 && w >= 4) __builtin_amdgcn_s_setprio(0);
    if (PRIO == 3 && w >= 4) __builtin_amdgcn_s_setprio(1);
    // operands written by VALU just before, like the kernel's v_cvt_pk
    h16x8 bb = b;
    bb[7] = (h16)((float)b[7] + (float)(it & 1));
    asm volatile("" : "+v"(bb));
    float early0, early1 = 0.f, late0, d5;
    // the result tuple is pinned to v[64:79] (clobbered) so that single registers of it can be named
    if (CONSUMER == 0) {
      asm volatile(
          "v_mfma_f32_32x32x16_f16 v[64:79], %3, %4, 0\n\t"
          "s_nop %5\n\t"
          "v_mov_b32 %0, v64\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "v_mov_b32 %1, v64\n\t"
          "v_mov_b32 %2, v69"
          : "=&v"(early0), "=&v"(late0), "=&v"(d5)
          : "v"(a), "v"(bb), "n"(K - 1)
          : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
    } else {
      // swap(v64, v65) as the kernel does: v65's lower half (row 1) -> v64's upper half; afterwards every lane of v64
      // holds the column sum of its column (rows 0 and 1 of D are equal)
      float t0 = junk;
      asm volatile(
          "v_mfma_f32_32x32x16_f16 v[64:79], %4, %5, 0\n\t"
          "s_nop %6\n\t"
          "v_permlane32_swap_b32 v64, v65\n\t"
          "s_nop 1\n\t"
          "v_mov_b32 %1, v64\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "v_mov_b32 %2, v65\n\t"
          "v_mov_b32 %3, v69"
          : "+v"(t0), "=&v"(early0), "=&v"(late0), "=&v"(d5)
          : "v"(a), "v"(bb), "n"(K - 1)
          : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
      early1 = t0; (void)early1;
    }
    if (PRIO == 3 && w >= 4) __builtin_amdgcn_s_setprio(0);
    // expected value of D[row 0][column n] = sum over k of bb(k, n): lanes (n, hi) hold k = 8 hi .. 8 hi + 7
    float own = 0.f;
    for (int e = 0; e < 8; ++e) own += (float)bb[e];
    const float other = __shfl_xor(own, 32);
    const float expect = own + other;                                   // same for both halves
    // rows 0..3 of D live in lanes 0-31 (registers 0..3); row 0 = register 0.  Lanes 32-63 of register 0 hold row 4 = 0.
    bool wrong;
    if (CONSUMER == 0) wrong = (early0 != late0) || (late0 != (lane < 32 ? expect : 0.f));
    else wrong = early0 != expect;
    nbad += wrong ? 1u : 0u;
    junk += d5 * 1e-30f + late0 * 1e-30f;
  }
  if (acc2[3] == 12345.678f) nbad += 1000000;                          // keep the burst alive
  if (junk == -1.f) nbad += 1000000;
  atomicAdd(&bad[(w >= 4 ? 4 : 0) + (lane >> 4)], nbad);
}

template <int K, int CONSUMER, int BURST, int PRIO = 0>
static void run(unsigned* dbad, const char* name) {
  CHECK(hipMemset(dbad, 0, 8 * sizeof(unsigned)));
  const int iters = 2000;
  hipLaunchKernelGGL((hazard_kernel<K, CONSUMER, BURST, PRIO>), dim3(256), dim3(512), 0, 0, dbad, iters, 12345u);
  CHECK(hipDeviceSynchronize());
  unsigned h[8];
  CHECK(hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost));
  const double tot = 256.0 * 4 * 16 * iters;    // lane-reads per (wave half, 16-lane group)
  printf("%-10s K=%2d burst=%d prio=%d : wrong early reads per 16-lane group  old waves [%u %u %u %u]  young waves [%u %u %u %u]  (of %.0f each)\n",
         name, K, BURST, PRIO, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], tot);
}

int main(int argc, char** argv) {
  unsigned* dbad;
  CHECK(hipMalloc(&dbad, 8 * sizeof(unsigned)));
#define ROW(K) run<K, 0, 0>(dbad, "v_mov"); run<K, 0, 8>(dbad, "v_mov"); run<K, 1, 0>(dbad, "permswap"); run<K, 1, 8>(dbad, "permswap");
  if (argc > 1 && argv[1][0] == 'p') {
    // the same probes with wave priorities in play: first launch of the process included (cold)
#define PROW(K, P) run<K, 0, 8, P>(dbad, "v_mov"); run<K, 1, 8, P>(dbad, "permswap"); run<K, 1, 0, P>(dbad, "permswap");
#define PALL(K) PROW(K, 1) PROW(K, 2) PROW(K, 3)
    PALL(12) PALL(8) PALL(6) PALL(5) PALL(4)
    return 0;
  }
  ROW(2) ROW(4) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(16) ROW(20)
  return 0;
}
