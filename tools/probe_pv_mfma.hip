// Hardware probe for the matrix-core P.V path (palu_amd/csrc/pv_mfma.h): checks, on a real gfx950,
//  (1) the lane mapping of ds_read_b64_tr_b16 that the design assumes,
//  (2) LDS-DMA (buffer_load_dwordx4 ... lds) + XOR swizzle + transpose read + v_mfma_f32_16x16x32_f16
//      end to end against a CPU sum, for NT = 1, 2, 4 col-tiles per wave, including ring-slot reuse and
//      the clamped tail (rows >= L hold NaN bit patterns and must never be multiplied in).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_pv_mfma.hip -o gpurun_in/probe_pv_mfma
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../palu_amd/csrc/pv_mfma.h"

void palu_set_error(const char*, ...) {}

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void tr_probe(const short* in, short* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* lds = reinterpret_cast<short*>(smem);
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = in[i];
  __syncthreads();
  // lane l supplies the address of 4 consecutive shorts: element index 4*l
  const unsigned addr = (unsigned)reinterpret_cast<uintptr_t>(smem) + threadIdx.x * 8;
  h16x4 v = pvm::tr_read(addr);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = __builtin_bit_cast(short, v[j]);
}

template <int NT>
__global__ void pv_probe(const h16* v, int ldv, int col0, const h16* pmat /*[4][rows_pad]*/, int rows_pad, int L,
                         float* out /*[4][16*NT]*/) {
  using C = pvm::Cfg<NT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned base = (unsigned)reinterpret_cast<uintptr_t>(smem);
  const int lane = threadIdx.x;
  const unsigned P_OFF = 3 * C::UB;   // after the 3-slot ring
  h16* pl = reinterpret_cast<h16*>(smem + P_OFF);
  for (int i = lane; i < 4 * rows_pad; i += 64) pl[i] = pmat[i];
  u32x4 rs;
  const unsigned long long vb = reinterpret_cast<unsigned long long>(v);
  rs[0] = __builtin_amdgcn_readfirstlane((unsigned)vb);
  rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(vb >> 32));
  rs[2] = __builtin_amdgcn_readfirstlane((unsigned)(((long long)(L - 1) * ldv + 16 * NT + col0) * 2));
  rs[3] = 0x00020000u;
  const unsigned row_bytes = (unsigned)ldv * 2;
  const pvm::Lane<NT> ln = pvm::make_lane<NT>(lane, col0, row_bytes, rows_pad * 2);
  f32x4 acc[NT];
  for (int ct = 0; ct < NT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nunit = (L + 31) / 32;
  for (int u = 0; u < 3 && u < nunit; ++u) pvm::dma_unit<NT>(ln, rs, base + u * C::UB, 32 * u, L, row_bytes, lane);
  for (int u = 0; u < nunit; ++u) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    pvm::pv_unit<NT>(acc, ln, base + (u % 3) * C::UB, base + P_OFF + 32 * u * 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (u + 3 < nunit) pvm::dma_unit<NT>(ln, rs, base + (u % 3) * C::UB, 32 * (u + 3), L, row_bytes, lane);
  }
  // D layout: lane l, reg j -> latent column 16*ct + 4*(l/16) + j, head slot l%16
  const int n = lane & 15, qd = lane >> 4;
  if (n < 4)
    for (int ct = 0; ct < NT; ++ct)
      for (int j = 0; j < 4; ++j) out[n * 16 * NT + 16 * ct + 4 * qd + j] = acc[ct][j];
  // duplicates (head slots 4..15) must equal head n&3: report mismatches through a flag row
  if (n >= 4)
    for (int ct = 0; ct < NT; ++ct)
      for (int j = 0; j < 4; ++j) out[4 * 16 * NT + (lane * NT + ct) * 4 + j] = acc[ct][j];
}

template <int NT>
int run_pv(int L, int col0) {
  const int ldv = 384, rows_alloc = ((L + 31) / 32) * 32 + 8, rows_pad = ((L + 31) / 32) * 32;
  std::vector<h16> v((size_t)rows_alloc * ldv), p((size_t)4 * rows_pad);
  srand(1234 + NT);
  for (int r = 0; r < rows_alloc; ++r)
    for (int c = 0; c < ldv; ++c) {
      float x = (float)((rand() % 2001) - 1000) / 500.f;
      h16 hv = (h16)x;
      if (r >= L) {
        unsigned short nanb = 0x7E00;   // NaN
        memcpy(&hv, &nanb, 2);
      }
      v[(size_t)r * ldv + c] = hv;
    }
  for (int h = 0; h < 4; ++h)
    for (int r = 0; r < rows_pad; ++r) p[(size_t)h * rows_pad + r] = r < L ? (h16)((float)(rand() % 1000) / 1000.f) : (h16)0.f;
  h16 *dv, *dp;
  float* dout;
  const size_t nout = 4 * 16 * NT + 64 * NT * 4;
  CK(hipMalloc(&dv, v.size() * 2));
  CK(hipMalloc(&dp, p.size() * 2));
  CK(hipMalloc(&dout, nout * 4));
  CK(hipMemcpy(dv, v.data(), v.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dp, p.data(), p.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dout, 0, nout * 4));
  const size_t lds = 3 * pvm::Cfg<NT>::UB + 4 * rows_pad * 2;
  hipLaunchKernelGGL(pv_probe<NT>, dim3(1), dim3(64), lds, 0, dv, ldv, col0, dp, rows_pad, L, dout);
  CK(hipDeviceSynchronize());
  std::vector<float> out(nout);
  CK(hipMemcpy(out.data(), dout, nout * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  int bad = 0;
  for (int h = 0; h < 4; ++h)
    for (int c = 0; c < 16 * NT; ++c) {
      double ref = 0;
      for (int r = 0; r < L; ++r) ref += (double)(float)p[(size_t)h * rows_pad + r] * (double)(float)v[(size_t)r * ldv + col0 + c];
      double e = fabs(ref - out[h * 16 * NT + c]);
      if (!(e <= 1e-3 * (1 + fabs(ref)))) ++bad;
      if (e > maxerr || e != e) maxerr = e;
    }
  printf("pv_probe NT=%d L=%d col0=%d: max abs err %.3g, bad %d / %d  -> %s\n", NT, L, col0, maxerr, bad, 4 * 16 * NT,
         bad == 0 ? "PASS" : "FAIL");
  if (bad) {
    printf("  first outputs (head 0): ");
    for (int c = 0; c < 8; ++c) printf("%.3f ", out[c]);
    printf("\n");
  }
  CK(hipFree(dv)); CK(hipFree(dp)); CK(hipFree(dout));
  return bad;
}

int main() {
  int fails = 0;
  {
    std::vector<short> in(1024), out(256);
    for (int i = 0; i < 1024; ++i) in[i] = (short)i;
    short *di, *dout;
    CK(hipMalloc(&di, 2048));
    CK(hipMalloc(&dout, 512));
    CK(hipMemcpy(di, in.data(), 2048, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 2048, 0, di, dout);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost));
    // assumed: within a 16-lane group, lane i elem j <- chunk of lane 4j + (i>>2), element i&3
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int grp = l & ~15, i = l & 15;
        const int expect = 4 * (grp + 4 * j + (i >> 2)) + (i & 3);
        if (out[l * 4 + j] != expect) ++bad;
      }
    printf("tr_read mapping: %s (%d mismatches)\n", bad ? "DIFFERENT FROM ASSUMPTION" : "as assumed", bad);
    if (bad) {
      for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf(" (src lane %2d, elem %d)", out[l * 4 + j] / 4, out[l * 4 + j] % 4);
        printf("\n");
      }
    }
    fails += bad != 0;
  }
  fails += run_pv<1>(150, 16) != 0;
  fails += run_pv<2>(150, 32) != 0;
  fails += run_pv<4>(150, 64) != 0;
  fails += run_pv<4>(160, 0) != 0;
  fails += run_pv<2>(33, 352) != 0;
  printf("probe_pv_mfma: %s\n", fails ? "FAILURES" : "ALL PASS");
  return fails ? 1 : 0;
}
