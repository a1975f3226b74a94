// Host side of the position-split two-band score kernel (abx_rope3_kernel.h); selected by abx_rope2.hip's launch
// decision (palu_abx2_try_launch) for fp16 latents at R in {32, 64, 128}.  Replaces the Triton `_abx_fwd`
// (kernel/abx_rope.py:79-111) like the other score kernels.  Built with -fno-slp-vectorize (palu_amd/build.py): packed
// fp32 VALU operations are not single issue slots on gfx950 (profiles/r03_abx_packed_fp32_epilogue.txt).
// Also the stand-alone query fold (abx_fold.h): one wave per (head, RoPE pair).
#include "abx_rope3_kernel.h"

namespace {
unsigned long long* g_abx3_timeline = nullptr;

template <int NKS>
int launch3(AbxParams p, hipStream_t stream) {
  int nch = palu_num_cus() / p.G;                            // one 4-wave workgroup per CU, nch workgroups per latent group
  if (nch < 1) nch = 1;
  const int ntiles = (p.L + TL - 1) / TL;
  if (nch > (ntiles + 3) / 4) nch = (ntiles + 3) / 4;      // (no workgroup without a tile)
  p.nch = nch;
#ifdef PALU_EXPERIMENTS
  if (g_abx3_timeline) {
    p.dbg = g_abx3_timeline;
    if (p.qfold) return launch_kernel(abx_rope3_kernel<NKS, true, true>, abx3_smem(NKS), p, nch * p.G, stream, ABX3_THREADS);
    return launch_kernel(abx_rope3_kernel<NKS, true, false>, abx3_smem(NKS), p, nch * p.G, stream, ABX3_THREADS);
  }
#endif
  if (p.qfold) return launch_kernel(abx_rope3_kernel<NKS, false, true>, abx3_smem(NKS), p, nch * p.G, stream, ABX3_THREADS);
  return launch_kernel(abx_rope3_kernel<NKS, false, false>, abx3_smem(NKS), p, nch * p.G, stream, ABX3_THREADS);
}

// one wave per (head, pair): 4 waves per workgroup
__global__ __launch_bounds__(256) void abx_fold_kernel(const h16* __restrict__ a, int64_t sa_h, int64_t sa_d,
                                                       const u32x4* __restrict__ bfrag2, u32x4* __restrict__ qfold, int H, int G,
                                                       int nks) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= H * 64) return;
  const int h = pair >> 6, i = pair & 63;
  const AbxFoldSrc src = abx_fold_load(bfrag2, G, nks, h >> 2, h & 3, i, lane);
  const h16 qa = a[(int64_t)h * sa_h + (int64_t)i * sa_d], qb = a[(int64_t)h * sa_h + (int64_t)(i + 64) * sa_d];
  abx_fold_store(src, qfold, nks, h >> 2, h & 3, i, qa, qb, lane);
}
}  // namespace

// debug (PALU_EXPERIMENTS builds): device buffer [workgroups][4 waves][64] of s_memtime stamps the next position-split
// launches fill (abx_rope3_kernel TIMING); 0 = off
extern "C" void palu_abx3_timeline_buffer(void* ptr) { g_abx3_timeline = (unsigned long long*)ptr; }

int palu_abx3_launch(const void* params, int nks, hipStream_t stream) {
  const AbxParams& p = *reinterpret_cast<const AbxParams*>(params);
  switch (nks) {
    case 2: return launch3<2>(p, stream);
    case 4: return launch3<4>(p, stream);
    default: return launch3<8>(p, stream);
  }
}

int palu_abx3_fold_launch(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag2, void* qfold, int H, int G, int nks,
                          hipStream_t stream) {
  if (!a || !bfrag2 || !qfold) {
    palu_set_error("abx fold: null pointer");
    return PALU_ERR_ARG;
  }
  hipLaunchKernelGGL(abx_fold_kernel, dim3((unsigned)(H * 16)), dim3(256), 0, stream, (const h16*)a, sa_h, sa_d,
                     (const u32x4*)bfrag2, (u32x4*)qfold, H, G, nks);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}
