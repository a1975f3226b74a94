// Host side of the position-split two-band score kernel (abx_rope3_kernel.h); selected by abx_rope2.hip's launch
// decision (palu_abx2_try_launch) for fp16 latents at R in {32, 64, 128}.  Replaces the Triton `_abx_fwd`
// (kernel/abx_rope.py:79-111) like the other score kernels.  Built with -fno-slp-vectorize (palu_amd/build.py): packed
// fp32 VALU operations are not single issue slots on gfx950 (profiles/r03_abx_packed_fp32_epilogue.txt).
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
    return launch_kernel(abx_rope3_kernel<NKS, true>, abx3_smem(NKS), p, nch * p.G, stream, ABX3_THREADS);
  }
#endif
  return launch_kernel(abx_rope3_kernel<NKS, false>, abx3_smem(NKS), p, nch * p.G, stream, ABX3_THREADS);
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
