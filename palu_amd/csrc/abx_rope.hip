// C-ABI entry points of the fused reconstruct-K -> RoPE -> q.K^T kernel for fp16 latents
// (kernels: abx_rope_kernel.h).  Replaces kernel/abx_rope.py:114-150 (`abx`).
#include "abx_rope_kernel.h"

namespace {

inline size_t abx_frag1_bytes(int G, const AbxPlan& pl) { return (size_t)G * pl.hb * 8 * pl.nmb * pl.nks_tot * 64 * sizeof(u32x4); }

template <int NKS, int NMB, bool FOLD>
int launch_abx_fast(const AbxParams& p, int nwg, hipStream_t stream) {
  // positions beyond 2^18: the second-order angle correction (abx_rope_kernel.h, ORDER2)
  if ((int64_t)p.pos0 + p.L > 262144) {
    return launch_kernel(abx_rope_kernel<NKS, NMB, FOLD, false, 0, true>, abx_smem_fast(NKS), p, nwg, stream);
  }
  return launch_kernel(abx_rope_kernel<NKS, NMB, FOLD>, abx_smem_fast(NKS), p, nwg, stream);
}

template <int NMB>
int launch_abx_generic(const AbxParams& p, int nwg, hipStream_t stream) {
  return launch_kernel(abx_rope_generic_kernel<8, NMB, true>, abx_smem_bytes(8, 2), p, nwg, stream);
}

int g_abx_fold = 1;

// ranks above 128 on the fast kernel: one launch per 128-column window of x (window kc: fragment k-steps 8 kc .. 8 kc + 7,
// columns beyond R masked), the partial scores accumulated in an fp32 array, one rounding at the end
template <int NMB, bool FOLD, int ACC>
int launch_abx_pass(const AbxParams& p, int nwg, hipStream_t stream) {
  if ((int64_t)p.pos0 + p.L > 262144)
    return launch_kernel(abx_rope_kernel<8, NMB, FOLD, false, 0, true, false, ACC>, abx_smem_fast(8), p, nwg, stream);
  return launch_kernel(abx_rope_kernel<8, NMB, FOLD, false, 0, false, false, ACC>, abx_smem_fast(8), p, nwg, stream);
}

__global__ void abx_round_kernel(const float* __restrict__ acc, int64_t acc_ld, h16* __restrict__ out, int64_t so_h, int H, int L) {
  const int h = blockIdx.y;
  for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x)
    out[(int64_t)h * so_h + l] = (h16)acc[(int64_t)h * acc_ld + l];
}

template <int NKS, int NMB>
int launch_abx_shared(const AbxParams& p, int nwg, hipStream_t stream) {
  if ((int64_t)p.pos0 + p.L > 262144) {
    return launch_kernel(abx_rope_kernel<NKS, NMB, false, false, 0, true, true>, abx_smem_fast(NKS), p, nwg, stream);
  }
  return launch_kernel(abx_rope_kernel<NKS, NMB, false, false, 0, false, true>, abx_smem_fast(NKS), p, nwg, stream);
}

}  // namespace

extern "C" int palu_abx_set_fold(int enable) {      // (enable < 0: query only)
  int o = g_abx_fold;
  if (enable >= 0) g_abx_fold = enable ? 1 : 0;
  return o;
}

extern "C" size_t palu_abx_bfrag_bytes(int H, int G, int R) {
  AbxPlan pl;
  if (!abx_plan(H, G, R, &pl)) return 0;
  // [fragments of abx_rope_kernel | fragments of the two-band kernel (gs = 4, R in {32, 64, 128}; abx_rope2_kernel.h)]
  return abx_frag1_bytes(G, pl) + palu_abx2_frag_bytes(H, G, R);
}

extern "C" int palu_abx_prepare_b(const void* b, int64_t sb_h, int64_t sb_r, int64_t sb_d, int H, int G, int R,
                                  int D, void* bfrag, palu_stream_t stream) {
  AbxPlan pl;
  PALU_REQUIRE(b && bfrag, PALU_ERR_ARG, "abx_prepare_b: null pointer");
  PALU_REQUIRE(abx_plan(H, G, R, &pl), PALU_ERR_ARG, "abx_prepare_b: bad shape H=%d G=%d R=%d", H, G, R);
  PALU_REQUIRE(D == HEAD_DIM, PALU_ERR_UNSUPPORTED, "abx: head_dim must be 128 (got %d)", D);
  PALU_REQUIRE(((uintptr_t)bfrag & 15) == 0, PALU_ERR_ARG, "abx_prepare_b: bfrag must be 16-byte aligned");
  int64_t total = (int64_t)G * pl.hb * 8 * pl.nmb * pl.nks_tot * 64;
  int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(abx_prepare_b_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const h16*)b, sb_h, sb_r, sb_d, H, G, R, pl.nmb, pl.hb, pl.nks_tot, (u32x4*)bfrag, total);
  PALU_LAUNCH_CHECK();
  return palu_abx2_prepare_b(b, sb_h, sb_r, sb_d, H, G, R, (char*)bfrag + abx_frag1_bytes(G, pl), (hipStream_t)stream);
}

const void* palu_abx_two_band_frags(const void* bfrag, int H, int G, int R) {
  AbxPlan pl;
  if (!bfrag || !abx_plan(H, G, R, &pl) || !palu_abx2_frag_bytes(H, G, R)) return nullptr;
  return (const char*)bfrag + abx_frag1_bytes(G, pl);
}

extern "C" size_t palu_abx_fold_bytes(int H, int G, int R) {
  if (G <= 0 || H != 4 * G || !(R == 32 || R == 64 || R == 128)) return 0;
  return (size_t)G * abx_fold_u32x4_per_group(R / 16) * sizeof(u32x4);
}

extern "C" size_t palu_abx_scratch_bytes(int H, int G, int L, int R) {
  AbxPlan pl;
  if (!abx_plan(H, G, R, &pl) || L <= 0) return 0;
  if (!pl.chunked) return palu_abx_fold_bytes(H, G, R);       // the folded fragments of the position-split kernel (0: other shapes)
  if (pl.nkc < 2) return 0;
  return (size_t)H * (((size_t)L + 7) & ~(size_t)7) * sizeof(float);
}

extern "C" int palu_abx_fold_f16(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, void* qfold, int H, int G, int R,
                                 palu_stream_t stream) {
  PALU_REQUIRE(a && bfrag && qfold, PALU_ERR_ARG, "abx_fold: null pointer");
  PALU_REQUIRE(palu_abx_fold_bytes(H, G, R) != 0, PALU_ERR_UNSUPPORTED,
               "abx_fold: needs 4 heads per group and R in {32, 64, 128} (H=%d G=%d R=%d)", H, G, R);
  PALU_REQUIRE((((uintptr_t)bfrag | (uintptr_t)qfold) & 15) == 0, PALU_ERR_ARG, "abx_fold: bfrag and qfold must be 16-byte aligned");
  return palu_abx3_fold_launch(a, sa_h, sa_d, palu_abx_two_band_frags(bfrag, H, G, R), qfold, H, G, R / 16, (hipStream_t)stream);
}

// scores from pre-folded fragments: the position-split kernel or nothing (the caller decided with
// palu_abx_position_split_selected when it asked for the fold)
extern "C" int palu_abx_rope_pf_f16(const void* qfold, const void* x, int64_t sx_g, int64_t sx_l, void* out, int64_t so_h,
                                    int H, int G, int L, int R, int D, const float* inv_freq, int pos0, palu_stream_t stream) {
  AbxPlan pl;
  PALU_REQUIRE(abx_plan(H, G, R, &pl) && palu_abx_fold_bytes(H, G, R) != 0, PALU_ERR_UNSUPPORTED,
               "abx_pf: needs 4 heads per group and R in {32, 64, 128} (H=%d G=%d R=%d)", H, G, R);
  PALU_REQUIRE(D == HEAD_DIM, PALU_ERR_UNSUPPORTED, "abx_pf: head_dim must be 128 (got %d)", D);
  PALU_REQUIRE(L >= 0, PALU_ERR_ARG, "abx_pf: negative L");
  if (L == 0) return PALU_OK;
  PALU_REQUIRE(qfold && x && out && inv_freq, PALU_ERR_ARG, "abx_pf: null pointer");
  PALU_REQUIRE((((uintptr_t)x | (uintptr_t)qfold) & 15) == 0 && sx_g % 8 == 0 && sx_l % 8 == 0 && sx_l >= R, PALU_ERR_ARG,
               "abx_pf: x rows and qfold must be 16-byte aligned, rows contiguous (sx_g=%lld sx_l=%lld R=%d)", (long long)sx_g,
               (long long)sx_l, R);
  PALU_REQUIRE(((int64_t)L + 3 * 128) * sx_l * 2 < ((int64_t)1 << 32), PALU_ERR_ARG,
               "abx_pf: one group's latent slab must stay below 4 GiB (L=%d sx_l=%lld)", L, (long long)sx_l);
  const int64_t ob = ((int64_t)(H - 1) * so_h + L) * 2;
  PALU_REQUIRE(ob > 0 && ob < 0xFFFFFFF0ll, PALU_ERR_UNSUPPORTED, "abx_pf: out extent must be < 4 GiB");
  PALU_REQUIRE(g_abx_fold != 0 && palu_abx_position_split_selected(inv_freq, H, G, L, R, pos0), PALU_ERR_UNSUPPORTED,
               "abx_pf: this launch does not take the position-split kernel (H=%d G=%d L=%d R=%d pos0=%d)", H, G, L, R, pos0);
  AbxParams p = {};
  p.x = (const h16*)x; p.sx_g = sx_g; p.sx_l = sx_l;
  p.out = (h16*)out; p.so_h = so_h; p.out_bytes = (unsigned)ob;
  p.inv_freq = inv_freq;
  const int nwg = abx_fill_params(p, pl, H, G, L, R, pos0);
  p.qfold = (const u32x4*)qfold;
  p.bfrag2 = p.qfold;                                 // (palu_abx2_try_launch insists on fragments; the PREFOLD kernel reads qfold only)
  const int rc = palu_abx2_try_launch(&p, nwg, 0, (hipStream_t)stream);
  PALU_REQUIRE(rc != PALU_ABX2_SKIP, PALU_ERR_UNSUPPORTED, "abx_pf: no coefficient table covers these positions");
  return rc;
}

extern "C" int palu_abx_rope_f16(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, const void* x,
                                 int64_t sx_g, int64_t sx_l, void* out, int64_t so_h, int H, int G, int L, int R,
                                 int D, const float* inv_freq, int pos0, palu_stream_t stream) {
  return palu_abx_rope_ws_f16(a, sa_h, sa_d, bfrag, x, sx_g, sx_l, out, so_h, H, G, L, R, D, inv_freq, pos0, nullptr, stream);
}

// The same with an optional fp32 scratch of palu_abx_scratch_bytes(H, G, L, R) bytes: ranks above 128 then run as
// ceil(R / 128) passes of the fast kernel (2.4x the chunked kernel's speed at R = 256); without it they take the
// chunked kernel.  Ranks <= 128 never need it.
extern "C" int palu_abx_rope_ws_f16(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, const void* x,
                                    int64_t sx_g, int64_t sx_l, void* out, int64_t so_h, int H, int G, int L, int R,
                                    int D, const float* inv_freq, int pos0, void* scratch, palu_stream_t stream) {
  AbxPlan pl;
  PALU_REQUIRE(abx_plan(H, G, R, &pl), PALU_ERR_ARG, "abx: bad shape H=%d G=%d R=%d", H, G, R);
  PALU_REQUIRE(D == HEAD_DIM, PALU_ERR_UNSUPPORTED, "abx: head_dim must be 128 (got %d)", D);
  PALU_REQUIRE(L >= 0, PALU_ERR_ARG, "abx: negative L");
  if (L == 0) return PALU_OK;
  PALU_REQUIRE(a && bfrag && x && out && inv_freq, PALU_ERR_ARG, "abx: null pointer");
  PALU_REQUIRE(((uintptr_t)x & 15) == 0 && sx_g % 8 == 0 && sx_l % 8 == 0 && sx_l >= R, PALU_ERR_ARG,
               "abx: x rows must be 16-byte aligned and contiguous (sx_g=%lld sx_l=%lld R=%d)", (long long)sx_g,
               (long long)sx_l, R);
  // the fast kernel stages X through a buffer descriptor with 32-bit byte offsets per group slab
  PALU_REQUIRE(((int64_t)L + 3 * 128) * sx_l * 2 < ((int64_t)1 << 32), PALU_ERR_ARG,
               "abx: one group's latent slab must stay below 4 GiB (L=%d sx_l=%lld)", L, (long long)sx_l);
  PALU_REQUIRE(((uintptr_t)bfrag & 15) == 0, PALU_ERR_ARG, "abx: bfrag must be 16-byte aligned");
  PALU_REQUIRE((int64_t)pos0 + L < (1 << 24), PALU_ERR_UNSUPPORTED, "abx: positions must stay below 2^24");
  const int64_t ob = ((int64_t)(H - 1) * so_h + L) * 2;
  PALU_REQUIRE(ob > 0 && ob < 0xFFFFFFF0ll, PALU_ERR_UNSUPPORTED, "abx: out extent must be < 4 GiB");

  AbxParams p = {};
  p.a = (const h16*)a; p.sa_h = sa_h; p.sa_d = sa_d;
  p.bfrag = (const u32x4*)bfrag;
  p.x = (const h16*)x; p.sx_g = sx_g; p.sx_l = sx_l;
  p.out = (h16*)out; p.so_h = so_h; p.out_bytes = (unsigned)ob;
  p.inv_freq = inv_freq;
  const int nwg = abx_fill_params(p, pl, H, G, L, R, pos0);
  hipStream_t s = (hipStream_t)stream;
  const bool fold = g_abx_fold != 0;
  if (pl.chunked && pl.nkc == 1 && ((int64_t)L + 3 * 128) * sx_l * 2 < ((int64_t)1 << 31)) {
    // a rank below 128 that is not 32 / 64 (96 of the rank search, 40, 72, ...): the 128-column fast kernel on the
    // zero-padded fragments, the columns beyond R masked in its tile staging -- 3x the chunked kernel's speed
    if (fold && palu_abx2_frag_bytes(H, G, R)) {
      // rank 96 at 4 heads per group: one 128-wide window of the two-band kernel, 96 valid columns (abx_rope2.hip)
      p.bfrag2 = (const u32x4*)((const char*)bfrag + abx_frag1_bytes(G, pl));
      const int rc2 = palu_abx2_try_launch_windows(&p, nwg, 0, nullptr, 0, s);
      if (rc2 != PALU_ABX2_SKIP) return rc2;
    }
    p.ncols = R;
    return fold ? (pl.nmb == 2 ? launch_abx_fast<8, 2, true>(p, nwg, s) : launch_abx_fast<8, 1, true>(p, nwg, s))
                : (pl.nmb == 2 ? launch_abx_fast<8, 2, false>(p, nwg, s) : launch_abx_fast<8, 1, false>(p, nwg, s));
  }
  if (pl.chunked && pl.nkc >= 2 && scratch && ((uintptr_t)scratch & 15) == 0 &&
      ((int64_t)L + 3 * 128) * sx_l * 2 < ((int64_t)1 << 31)) {
    const int64_t acc_ld = ((int64_t)L + 7) & ~(int64_t)7;
    if (fold && palu_abx2_frag_bytes(H, G, R)) {
      // 4 heads per group: the column windows (128, ..., 64, 32) through the two-band kernel when a coefficient table
      // covers the positions (abx_rope2.hip); fp32 partial scores in the scratch, one rounding below
      p.bfrag2 = (const u32x4*)((const char*)bfrag + abx_frag1_bytes(G, pl));
      const int rc2 = palu_abx2_try_launch_windows(&p, nwg, 0, scratch, acc_ld, s);
      if (rc2 != PALU_ABX2_SKIP) {
        if (rc2) return rc2;
        return PALU_OK;
      }
    }
    for (int kc = 0; kc < pl.nkc; ++kc) {
      AbxParams pk = p;
      pk.x = (const h16*)x + 128 * kc;                // window kc of every row (row stride unchanged)
      pk.ncols = R - 128 * kc < 128 ? R - 128 * kc : 128;
      pk.nks_frag = pl.nks_tot;
      pk.ks0 = 8 * kc;
      pk.acc = (float*)scratch;
      pk.acc_ld = acc_ld;
      int rc;
      if (kc == 0)
        rc = fold ? (pl.nmb == 2 ? launch_abx_pass<2, true, 1>(pk, nwg, s) : launch_abx_pass<1, true, 1>(pk, nwg, s))
                  : (pl.nmb == 2 ? launch_abx_pass<2, false, 1>(pk, nwg, s) : launch_abx_pass<1, false, 1>(pk, nwg, s));
      else
        rc = fold ? (pl.nmb == 2 ? launch_abx_pass<2, true, 2>(pk, nwg, s) : launch_abx_pass<1, true, 2>(pk, nwg, s))
                  : (pl.nmb == 2 ? launch_abx_pass<2, false, 2>(pk, nwg, s) : launch_abx_pass<1, false, 2>(pk, nwg, s));
      if (rc) return rc;
    }
    int bx = (L + 255) / 256;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(abx_round_kernel, dim3(bx, H), dim3(256), 0, s, (const float*)scratch, acc_ld, (h16*)out, so_h, H, L);
    PALU_LAUNCH_CHECK();
    return PALU_OK;
  }
  if (pl.chunked) return pl.nmb == 2 ? launch_abx_generic<2>(p, nwg, s) : launch_abx_generic<1>(p, nwg, s);
  if (fold && palu_abx2_frag_bytes(H, G, R)) {
    // gs = 4 at a fast rank: the two-band kernel when a coefficient table covers the positions (abx_rope2.hip)
    p.bfrag2 = (const u32x4*)((const char*)bfrag + abx_frag1_bytes(G, pl));
    if (scratch && ((uintptr_t)scratch & 15) == 0 && palu_abx_position_split_selected(inv_freq, H, G, L, R, pos0)) {
      // the position-split form on fragments folded once per launch (abx_fold.h) instead of once per workgroup
      const int rcf = palu_abx3_fold_launch(a, sa_h, sa_d, p.bfrag2, scratch, H, G, R / 16, s);
      if (rcf) return rcf;
      p.qfold = (const u32x4*)scratch;
    }
    const int rc = palu_abx2_try_launch(&p, nwg, 0, s);
    if (rc != PALU_ABX2_SKIP) return rc;
  }
#define PALU_ABX_DISPATCH(NKS)                                                                       \
  (fold ? (pl.nmb == 2 ? launch_abx_fast<NKS, 2, true>(p, nwg, s) : launch_abx_fast<NKS, 1, true>(p, nwg, s)) \
        : (pl.nmb == 2 ? launch_abx_fast<NKS, 2, false>(p, nwg, s) : launch_abx_fast<NKS, 1, false>(p, nwg, s)))
  switch (R) {
    case 32: return PALU_ABX_DISPATCH(2);
    case 64: return PALU_ABX_DISPATCH(4);
    default: return PALU_ABX_DISPATCH(8);
  }
#undef PALU_ABX_DISPATCH
}

// Debug/profiling entry (not part of the stable ABI): C2-class shapes only (R = 128, gs >= 3).
// dbg: [nwg][8 waves][64] cycle stamps (s_memtime), see the stamp() calls in abx_rope_kernel.
extern "C" int palu_abx_rope_f16_timed(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, const void* x,
                                       int64_t sx_g, int64_t sx_l, void* out, int64_t so_h, int H, int G, int L,
                                       int R, const float* inv_freq, int pos0, unsigned long long* dbg,
                                       int* nwg_out, palu_stream_t stream) {
  AbxPlan pl;
  PALU_REQUIRE(abx_plan(H, G, R, &pl) && R == 128 && pl.nmb == 2 && L > 0 && dbg, PALU_ERR_UNSUPPORTED,
               "abx timed: needs R=128, gs>=3");
  AbxParams p = {};
  p.a = (const h16*)a; p.sa_h = sa_h; p.sa_d = sa_d;
  p.bfrag = (const u32x4*)bfrag;
  p.x = (const h16*)x; p.sx_g = sx_g; p.sx_l = sx_l;
  p.out = (h16*)out; p.so_h = so_h;
  p.out_bytes = (unsigned)(((int64_t)(H - 1) * so_h + L) * 2);
  p.inv_freq = inv_freq;
  const int nwg = abx_fill_params(p, pl, H, G, L, R, pos0);
  p.dbg = dbg;
  if (nwg_out) *nwg_out = nwg;
  return launch_kernel(abx_rope_kernel<8, 2, true, true>, abx_smem_fast(8), p, nwg, (hipStream_t)stream);
}

// abx when every head of a latent group uses the same B (true-GQA: the query heads of a group share one KV head):
// bfrag = palu_abx_prepare_b of the shared factor b_g [G, R, D] (as H = G heads in G groups).  Same math and output as
// palu_abx_rope_f16 with b[h] = b_g[h / gs]; K is reconstructed once per group.  R in {32, 64, 128}, gs in {2, 3, 4}.
extern "C" int palu_abx_rope_shared_f16(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag_shared, const void* x,
                                        int64_t sx_g, int64_t sx_l, void* out, int64_t so_h, int H, int G, int L, int R,
                                        int D, const float* inv_freq, int pos0, palu_stream_t stream) {
  AbxPlan pl;
  PALU_REQUIRE(abx_plan(H, G, R, &pl), PALU_ERR_ARG, "abx_shared: bad shape H=%d G=%d R=%d", H, G, R);
  PALU_REQUIRE(D == HEAD_DIM, PALU_ERR_UNSUPPORTED, "abx_shared: head_dim must be 128 (got %d)", D);
  PALU_REQUIRE(!pl.chunked && pl.gs >= 2 && pl.gs <= 4, PALU_ERR_UNSUPPORTED,
               "abx_shared: needs R in {32,64,128} and 2..4 heads per group (R=%d gs=%d)", R, pl.gs);
  PALU_REQUIRE(L >= 0, PALU_ERR_ARG, "abx_shared: negative L");
  if (L == 0) return PALU_OK;
  PALU_REQUIRE(a && bfrag_shared && x && out && inv_freq, PALU_ERR_ARG, "abx_shared: null pointer");
  PALU_REQUIRE(((uintptr_t)x & 15) == 0 && sx_g % 8 == 0 && sx_l % 8 == 0 && sx_l >= R, PALU_ERR_ARG,
               "abx_shared: x rows must be 16-byte aligned and contiguous");
  PALU_REQUIRE(((int64_t)L + 3 * 128) * sx_l * 2 < ((int64_t)1 << 32), PALU_ERR_ARG,
               "abx_shared: one group's latent slab must stay below 4 GiB");
  PALU_REQUIRE(((uintptr_t)bfrag_shared & 15) == 0, PALU_ERR_ARG, "abx_shared: bfrag must be 16-byte aligned");
  PALU_REQUIRE((int64_t)pos0 + L < (1 << 24), PALU_ERR_UNSUPPORTED, "abx_shared: positions must stay below 2^24");
  const int64_t ob = ((int64_t)(H - 1) * so_h + L) * 2;
  PALU_REQUIRE(ob > 0 && ob < 0xFFFFFFF0ll, PALU_ERR_UNSUPPORTED, "abx_shared: out extent must be < 4 GiB");
  AbxParams p = {};
  p.a = (const h16*)a; p.sa_h = sa_h; p.sa_d = sa_d;
  p.bfrag = (const u32x4*)bfrag_shared;
  p.x = (const h16*)x; p.sx_g = sx_g; p.sx_l = sx_l;
  p.out = (h16*)out; p.so_h = so_h; p.out_bytes = (unsigned)ob;
  p.inv_freq = inv_freq;
  const int nwg = abx_fill_params(p, pl, H, G, L, R, pos0);
  p.prio_mode = abx_prio_mode(true);
  hipStream_t s = (hipStream_t)stream;
#define PALU_ABX_SH(NKS) (pl.nmb == 2 ? launch_abx_shared<NKS, 2>(p, nwg, s) : launch_abx_shared<NKS, 1>(p, nwg, s))
  switch (R) {
    case 32: return PALU_ABX_SH(2);
    case 64: return PALU_ABX_SH(4);
    default: return PALU_ABX_SH(8);
  }
#undef PALU_ABX_SH
}
