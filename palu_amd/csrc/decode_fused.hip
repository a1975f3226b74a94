// C-ABI entry of the fused attention core of a decode step (kernel: decode_fused_kernel.h):
//   scores (abx math) -> /sqrt(D) -> softmax -> latent P.V   in ONE kernel + the split merge.
// Replaces kernel/palu_attention.py:219 (recompute_k_gemv + /sqrt(D)), :238 (softmax) and :246-251 (latent P.V).
#include "decode_fused_kernel.h"

int palu_pv_combine_launch(float* ws, void* ctx, int H, int G, int Rv, int ns, hipStream_t s, int ctx_ld);   // decode_pv.hip

namespace {

template <int NKS, int NTS, int NTL, bool TIMING = false>
int launch_fused(const FusedParams& p, int nwg, hipStream_t stream) {
  constexpr int smem = (int)FusedLds<NKS, NTS, NTL>::TOTAL;
  static_assert(smem <= 160 * 1024, "fused decode kernel: LDS budget");
  auto kern = decode_fused_kernel<NKS, NTS, NTL, TIMING>;
  const int rc = palu_func_max_lds(reinterpret_cast<const void*>(kern), smem);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), smem, stream, p);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

// column plan: waves 4-7 own NTL col-tiles (16 latent columns each), waves 0-3 NTS; Rv = 64 * (NTS + NTL)
bool fused_col_plan(int Rv, int* nts, int* ntl) {
  switch (Rv) {
    case 384: *nts = 2; *ntl = 4; return true;
    case 192: *nts = 1; *ntl = 2; return true;
    case 256: *nts = 2; *ntl = 2; return true;
    case 128: *nts = 1; *ntl = 1; return true;
    default: return false;
  }
}

int g_fused_exp = -1;
int fused_exp_flags() {
  if (g_fused_exp < 0) {
    const char* e = palu_exp_env("PALU_FUSED_EXP");
    g_fused_exp = e ? atoi(e) : 0;
  }
  return g_fused_exp;
}

int fused_prio_mode() {   // 0 none, 2 score-block alternation (abx_rope_kernel.h), 3 (default) priority to the side work
  static int m = -1;
  if (m < 0) {
    const char* e = palu_exp_env("PALU_FUSED_PRIO");
    m = e ? atoi(e) : 3;
  }
  return m;
}

// PALU_FUSED_ATTN: 0 = never, 1 = whenever the shape is covered, unset = auto.  Measured on MI355X (same box, interleaved,
// tools/bench_fused_variants.py, profiles/r04_fused_vs_two_kernel_policy_sweep.txt), fused vs two kernels in us, round 4
// (two-band score kernel in the two-kernel path):
//   G=1: L=16k 24.2/27.0   64k 40.1/39.9   128k 57.0/63.3   256k 94.0/104.0
//   G=2: 16k 29.7/28.6   32k 38.9/37.3   64k 55.9/54.5   128k 92.8/85.9
//   G=4: 8k 28.5/27.3   16k 37.8/36.6   32k 57.5/55.0   64k 94.7/80.6
//   G=8: 2k 20.7/24.2   4k 25.6/26.6   8k 37.5/34.3   16k 58.8/51.6   32k 95.9/81.8
// i.e. it wins with ONE latent group per launch (every rank of an 8-GPU head-group sharding) and on very short caches,
// where its one launch less counts; from G = 2 and a few ten thousand rows per launch the 160 KB of LDS cannot hold enough
// latent rows in flight next to the score pipeline's tiles and the two kernels win.
int fused_mode() {
  static int m = -2;
  if (m == -2) {
    const char* e = getenv("PALU_FUSED_ATTN");
    m = e ? (atoi(e) != 0 ? 1 : 0) : -1;
  }
  return m;
}

}  // namespace

// debug knob (not part of the stable ABI): experiment flags of the fused kernel, see FusedParams::exp_flags
extern "C" int palu_decode_attn_set_exp(int flags) {
  const int o = fused_exp_flags();
  g_fused_exp = flags;
  return o;
}

extern "C" int palu_decode_attn_supported(int H, int G, int Rk, int Rv, int D) {
  if (H <= 0 || G <= 0 || H % G != 0 || D != HEAD_DIM) return 0;
  const int gs = H / G;
  int nts, ntl;
  return (gs == 3 || gs == 4) && (Rk == 64 || Rk == 128) && fused_col_plan(Rv, &nts, &ntl);
}

extern "C" int palu_decode_attn_preferred(int H, int G, int L, int Rk, int Rv, int D) {
  if (!palu_decode_attn_supported(H, G, Rk, Rv, D) || L <= 0) return 0;
  const int m = fused_mode();
  // round 4 (the two-kernel path runs the two-band score kernel): one latent group per launch -- the per-GPU slice of an
  // 8-way head-group sharding -- or very short caches (profiles/r04_fused_vs_two_kernel_policy_sweep.txt).  Round 5: the
  // two-band kernel covers 2^18 + 4096 positions, and from ~200k positions of ONE group on the two kernels win again
  // (262 145 positions: 85.3 us against 90.5 fused; 131k: 63.3 against 57.0 -- profiles/r05_c5_slice_policy.txt)
  return m >= 0 ? m : (G == 1 ? L <= 196608 : (int64_t)G * L <= 24576);
}

extern "C" int palu_decode_attn_nsplit(int G, int L) {
  if (G <= 0 || L <= 0) return 0;
  const int nt = (L + FTL - 1) / FTL;
  int nch = palu_num_cus() / G;
  if (nch < 1) nch = 1;
  if (nch > nt) nch = nt;
  return nch;
}

extern "C" size_t palu_decode_attn_stats_offset(int H, int G, int L, int Rv) {
  (void)H; (void)G; (void)L; (void)Rv;
  return 0;     // (max, sum) pairs lead the workspace (pv_ws_stats_floats, palu_common.h)
}

extern "C" int palu_decode_attn_f16(const void* q, int64_t sq_h, int64_t sq_d, const void* bfrag, const void* k,
                                    int64_t sk_g, int64_t sk_l, const void* v, int64_t sv_g, int64_t sv_l, void* ctx,
                                    void* workspace, int H, int G, int L, int Rk, int Rv, int D,
                                    const float* inv_freq, int pos0, float sqrt_d, palu_stream_t stream) {
  return palu_decode_attn_mask_f16(q, sq_h, sq_d, bfrag, k, sk_g, sk_l, v, sv_g, sv_l, nullptr, ctx, workspace, H, G, L, Rk,
                                   Rv, D, inv_freq, pos0, sqrt_d, stream);
}

// the same with an additive attention mask [L] fp16 (kernel/palu_attention.py:229-234; null = none)
extern "C" int palu_decode_attn_mask_f16(const void* q, int64_t sq_h, int64_t sq_d, const void* bfrag, const void* k,
                                         int64_t sk_g, int64_t sk_l, const void* v, int64_t sv_g, int64_t sv_l,
                                         const void* mask, void* ctx, void* workspace, int H, int G, int L, int Rk, int Rv,
                                         int D, const float* inv_freq, int pos0, float sqrt_d, palu_stream_t stream) {
  PALU_REQUIRE(palu_decode_attn_supported(H, G, Rk, Rv, D), PALU_ERR_UNSUPPORTED,
               "decode_attn: shape H=%d G=%d Rk=%d Rv=%d D=%d not covered by the fused kernel", H, G, Rk, Rv, D);
  PALU_REQUIRE(L > 0, PALU_ERR_ARG, "decode_attn: L must be positive");
  PALU_REQUIRE(q && bfrag && k && v && ctx && workspace && inv_freq, PALU_ERR_ARG, "decode_attn: null pointer");
  PALU_REQUIRE(((uintptr_t)k & 15) == 0 && sk_g % 8 == 0 && sk_l % 8 == 0 && sk_l >= Rk, PALU_ERR_ARG,
               "decode_attn: latent-K rows must be 16-byte aligned");
  PALU_REQUIRE(((uintptr_t)v & 15) == 0 && sv_g % 8 == 0 && sv_l % 8 == 0 && sv_l >= Rv, PALU_ERR_ARG,
               "decode_attn: latent-V rows must be 16-byte aligned");
  PALU_REQUIRE(((int64_t)L + 3 * 128) * sk_l * 2 < ((int64_t)1 << 32) && ((int64_t)L + 3 * 128) * sv_l * 2 < ((int64_t)1 << 32),
               PALU_ERR_ARG, "decode_attn: one group's latent slab must stay below 4 GiB");
  PALU_REQUIRE(((uintptr_t)bfrag & 15) == 0, PALU_ERR_ARG, "decode_attn: bfrag must be 16-byte aligned");
  PALU_REQUIRE((int64_t)pos0 + L < (1 << 24), PALU_ERR_UNSUPPORTED, "decode_attn: positions must stay below 2^24");
  FusedParams p = {};
  p.a = (const h16*)q; p.sa_h = sq_h; p.sa_d = sq_d;
  p.bfrag = (const u32x4*)bfrag;
  p.x = (const h16*)k; p.sx_g = sk_g; p.sx_l = sk_l;
  p.v = (const h16*)v; p.sv_g = sv_g; p.sv_l = sv_l;
  p.mask = (const h16*)mask;
  p.inv_freq = inv_freq;
  p.H = H; p.G = G; p.gs = H / G; p.L = L; p.R = Rk; p.Rv = Rv; p.pos0 = pos0;
  p.nt_total = (L + FTL - 1) / FTL;
  p.nch = palu_decode_attn_nsplit(G, L);
  p.sqrt_d = sqrt_d;
  p.prio_mode = fused_prio_mode();
  p.exp_flags = fused_exp_flags();
  p.dbg = nullptr;
  float* ws = (float*)workspace;
  const int ns = p.nch;
  p.part = ws + pv_ws_stats_floats(H);
  p.ml = p.part + (size_t)H * ns * Rv;
  hipStream_t s = (hipStream_t)stream;
  const int nwg = ns * G;
  int nts = 0, ntl = 0;
  fused_col_plan(Rv, &nts, &ntl);
  int rc;
#define PALU_FUSED(NKS)                                                    \
  (ntl == 4   ? launch_fused<NKS, 2, 4>(p, nwg, s)                         \
   : nts == 2 ? launch_fused<NKS, 2, 2>(p, nwg, s)                         \
   : ntl == 2 ? launch_fused<NKS, 1, 2>(p, nwg, s)                         \
              : launch_fused<NKS, 1, 1>(p, nwg, s))
  if (Rk == 128) rc = PALU_FUSED(8);
  else rc = PALU_FUSED(4);
#undef PALU_FUSED
  if (rc) return rc;
  return palu_pv_combine_launch(ws, ctx, H, G, Rv, ns, s, 0);
}

// Debug/profiling entry (not part of the stable ABI): C2-class shape only (Rk = 128, Rv = 384); per-wave cycle stamps
// dbg [nwg][8 waves][64]: 0 start, 1 B loads issued, 2 rope init, 3 fold, 4 first DMAs landed, then per step (arrive at
// the barrier, leave it), loop end, kernel end.
extern "C" int palu_decode_attn_f16_timed(const void* q, int64_t sq_h, int64_t sq_d, const void* bfrag, const void* k,
                                          int64_t sk_g, int64_t sk_l, const void* v, int64_t sv_g, int64_t sv_l,
                                          void* workspace, int H, int G, int L, int Rk, int Rv, const float* inv_freq,
                                          int pos0, float sqrt_d, unsigned long long* dbg, int* nwg_out,
                                          palu_stream_t stream) {
  PALU_REQUIRE(Rk == 128 && Rv == 384 && H / G == 4 && dbg && L > 0, PALU_ERR_UNSUPPORTED, "decode_attn timed: C2-class shape only");
  FusedParams p = {};
  p.a = (const h16*)q; p.sa_h = sq_h; p.sa_d = sq_d;
  p.bfrag = (const u32x4*)bfrag;
  p.x = (const h16*)k; p.sx_g = sk_g; p.sx_l = sk_l;
  p.v = (const h16*)v; p.sv_g = sv_g; p.sv_l = sv_l;
  p.inv_freq = inv_freq;
  p.H = H; p.G = G; p.gs = H / G; p.L = L; p.R = Rk; p.Rv = Rv; p.pos0 = pos0;
  p.nt_total = (L + FTL - 1) / FTL;
  p.nch = palu_decode_attn_nsplit(G, L);
  p.sqrt_d = sqrt_d;
  p.prio_mode = fused_prio_mode();
  p.exp_flags = fused_exp_flags();
  p.dbg = dbg;
  float* ws = (float*)workspace;
  p.part = ws + pv_ws_stats_floats(H);
  p.ml = p.part + (size_t)H * p.nch * Rv;
  if (nwg_out) *nwg_out = p.nch * G;
  return launch_fused<8, 2, 4, true>(p, p.nch * G, (hipStream_t)stream);
}
