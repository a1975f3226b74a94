// One-token decode of the low-rank attention module as a fixed sequence of kernel launches on one
// stream (hipGraph-capturable): the drop-in for the decode branch of LlamaPaluAttention.forward
// (kernel/palu_attention.py:147-263, branch :207-219).
//   qkv GEMV + q-RoPE + cache append -> abx scores -> softmax.PV + split merge -> o_proj            (5 launches)
// or, where palu_decode_attn_preferred() says the single-kernel attention core is faster (one latent group per launch, or a very short cache;
// no attention weights requested; an additive mask is taken):  qkv -> fused scores/softmax/P.V (decode_fused.hip) + split merge -> o_proj.
#include "palu_common.h"

namespace {
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
struct StepWs {
  size_t q, scores, ctx, pv, knew, vnew, acc, qfold, total;
};
StepWs step_layout(int H, int G, int D, int Lcap, int Rv) {
  StepWs w;
  size_t o = 0;
  w.q = o;      o += align256((size_t)H * D * 2);
  w.scores = o; o += align256((size_t)H * ((size_t)Lcap + 8) * 2);
  w.ctx = o;    o += align256((size_t)H * Rv * 2);
  w.pv = o;     o += align256(palu_pv_workspace_bytes(H, G, Lcap, Rv));
  w.knew = o;   o += align256((size_t)4096 * 2);   // new latent rows before quantisation (G*Rk <= 4096)
  w.vnew = o;   o += align256((size_t)16384 * 2);
  const size_t acc_b = (size_t)H * (((size_t)Lcap + 8 + 7) & ~(size_t)7) * 4;            // fp32 scores of a multi-pass rank (> 128) ...
  w.acc = o;    o += align256(acc_b > (size_t)H * 32 * 128 ? acc_b : (size_t)H * 32 * 128);  // ... and never less than palu_abx_scratch_bytes of a rank <= 128
  w.qfold = o;  o += align256((size_t)H * 32 * 128);   // folded fragments of the position-split score kernel (palu_abx_fold_bytes: 4 KB per head and 16 columns, R <= 128)
  w.total = o;
  return w;
}
}  // namespace

// The step's score launch takes the position-split kernel on fragments the projection kernel folds in its tail
// (palu_decode_qkv_fold_f16 -> palu_abx_rope_pf_f16) whenever palu_abx_rope_ws_f16 would select that kernel anyway.
static bool step_prefold(int H, int G, int L, int Rk, int D, const float* inv_freq) {
  return D == 128 && palu_abx_fold_bytes(H, G, Rk) != 0 && palu_abx_set_fold(-1) != 0 &&
         palu_abx_position_split_selected(inv_freq, H, G, L, Rk, 0) != 0;
}

extern "C" size_t palu_decode_workspace_bytes(int H, int G, int D, int Lcap, int Rv) {
  if (H <= 0 || G <= 0 || D <= 0 || Lcap <= 0 || Rv <= 0) return 0;
  return step_layout(H, G, D, Lcap, Rv).total;
}

static int decode_step_impl(bool shared_b, const void* hidden, const void* wq, int64_t ldq, const void* vtk, int64_t ldk,
                            const void* vtv, int64_t ldv, const void* bfrag, const void* wo, int64_t ldo,
                            void* k_cache, int64_t sk_g, int64_t sk_l, void* v_cache, int64_t sv_g,
                            int64_t sv_l, const void* mask, const float* inv_freq, void* out, void* probs,
                            int64_t sp_h, void* workspace, int Lcap, int H, int G, int D, int hidden_size,
                            int Rk, int Rv, int cache_len, int pos, palu_stream_t stream) {
  PALU_REQUIRE(workspace && Lcap > cache_len && cache_len >= 0, PALU_ERR_ARG,
               "decode_step: cache_len %d must be < workspace capacity %d", cache_len, Lcap);
  const StepWs w = step_layout(H, G, D, Lcap, Rv);
  char* ws = (char*)workspace;
  void* q = ws + w.q;
  void* scores = ws + w.scores;
  void* ctx = ws + w.ctx;
  void* pvws = ws + w.pv;
  const int L = cache_len + 1;
  const int64_t ss_h = ((int64_t)Lcap + 8) & ~(int64_t)7;
  const bool fused = !shared_b && !probs && palu_decode_attn_preferred(H, G, L, Rk, Rv, D);
  const bool prefold = !shared_b && !fused && step_prefold(H, G, L, Rk, D, inv_freq);
  int rc = palu_decode_qkv_fold_f16(wq, ldq, nullptr, vtk, ldk, vtv, ldv, hidden, q, k_cache, sk_g, sk_l, v_cache, sv_g, sv_l,
                                    inv_freq, H, D, hidden_size, G, Rk, Rv, pos, cache_len, prefold ? bfrag : nullptr,
                                    prefold ? ws + w.qfold : nullptr, stream);
  if (rc) return rc;
  if (fused) {
    rc = palu_decode_attn_mask_f16(q, D, 1, bfrag, k_cache, sk_g, sk_l, v_cache, sv_g, sv_l, mask, ctx, pvws, H, G, L, Rk,
                                   Rv, D, inv_freq, 0, sqrtf((float)D), stream);
  } else {
    rc = shared_b ? palu_abx_rope_shared_f16(q, D, 1, bfrag, k_cache, sk_g, sk_l, scores, ss_h, H, G, L, Rk, D, inv_freq, 0, stream)
         : prefold ? palu_abx_rope_pf_f16(ws + w.qfold, k_cache, sk_g, sk_l, scores, ss_h, H, G, L, Rk, D, inv_freq, 0, stream)
                   : palu_abx_rope_ws_f16(q, D, 1, bfrag, k_cache, sk_g, sk_l, scores, ss_h, H, G, L, Rk, D, inv_freq, 0,
                                          ws + w.acc, stream);
    if (rc) return rc;
    rc = palu_softmax_pv_f16(scores, ss_h, mask, v_cache, sv_g, sv_l, ctx, probs, sp_h, pvws, H, G, L, Rv,
                             sqrtf((float)D), stream);
  }
  if (rc) return rc;
  return palu_gemv_f16(wo, ldo, ctx, out, hidden_size, H * Rv, stream);
}

extern "C" int palu_decode_step_f16(const void* hidden, const void* wq, int64_t ldq, const void* vtk, int64_t ldk,
                                    const void* vtv, int64_t ldv, const void* bfrag, const void* wo, int64_t ldo,
                                    void* k_cache, int64_t sk_g, int64_t sk_l, void* v_cache, int64_t sv_g,
                                    int64_t sv_l, const void* mask, const float* inv_freq, void* out, void* probs,
                                    int64_t sp_h, void* workspace, int Lcap, int H, int G, int D, int hidden_size,
                                    int Rk, int Rv, int cache_len, int pos, palu_stream_t stream) {
  return decode_step_impl(false, hidden, wq, ldq, vtk, ldk, vtv, ldv, bfrag, wo, ldo, k_cache, sk_g, sk_l, v_cache, sv_g, sv_l,
                          mask, inv_freq, out, probs, sp_h, workspace, Lcap, H, G, D, hidden_size, Rk, Rv, cache_len, pos, stream);
}

// The same step for weights whose heads share B inside a latent group (true GQA): `bfrag` are the
// fragments of the [G, R, D] shared factor (palu_abx_prepare_b(b_g, H := G, G)) and the scores come from
// palu_abx_rope_shared_f16 (keys reconstructed once per group).
extern "C" int palu_decode_step_sharedb_f16(const void* hidden, const void* wq, int64_t ldq, const void* vtk, int64_t ldk,
                                            const void* vtv, int64_t ldv, const void* bfrag_shared, const void* wo,
                                            int64_t ldo, void* k_cache, int64_t sk_g, int64_t sk_l, void* v_cache,
                                            int64_t sv_g, int64_t sv_l, const void* mask, const float* inv_freq, void* out,
                                            void* probs, int64_t sp_h, void* workspace, int Lcap, int H, int G, int D,
                                            int hidden_size, int Rk, int Rv, int cache_len, int pos, palu_stream_t stream) {
  return decode_step_impl(true, hidden, wq, ldq, vtk, ldk, vtv, ldv, bfrag_shared, wo, ldo, k_cache, sk_g, sk_l, v_cache, sv_g,
                          sv_l, mask, inv_freq, out, probs, sp_h, workspace, Lcap, H, G, D, hidden_size, Rk, Rv, cache_len, pos,
                          stream);
}

// The step without its last GEMV: everything that is local to a head-group shard; the caller
// all-gathers `ctx` ([H, Rv] fp16 for the H heads it owns) and runs o_proj on the gathered vector.
extern "C" int palu_decode_attend_f16(const void* hidden, const void* wq, int64_t ldq, const void* vtk, int64_t ldk,
                                      const void* vtv, int64_t ldv, const void* bfrag, void* k_cache, int64_t sk_g,
                                      int64_t sk_l, void* v_cache, int64_t sv_g, int64_t sv_l, const void* mask,
                                      const float* inv_freq, void* ctx, void* workspace, int Lcap, int H, int G, int D,
                                      int hidden_size, int Rk, int Rv, int cache_len, int pos, palu_stream_t stream) {
  PALU_REQUIRE(workspace && ctx && Lcap > cache_len && cache_len >= 0, PALU_ERR_ARG,
               "decode_attend: cache_len %d must be < workspace capacity %d", cache_len, Lcap);
  const StepWs w = step_layout(H, G, D, Lcap, Rv);
  char* ws = (char*)workspace;
  void* q = ws + w.q;
  void* scores = ws + w.scores;
  void* pvws = ws + w.pv;
  const int L = cache_len + 1;
  const int64_t ss_h = ((int64_t)Lcap + 8) & ~(int64_t)7;
  const bool fused = palu_decode_attn_preferred(H, G, L, Rk, Rv, D) != 0;
  const bool prefold = !fused && step_prefold(H, G, L, Rk, D, inv_freq);
  int rc = palu_decode_qkv_fold_f16(wq, ldq, nullptr, vtk, ldk, vtv, ldv, hidden, q, k_cache, sk_g, sk_l, v_cache, sv_g, sv_l,
                                    inv_freq, H, D, hidden_size, G, Rk, Rv, pos, cache_len, prefold ? bfrag : nullptr,
                                    prefold ? ws + w.qfold : nullptr, stream);
  if (rc) return rc;
  if (fused)
    return palu_decode_attn_mask_f16(q, D, 1, bfrag, k_cache, sk_g, sk_l, v_cache, sv_g, sv_l, mask, ctx, pvws, H, G, L, Rk,
                                     Rv, D, inv_freq, 0, sqrtf((float)D), stream);
  rc = prefold ? palu_abx_rope_pf_f16(ws + w.qfold, k_cache, sk_g, sk_l, scores, ss_h, H, G, L, Rk, D, inv_freq, 0, stream)
               : palu_abx_rope_ws_f16(q, D, 1, bfrag, k_cache, sk_g, sk_l, scores, ss_h, H, G, L, Rk, D, inv_freq, 0, ws + w.acc, stream);
  if (rc) return rc;
  return palu_softmax_pv_f16(scores, ss_h, mask, v_cache, sv_g, sv_l, ctx, nullptr, 0, pvws, H, G, L, Rv,
                             sqrtf((float)D), stream);
}

// Same step on a QUANTISED latent cache (3/4-bit codes + per-row (scale, zero); quant.hip layout):
// qkv GEMV -> quantise+pack the two new latent rows into row `cache_len` -> abx with in-register
// dequantisation -> softmax.PV on the codes -> o_proj.  6 launches.
extern "C" int palu_decode_step_q(const void* hidden, const void* wq, int64_t ldq, const void* vtk, int64_t ldk,
                                  const void* vtv, int64_t ldv, const void* bfrag, const void* wo, int64_t ldo,
                                  void* k_codes, int64_t skc_g, int64_t skc_l, void* k_meta, int64_t skm_g, int64_t skm_l,
                                  void* v_codes, int64_t svc_g, int64_t svc_l, void* v_meta, int64_t svm_g, int64_t svm_l,
                                  const void* mask, const float* inv_freq, void* out, void* probs, int64_t sp_h,
                                  void* workspace, int Lcap, int H, int G, int D, int hidden_size, int Rk, int Rv,
                                  int bits, int cache_len, int pos, palu_stream_t stream) {
  return palu_decode_step_qg(hidden, wq, ldq, vtk, ldk, vtv, ldv, bfrag, wo, ldo, k_codes, skc_g, skc_l, k_meta, skm_g, skm_l,
                             v_codes, svc_g, svc_l, v_meta, svm_g, svm_l, mask, inv_freq, out, probs, sp_h, workspace, Lcap, H,
                             G, D, hidden_size, Rk, Rv, bits, 0, cache_len, pos, stream);
}

// The same step when the latents are quantised in column groups of `group_size` (quantize_tensor(..., group_size),
// quant.py:11-13; --lt_group_size of utils.py:105): k_meta [G, Lcap, Rk / group_size, 2], v_meta [G, Lcap, Rv / group_size, 2]
// (group_size = 0: one pair per row).  The new rows are quantised group by group (each group a row of the quantiser),
// the score kernel looks the pair of a column up by its group, P.V runs once per column group.
extern "C" int palu_decode_step_qg(const void* hidden, const void* wq, int64_t ldq, const void* vtk, int64_t ldk,
                                   const void* vtv, int64_t ldv, const void* bfrag, const void* wo, int64_t ldo,
                                   void* k_codes, int64_t skc_g, int64_t skc_l, void* k_meta, int64_t skm_g, int64_t skm_l,
                                   void* v_codes, int64_t svc_g, int64_t svc_l, void* v_meta, int64_t svm_g, int64_t svm_l,
                                   const void* mask, const float* inv_freq, void* out, void* probs, int64_t sp_h,
                                   void* workspace, int Lcap, int H, int G, int D, int hidden_size, int Rk, int Rv,
                                   int bits, int group_size, int cache_len, int pos, palu_stream_t stream) {
  PALU_REQUIRE(workspace && Lcap > cache_len && cache_len >= 0, PALU_ERR_ARG,
               "decode_step_q: cache_len %d must be < workspace capacity %d", cache_len, Lcap);
  PALU_REQUIRE(group_size == 0 || (group_size > 0 && Rk % group_size == 0 && Rv % group_size == 0 && group_size % 32 == 0),
               PALU_ERR_UNSUPPORTED, "decode_step_qg: group_size %d must divide Rk %d and Rv %d and be a multiple of 32",
               group_size, Rk, Rv);
  PALU_REQUIRE((size_t)G * Rk <= 4096 && (size_t)G * Rv <= 16384, PALU_ERR_UNSUPPORTED, "decode_step_q: rank too large");
  const StepWs w = step_layout(H, G, D, Lcap, Rv);
  char* ws = (char*)workspace;
  void* q = ws + w.q;
  void* scores = ws + w.scores;
  void* ctx = ws + w.ctx;
  void* pvws = ws + w.pv;
  void* knew = ws + w.knew;
  void* vnew = ws + w.vnew;
  const int L = cache_len + 1;
  const int64_t ss_h = ((int64_t)Lcap + 8) & ~(int64_t)7;
  int rc = palu_decode_qkv_f16(wq, ldq, vtk, ldk, vtv, ldv, hidden, q, knew, Rk, 0, vnew, Rv, 0, inv_freq, H, D,
                               hidden_size, G, Rk, Rv, pos, 0, stream);
  if (rc) return rc;
  if (group_size == 0) {
    rc = palu_quantize_pack_kv(knew, Rk, (char*)k_codes + (int64_t)cache_len * skc_l, skc_g,
                               (h16*)k_meta + (int64_t)cache_len * skm_l, skm_g, Rk, vnew, Rv,
                               (char*)v_codes + (int64_t)cache_len * svc_l, svc_g,
                               (h16*)v_meta + (int64_t)cache_len * svm_l, svm_g, Rv, G, bits, stream);
  } else {
    // every column group of the new row is one row of the quantiser: [G, R / group_size, group_size]
    const int gb = group_size * bits / 8;
    rc = palu_quantize_pack(knew, Rk, group_size, (char*)k_codes + (int64_t)cache_len * skc_l, skc_g, gb,
                            (h16*)k_meta + (int64_t)cache_len * skm_l, skm_g, 2, nullptr, 0, 0, G, Rk / group_size, group_size,
                            bits, stream);
    if (rc) return rc;
    rc = palu_quantize_pack(vnew, Rv, group_size, (char*)v_codes + (int64_t)cache_len * svc_l, svc_g, gb,
                            (h16*)v_meta + (int64_t)cache_len * svm_l, svm_g, 2, nullptr, 0, 0, G, Rv / group_size, group_size,
                            bits, stream);
  }
  if (rc) return rc;
  rc = palu_abx_rope_qg(q, D, 1, bfrag, k_codes, skc_g, skc_l, k_meta, skm_g, skm_l, scores, ss_h, H, G, L, Rk, D, bits,
                        group_size, inv_freq, 0, ws + w.acc, stream);
  if (rc) return rc;
  rc = palu_softmax_pv_qg(scores, ss_h, mask, v_codes, svc_g, svc_l, v_meta, svm_g, svm_l, ctx, probs, sp_h, pvws, H, G, L,
                          Rv, bits, group_size, sqrtf((float)D), stream);
  if (rc) return rc;
  return palu_gemv_f16(wo, ldo, ctx, out, hidden_size, H * Rv, stream);
}
