/*
 * palu_hip.h -- C ABI of the MI355X (gfx950) low-rank-KV attention decode path.
 *
 * The reference (shadowpa0327/Palu) is pure Python; its boundary for this path is the Python
 * call `kernel.abx_rope.abx(a, b, x)` (kernel/abx_rope.py:114-150, called at
 * kernel/palu_attention.py:219) plus the stock torch ops of the decode branch
 * (kernel/palu_attention.py:207-257).  This header is what a ctypes/cffi stub binds instead
 * (see INTEGRATION.md): plain pointers + sizes, caller-owned DEVICE buffers, a HIP stream.
 *
 * Conventions for every entry point:
 *   - pointers are device pointers unless the name ends in _host;
 *   - strides are in ELEMENTS of the pointed-to type;
 *   - work is enqueued on `stream` (a hipStream_t cast to void*; NULL = default stream),
 *     nothing synchronises, nothing allocates: every call is hipGraph-capturable
 *     (run_latency_attention.py:81-90 captures the step in a graph);
 *   - returns 0 on success or a negative PALU_ERR_* code; palu_last_error() gives the text.
 *     Never throws, never aborts.
 */
#ifndef PALU_HIP_H
#define PALU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PALU_OK 0
#define PALU_ERR_ARG (-1)      /* bad shape / stride / alignment / null pointer          */
#define PALU_ERR_UNSUPPORTED (-2) /* valid but not implemented (e.g. head_dim != 128)    */
#define PALU_ERR_LAUNCH (-3)   /* HIP reported a launch error                            */

typedef void* palu_stream_t;

const char* palu_last_error(void);
int palu_version(void);

/* ------------------------------------------------------------------------------------------
 * RoPE frequencies.  kernel/pytorch_reference.py:4: inv_freq[i] = 1 / theta^(2i/D) in fp32.
 * Host helper (fills D/2 floats).  The kernels take the table as a device pointer so that a
 * caller can pass exactly the values its framework computed.
 */
int palu_rope_inv_freq_host(float theta, int head_dim, float* out_host);

/* ------------------------------------------------------------------------------------------
 * abx: fused reconstruct-K -> RoPE -> q.K^T   (replaces kernel/abx_rope.py:44-150)
 *
 *   out[h, l] = sum_d a[h,d] * RoPE_{pos0+l}( sum_r x[h/gs, l, r] * b[h, r, d] )[d]
 *
 * a: [H, D] fp16 (the already-rotated query; reference shape [H,1,D]), b: [H, R, D] fp16,
 * x: [G, L, R] fp16 latent keys (row l = absolute position pos0 + l), out: [H, L] fp16.
 * No 1/sqrt(D) (palu_attention.py:219 divides afterwards).  D must be 128 (the reference
 * hard-codes it, abx_rope.py:21-22,125); R % 8 == 0; H % G == 0; L >= 0 arbitrary (masked tail).
 *
 * `b` is a weight: it is re-laid-out once into MFMA A-operand fragments by palu_abx_prepare_b
 * (bfrag must hold palu_abx_bfrag_bytes() bytes) and the hot call consumes the fragments.
 * x rows must be 16-byte aligned (sx_g % 8 == 0, sx_l % 8 == 0, innermost stride 1).
 */
size_t palu_abx_bfrag_bytes(int H, int G, int R);
/* Numerics switch of the R in {32,64,128} fast path (process-wide, returns the previous value):
 * 1 (default) folds the query into the B fragments once per launch -- one extra fp16 operand
 * rounding, same size as the oracle's own rounding of K (abx_rope.py:164); 0 keeps q in fp32; a negative value only queries. */
int palu_abx_set_fold(int enable);
int palu_abx_prepare_b(const void* b, int64_t sb_h, int64_t sb_r, int64_t sb_d,
                       int H, int G, int R, int D, void* bfrag, palu_stream_t stream);
int palu_abx_rope_f16(const void* a, int64_t sa_h, int64_t sa_d,
                      const void* bfrag,
                      const void* x, int64_t sx_g, int64_t sx_l,
                      void* out, int64_t so_h,
                      int H, int G, int L, int R, int D,
                      const float* inv_freq, int pos0, palu_stream_t stream);
/* Ranks above 128 (R % 8 == 0) run as ceil(R / 128) passes of the 128-column kernel when the caller provides an fp32
 * scratch of palu_abx_scratch_bytes(H, G, L, R) bytes (16-byte aligned; 0 for R <= 128): partial scores are accumulated
 * in fp32 and rounded once.  Without scratch (palu_abx_rope_f16, or scratch = 0) such ranks take the slower chunked kernel.
 * Ranks below 128 other than 32 / 64 (e.g. 96) always run the 128-column kernel with the missing columns masked. */
/* Low-band RoPE coefficient table of the two-band score kernel (csrc/abx_rope2_kernel.h).  With 4 heads per latent group
 * and R in {32, 64, 128} -- or rank 96 / a multiple of 32 above 128, as column windows of those widths (128-wide ones, then
 * 32 or 64, or a 128-wide window with 96 valid columns; fp32 partial scores in the scratch, the last window rounds) --
 * the abx entry points (fp16 and packed latents, and the decode steps built on them) run a kernel
 * that treats the 32 low-frequency RoPE pairs of a 128-position tile as a degree-7 polynomial in the in-tile position:
 * a quarter of the reconstruction GEMM and of the per-position rotation work disappears.  It needs, per 128-position
 * tile, the fp16 coefficients (psi_i/psi_max)^k cos/sin(phi_i + k pi/2) of the tile's centre angle -- a function of the
 * positions and frequencies only (like the cos/sin cache of kernel/pytorch_reference.py:3-9), built once:
 *   table = palu_rope_table_bytes(npos) bytes, 16-byte aligned, filled by palu_rope_table_build for the positions
 *   [pos_first, pos_first + npos), pos_first % 128 == 0;
 *   palu_rope_table_register ties it to the DEVICE POINTER the caller passes as `inv_freq` (the registry key; the table
 *   must stay alive while registered; inv_freq_32 = the host value of inv_freq[32], which bounds the band's angles).
 * A launch takes the two-band kernel when a registered table covers its positions, pos0 % 128 == 0, pos0 + L <= 2^18 + 4096 and
 * inv_freq[32] * (pos0 + L) < 2700 rad (the band uses the exact angle l*f; the oracle's fp32 rounding of l*f is <= 2^-13 rad
 * there, measured at 262 145 positions: error and rms at the oracle's own level) and 64 * inv_freq[32] <= 0.7 rad (the polynomial's remainder: theta >= ~8400 at head_dim 128); otherwise, or with PALU_ABX_TWO_BAND=0 in the environment, it runs the one-band kernel -- same results within
 * the oracle's own fp16 rounding.  palu_abx_two_band_selected reports the decision for a launch. */
/* (palu_rope_table_register takes the SAME (pos_first, npos) the table was built with: the start tables of the position-split
 * kernel follow the table's own coefficient tiles; a table built in this process is checked, a mismatch is PALU_ERR_ARG.) */
size_t palu_rope_table_bytes(int npos);
int palu_rope_table_build(const float* inv_freq, int pos_first, int npos, void* table, palu_stream_t stream);
int palu_rope_table_register(const float* inv_freq, const void* table, int pos_first, int npos, float inv_freq_32);
int palu_rope_table_unregister(const float* inv_freq);
int palu_abx_two_band_selected(const float* inv_freq, int H, int G, int L, int R, int pos0);
/* The two-band kernel has two forms with the same algebra, fragments and table: the position-split one
 * (csrc/abx_rope3_kernel.h: 4 waves per workgroup, a wave owns whole 32-position blocks for all RoPE pairs, high-band
 * fragments in AGPRs, no cross-wave reduction) and the pair-split one (csrc/abx_rope2_kernel.h: 8 waves, 4 pairs each,
 * per-tile LDS reduction).  The first needs 2/3 of the cycles per tile and a longer prologue: it is selected (fp16 latents,
 * one launch) from n = 1 tile per wave on.  palu_abx_set_position_split(n): 0 = never (PALU_ABX_SPLIT=0 in the environment
 * starts there), n >= 1 = from n tiles per wave on, n < 0 = on every shape the kernel takes (tests, A/B); returns the previous setting
 * (in the same encoding).
 * Process-wide, for A/B measurements. */
int palu_abx_set_position_split(int enable);
int palu_abx_position_split_selected(const float* inv_freq, int H, int G, int L, int R, int pos0);
/* The query fold as its own step (round 6; csrc/abx_fold.h).  The position-split kernel multiplies the latents with
 * P[r,i] = q_i B[r,i] + q_{i+64} B[r,i+64], Q[r,i] = q_{i+64} B[r,i] - q_i B[r,i+64] -- the weight of the reference's
 * `_abx_fwd` (kernel/abx_rope.py:79-111) with the query moved onto B.  Folding inside the kernel repeated the same work in every
 * workgroup of a latent group; it is now done once per launch, by one wave per (head, RoPE pair):
 *   palu_abx_fold_bytes(H, G, R)  bytes of the folded-fragment buffer `qfold` (16 R KB per group; 0 = shape without the
 *                                 position-split kernel: it needs H == 4 G and R in {32, 64, 128});
 *   palu_abx_fold_f16             a [H, D] (strides sa_h, sa_d), bfrag from palu_abx_prepare_b -> qfold: the stand-alone fold;
 *   palu_decode_qkv_fold_f16      (below) the decode step's form: the q waves of the projection kernel fold their own pair;
 *   palu_abx_rope_pf_f16          scores from the folded fragments: same operands as palu_abx_rope_f16 with `qfold` in the
 *                                 place of (a, bfrag).  PALU_ERR_UNSUPPORTED when the launch would not take the position-split
 *                                 kernel (palu_abx_position_split_selected(...) == 0, or palu_abx_set_fold(0)).
 * palu_abx_rope_ws_f16 runs fold + kernel itself when its scratch holds palu_abx_scratch_bytes() bytes (which now covers
 * palu_abx_fold_bytes()); without scratch (palu_abx_rope_f16) the kernel folds in its own prologue as in round 5.  All three
 * forms produce bit-identical scores (the same v_dot2_f32_f16 arithmetic on the same operands). */
size_t palu_abx_fold_bytes(int H, int G, int R);
int palu_abx_fold_f16(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, void* qfold, int H, int G, int R,
                      palu_stream_t stream);
int palu_abx_rope_pf_f16(const void* qfold, const void* x, int64_t sx_g, int64_t sx_l, void* out, int64_t so_h,
                         int H, int G, int L, int R, int D, const float* inv_freq, int pos0, palu_stream_t stream);
size_t palu_abx_scratch_bytes(int H, int G, int L, int R);
int palu_abx_rope_ws_f16(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag,
                         const void* x, int64_t sx_g, int64_t sx_l, void* out, int64_t so_h,
                         int H, int G, int L, int R, int D, const float* inv_freq, int pos0, void* scratch,
                         palu_stream_t stream);

/* Shared-B fast path: when b[h] is identical for the gs heads of every group (true-GQA checkpoints,
 * palu/model/svd_mistral/modeling_palu_mistral.py:37-59) the keys are reconstructed once per group instead of once per
 * head -- a quarter of the MFMA work, which makes the kernel VALU/HBM- instead of MFMA-bound.  bfrag_shared =
 * palu_abx_prepare_b(b_g [G, R, D], H := G, G) of palu_abx_bfrag_bytes(G, G, R) bytes.  Same output as
 * palu_abx_rope_f16 with b[h] = b_g[h / gs] (q is kept in fp32, like palu_abx_set_fold(0)).  R in {32,64,128}, gs 2..4. */
int palu_abx_rope_shared_f16(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag_shared,
                             const void* x, int64_t sx_g, int64_t sx_l, void* out, int64_t so_h,
                             int H, int G, int L, int R, int D, const float* inv_freq, int pos0, palu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * softmax + latent-space P.V   (replaces kernel/palu_attention.py:219 "/sqrt(D)", :229-234 mask,
 * :238 softmax(fp32)->fp16, :246-251 attn[1,G,gs,L] @ V_lat[1,G,L,Rv])
 *
 *   x[h,l]  = fp16(fp16(scores[h,l]) / sqrt_d) (+ mask[l])          -- the reference's fp16 tensors
 *   ctx[h,:] = sum_l softmax_l(x[h,:])[l] * v[h/gs, l, :]            -- fp32 math, one fp16 rounding
 *
 * scores: [H, L] fp16 (abx output), mask: [L] fp16 additive or NULL, v: [G, L, Rv] fp16 with
 * 16-byte aligned rows, ctx: [H, Rv] fp16 (= the [1,1,H*Rv] o_proj input), probs: [H, L] fp16 or
 * NULL (the attn_weights of output_attentions=True).  workspace: palu_pv_workspace_bytes() bytes.
 * gs = H/G in {1,2,4,8}; Rv % 8 == 0.
 * palu_pv_workspace_bytes(H, G, Lcap, Rv) is an upper bound over every L <= Lcap (the split count is not
 * monotone in L), so a workspace sized for a cache capacity serves every fill level.
 */
int palu_pv_nsplit(int G, int L);
/* split count of the register-direct matrix-core kernel behind palu_softmax_pv_q (bits 3 / 4) and, opt-in, behind
 * palu_softmax_pv_f16 (bits 16); covered by palu_pv_workspace_bytes for every L <= the capacity it was sized for */
int palu_pv_direct_nsplit(int G, int L, int Rv, int bits);
size_t palu_pv_workspace_bytes(int H, int G, int L, int Rv);
/* Byte offset, inside the workspace, of the per-head softmax statistics the call leaves behind:
 * float stats[H][2] = (max_l x[h,l], sum_l exp(x[h,l] - max)).  With ctx they are what a split-L
 * (multi-GPU or chunked) caller needs to LSE-merge partial results. */
size_t palu_pv_stats_offset(int H, int G, int L, int Rv);
int palu_softmax_pv_f16(const void* scores, int64_t ss_h, const void* mask,
                        const void* v, int64_t sv_g, int64_t sv_l,
                        void* ctx, void* probs, int64_t sp_h, void* workspace,
                        int H, int G, int L, int Rv, float sqrt_d, palu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused attention core of one decode step: abx scores -> /sqrt(D) -> softmax -> latent P.V in ONE kernel
 * (replaces kernel/palu_attention.py:219 recompute_k_gemv(...)/sqrt(D), :238 softmax, :246-251 latent P.V;
 * the [H, L] score tensor never exists in memory).  Same operands as palu_abx_rope_f16 + palu_softmax_pv_f16:
 *   q [H, D] rotated query, bfrag from palu_abx_prepare_b, k [G, L, Rk] / v [G, L, Rv] fp16 latents (16-byte
 *   aligned rows), ctx [H, Rv] fp16, key row l at position pos0 + l, no attention weights (mask: see _mask_ below).
 * palu_decode_attn_supported() != 0 for the shapes the kernel covers (D = 128, gs in {3,4}, Rk in {64,128},
 * Rv in {128,192,256,384}); palu_decode_attn_preferred() != 0 where palu_decode_step_f16 / palu_decode_attend_f16
 * pick it over the two-kernel path (measured: G * L <= ~300k rows, i.e. the head-group shards of a multi-GPU run and
 * short caches; PALU_FUSED_ATTN=1 / 0 in the environment forces it on for every covered shape / off).  workspace:
 * palu_pv_workspace_bytes(H, G, L, Rv) bytes; the per-head (max, sum) statistics are left at
 * palu_decode_attn_stats_offset() like palu_softmax_pv_f16 leaves them at palu_pv_stats_offset().
 */
int palu_decode_attn_supported(int H, int G, int Rk, int Rv, int D);
int palu_decode_attn_preferred(int H, int G, int L, int Rk, int Rv, int D);
int palu_decode_attn_nsplit(int G, int L);
size_t palu_decode_attn_stats_offset(int H, int G, int L, int Rv);
int palu_decode_attn_f16(const void* q, int64_t sq_h, int64_t sq_d, const void* bfrag,
                         const void* k, int64_t sk_g, int64_t sk_l, const void* v, int64_t sv_g, int64_t sv_l,
                         void* ctx, void* workspace, int H, int G, int L, int Rk, int Rv, int D,
                         const float* inv_freq, int pos0, float sqrt_d, palu_stream_t stream);
/* The same with an additive attention mask [L] fp16 (kernel/palu_attention.py:229-234: added to the fp16 logits
 * before the softmax; 0 = none): what palu_decode_step_f16 / palu_decode_attend_f16 call, so that a masked step
 * (e.g. a left-padded prompt) stays on the single-kernel core where that one is selected. */
int palu_decode_attn_mask_f16(const void* q, int64_t sq_h, int64_t sq_d, const void* bfrag,
                              const void* k, int64_t sk_g, int64_t sk_l, const void* v, int64_t sv_g, int64_t sv_l,
                              const void* mask, void* ctx, void* workspace, int H, int G, int L, int Rk, int Rv, int D,
                              const float* inv_freq, int pos0, float sqrt_d, palu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Batch-1 projections.
 * palu_gemv_f16: y[N] = W[N,K] x[K], fp16 in/out, fp32 accumulate (nn.Linear without bias:
 *   o_proj at kernel/palu_attention.py:257).  K % 8 == 0, K <= 32768.
 * palu_decode_qkv_f16: the three input projections of one token (palu_attention.py:164-168),
 *   RoPE of q at `pos` (:214-215; angle = fl32(pos*inv_freq)), and the append of the new latent
 *   rows into row `row` of the pre-allocated caches k_cache [G,Lmax,Rk], v_cache [G,Lmax,Rv]
 *   (replaces DynamicCache.update's torch.cat at :193).  q_out: [H*D] fp16.
 */
int palu_gemv_f16(const void* W, int64_t ldw, const void* x, void* y, int N, int K, palu_stream_t stream);
/* y = W x + bias, bias [N] fp16 added to the fp32 accumulator before the rounding (or NULL): o_proj of a model built with
 * config.attention_bias (kernel/palu_attention.py:145). */
int palu_gemv_bias_f16(const void* W, int64_t ldw, const void* x, const void* bias, void* y, int N, int K, palu_stream_t stream);
/* Whole-model decode (outside the attention module): y[n] = silu(Wg[n] . x) * (Wu[n] . x), the gate and up
 * projections of a gated MLP (transformers LlamaMLP: act_fn(gate_proj(x)) * up_proj(x)) for ONE token in one pass; fp16
 * roundings where the torch composition has them.  Wg, Wu: [N, K] fp16 (ldg, ldu), K % 8 == 0, K <= 32768. */
int palu_gemv_silu_mul_f16(const void* Wg, int64_t ldg, const void* Wu, int64_t ldu, const void* x, void* y,
                           int N, int K, palu_stream_t stream);
/* RMSNorm of one token (transformers LlamaRMSNorm: fp32 normalisation, fp16 rounding, times the fp16 weight), one launch. */
int palu_rmsnorm_row_f16(const void* x, const void* w, void* y, int K, float eps, palu_stream_t stream);
/* Same product, fp32 accumulators written out unrounded (y: [N] fp32): the per-rank partial of a column-sharded o_proj
 * (kernel/palu_attention.py:254-257): W = this rank's [hidden, H/N*Rv] column block (ldw = H*Rv),
 * x = its context slice; the ranks all-reduce the partials and round to fp16 once. */
int palu_gemv_f16_acc32(const void* W, int64_t ldw, const void* x, float* y, int N, int K, palu_stream_t stream);
int palu_decode_qkv_f16(const void* wq, int64_t ldq, const void* vtk, int64_t ldk, const void* vtv, int64_t ldv,
                        const void* x, void* q_out,
                        void* k_cache, int64_t sk_g, int64_t sk_l, void* v_cache, int64_t sv_g, int64_t sv_l,
                        const float* inv_freq, int H, int D, int hidden, int G, int Rk, int Rv,
                        int pos, int row, palu_stream_t stream);
/* The same with q_proj.bias ([H*D] fp16 or NULL), added before the rotation.  VT has no bias (kernel/palu_attention.py:33)
 * and the decode branch never applies the biases of U (:207-219), so with config.attention_bias q and o_proj are the two
 * biased products of a decode step (palu_gemv_bias_f16 is the other). */
int palu_decode_qkv_bias_f16(const void* wq, int64_t ldq, const void* q_bias, const void* vtk, int64_t ldk,
                             const void* vtv, int64_t ldv, const void* x, void* q_out, void* k_cache,
                             int64_t sk_g, int64_t sk_l, void* v_cache, int64_t sv_g, int64_t sv_l,
                             const float* inv_freq, int H, int D, int hidden, int G, int Rk, int Rv, int pos,
                             int row, palu_stream_t stream);

/* The same launch with the query fold of the position-split score kernel in the tail of its q waves: the wave that has just
 * produced the rotated (q_i, q_{i+64}) of head h folds them into its 2 x R fragment values of B (bfrag = palu_abx_prepare_b's
 * fragments) and writes them to `qfold` (palu_abx_fold_bytes(H, G, Rk) bytes), which palu_abx_rope_pf_f16 consumes later on the
 * same stream.  bfrag = qfold = NULL: plain palu_decode_qkv_bias_f16. */
int palu_decode_qkv_fold_f16(const void* wq, int64_t ldq, const void* q_bias, const void* vtk, int64_t ldk,
                             const void* vtv, int64_t ldv, const void* x, void* q_out, void* k_cache,
                             int64_t sk_g, int64_t sk_l, void* v_cache, int64_t sv_g, int64_t sv_l,
                             const float* inv_freq, int H, int D, int hidden, int G, int Rk, int Rv, int pos,
                             int row, const void* bfrag, void* qfold, palu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Whole decode step (5 launches on `stream`): qkv+RoPE+append -> abx -> softmax.PV -> o_proj.
 * Drop-in for the decode branch of LlamaPaluAttention.forward (kernel/palu_attention.py:207-257),
 * batch 1.  hidden: [hidden_size] fp16; caches hold `cache_len` valid rows on entry and
 * cache_len+1 on return; mask: [cache_len+1] fp16 additive or NULL; out: [hidden_size] fp16;
 * probs: [H, cache_len+1] fp16 or NULL.  workspace: palu_decode_workspace_bytes(H,G,D,Lcap,Rv)
 * bytes, valid for any cache_len < Lcap.  wo: [hidden_size, H*Rv] (U_v already folded in).
 */
size_t palu_decode_workspace_bytes(int H, int G, int D, int Lcap, int Rv);
int palu_decode_step_f16(const void* hidden,
                         const void* wq, int64_t ldq, const void* vtk, int64_t ldk, const void* vtv, int64_t ldv,
                         const void* bfrag, const void* wo, int64_t ldo,
                         void* k_cache, int64_t sk_g, int64_t sk_l, void* v_cache, int64_t sv_g, int64_t sv_l,
                         const void* mask, const float* inv_freq, void* out, void* probs, int64_t sp_h,
                         void* workspace, int Lcap, int H, int G, int D, int hidden_size, int Rk, int Rv,
                         int cache_len, int pos, palu_stream_t stream);

/* The same step for checkpoints whose heads share B inside a latent group (true GQA; the reference's
 * svd_mistral modules, palu/model/svd_mistral/modeling_palu_mistral.py:37-59): bfrag = fragments of the [G, R, D]
 * shared factor, palu_abx_prepare_b(b_g, H := G, G); scores by palu_abx_rope_shared_f16.  Same arguments otherwise. */
int palu_decode_step_sharedb_f16(const void* hidden,
                         const void* wq, int64_t ldq, const void* vtk, int64_t ldk, const void* vtv, int64_t ldv,
                         const void* bfrag, const void* wo, int64_t ldo,
                         void* k_cache, int64_t sk_g, int64_t sk_l, void* v_cache, int64_t sv_g, int64_t sv_l,
                         const void* mask, const float* inv_freq, void* out, void* probs, int64_t sp_h,
                         void* workspace, int Lcap, int H, int G, int D, int hidden_size, int Rk, int Rv,
                         int cache_len, int pos, palu_stream_t stream);
/* The same step WITHOUT the final o_proj: what one rank of the head-group sharding runs on the heads it owns
 * (H, G = local counts); ctx [H, Rv] fp16 is the slice it contributes to the all-gather.
 * Workspace as palu_decode_step_f16. */
int palu_decode_attend_f16(const void* hidden,
                           const void* wq, int64_t ldq, const void* vtk, int64_t ldk, const void* vtv, int64_t ldv,
                           const void* bfrag,
                           void* k_cache, int64_t sk_g, int64_t sk_l, void* v_cache, int64_t sv_g, int64_t sv_l,
                           const void* mask, const float* inv_freq, void* ctx,
                           void* workspace, int Lcap, int H, int G, int D, int hidden_size, int Rk, int Rv,
                           int cache_len, int pos, palu_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * 3/4-bit latent quantisation (palu/model/modules/quant.py:5-41, reference defaults: asymmetric,
 * one (scale, zero) per (token, head-group) row, clip 1.0 -- utils.py:103-108, svd_linear.py:124-139).
 * The reference only fake-quantises; the packed layout is this build's:
 *   codes [G, rows, R*bits/8] bytes: little-endian bit stream, code j at bits [j*bits,(j+1)*bits)
 *   meta  [G, rows, 2] fp16: (scale, zero)
 * Codes and dequantised values are bit-exact with quantize_tensor's fp16 arithmetic.
 * x / dequant / out: [G, rows, R] fp16 with the given element strides; dequant may be NULL.
 * bits = 4 needs R % 8 == 0, bits = 3 needs R % 32 == 0.  Byte strides for codes.
 */
size_t palu_packed_row_bytes(int R, int bits);
int palu_quantize_pack(const void* x, int64_t sx_g, int64_t sx_l,
                       void* codes, int64_t sc_g, int64_t sc_l, void* meta, int64_t sm_g, int64_t sm_l,
                       void* dequant, int64_t sd_g, int64_t sd_l,
                       int G, int nrows, int R, int bits, palu_stream_t stream);
/* The other modes of quantize_tensor (quant.py:18-36; flags lt_sym / lt_clip_ratio of utils.py:101-109):
 * sym != 0: scale = clamp(amax|w|, 1e-5)[* clip] / (2^(b-1) - 1), signed codes; stored in offset binary
 * (code + 2^(b-1), zero = 2^(b-1)) so that (stored - zero) * scale is the reference's value bit for bit and every decode
 * kernel works unchanged.  clip_ratio in (0, 1]: max / min (asym) or amax (sym) are scaled before the grid is built. */
int palu_quantize_pack_ex(const void* x, int64_t sx_g, int64_t sx_l,
                          void* codes, int64_t sc_g, int64_t sc_l, void* meta, int64_t sm_g, int64_t sm_l,
                          void* dequant, int64_t sd_g, int64_t sd_l,
                          int G, int nrows, int R, int bits, int sym, float clip_ratio, palu_stream_t stream);
int palu_unpack_dequant(const void* codes, int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l,
                        void* out, int64_t so_g, int64_t so_l, int G, int nrows, int R, int bits,
                        palu_stream_t stream);
/* raw integer pack/unpack of uint8 codes (ncodes % 8 == 0), bit-exact inverse pair */
int palu_pack_codes(const void* codes_u8, void* packed, int64_t ncodes, int bits, palu_stream_t stream);
int palu_unpack_codes(const void* packed, void* codes_u8, int64_t ncodes, int bits, palu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Hadamard: y = (x . H_n) * scale over the last dim of a [rows, n] array, Sylvester order, n = 2^m.
 * Replaces fast_hadamard_transform.hadamard_transform (external CUDA op; call sites
 * palu/model/modules/hadamard_utils.py:141,145,177).  dtype: 0 = fp16, 1 = fp32.  In place allowed.
 */
int palu_hadamard_transform(const void* x, void* y, int64_t rows, int n, float scale, int dtype,
                            palu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Quantised-latent variants (3/4-bit codes + per-row (scale, zero), layout above).  The reference has
 * no such kernels (README.md:24 TODO); semantics = the fp16 entry points applied to the fake-quantised
 * latents quantize_tensor(x) (svd_linear.py:84-90,124-139), i.e. dequantised values bit-identical.
 * abx_q: 3 bit with R % 32 == 0, 4 bit with R % 8 == 0 -- (4,32), (4,64), (4,128), (3,128) on the fast kernels, every
 * other rank (the 96 / 160 / 224 / 256 of palu/rank_search.py:11-17 ...) on the chunked one; softmax_pv_q needs
 * Rv % 32 == 0, gs in {1,2,3,4,8}.  Byte strides for codes, element strides for meta.
 * The *_qg entry points take rows quantised in column groups (quantize_tensor(..., group_size > 0), quant.py:11-13, the
 * --lt_group_size option of utils.py:105): meta [G, L, R / group_size, 2] (sm_l >= 2 R / group_size); the packed codes
 * are laid out exactly as for whole-row quantisation (group_size % 8 == 0; % 32 for P.V and the step); group_size = 0
 * or = R is the whole-row form.
 */
int palu_abx_rope_q(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag,
                    const void* codes, int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l,
                    void* out, int64_t so_h, int H, int G, int L, int R, int D, int bits,
                    const float* inv_freq, int pos0, palu_stream_t stream);
int palu_softmax_pv_q(const void* scores, int64_t ss_h, const void* mask,
                      const void* codes, int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l,
                      void* ctx, void* probs, int64_t sp_h, void* workspace,
                      int H, int G, int L, int Rv, int bits, float sqrt_d, palu_stream_t stream);
int palu_abx_rope_qg(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag,
                     const void* codes, int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l,
                     void* out, int64_t so_h, int H, int G, int L, int R, int D, int bits, int group_size,
                     const float* inv_freq, int pos0, void* scratch /* palu_abx_scratch_bytes() or 0 */,
                     palu_stream_t stream);
int palu_softmax_pv_qg(const void* scores, int64_t ss_h, const void* mask,
                       const void* codes, int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l,
                       void* ctx, void* probs, int64_t sp_h, void* workspace,
                       int H, int G, int L, int Rv, int bits, int group_size, float sqrt_d, palu_stream_t stream);
int palu_decode_step_qg(const void* hidden,
                        const void* wq, int64_t ldq, const void* vtk, int64_t ldk, const void* vtv, int64_t ldv,
                        const void* bfrag, const void* wo, int64_t ldo,
                        void* k_codes, int64_t skc_g, int64_t skc_l, void* k_meta, int64_t skm_g, int64_t skm_l,
                        void* v_codes, int64_t svc_g, int64_t svc_l, void* v_meta, int64_t svm_g, int64_t svm_l,
                        const void* mask, const float* inv_freq, void* out, void* probs, int64_t sp_h,
                        void* workspace, int Lcap, int H, int G, int D, int hidden_size, int Rk, int Rv,
                        int bits, int group_size, int cache_len, int pos, palu_stream_t stream);
int palu_decode_step_q(const void* hidden,
                       const void* wq, int64_t ldq, const void* vtk, int64_t ldk, const void* vtv, int64_t ldv,
                       const void* bfrag, const void* wo, int64_t ldo,
                       void* k_codes, int64_t skc_g, int64_t skc_l, void* k_meta, int64_t skm_g, int64_t skm_l,
                       void* v_codes, int64_t svc_g, int64_t svc_l, void* v_meta, int64_t svm_g, int64_t svm_l,
                       const void* mask, const float* inv_freq, void* out, void* probs, int64_t sp_h,
                       void* workspace, int Lcap, int H, int G, int D, int hidden_size, int Rk, int Rv,
                       int bits, int cache_len, int pos, palu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Prefill down-projection (MFMA GEMM): latents = X . VT^T for a whole prompt
 * (HeadwiseLowRankModule.project_to_latent with q_len = L: kernel/palu_attention.py:59-65,167-168),
 * written straight into the latent-cache layout:
 *   out[(n / R) * so_g + (row0 + m) * so_l + n % R] = sum_k x[m, k] * w[n, k]      m < M, n < N
 * x: [M, K] fp16 (ldx), w = VT: [N, K] fp16 (ldw), K % 64 == 0, N % R == 0, fp32 accumulate, fp16 out.
 */
int palu_lowrank_project_gemm(const void* x, int64_t ldx, const void* w, int64_t ldw,
                              void* out, int64_t so_g, int64_t so_l,
                              int M, int N, int K, int R, int row0, palu_stream_t stream);
/* The same with a fused quantise + pack epilogue (a packed-cache prompt pass never materialises fp16
 * latents): the tile's (token, group) rows are quantised as quantize_tensor does with the reference defaults (asymmetric,
 * group_size 0, clip_ratio 1: palu/model/modules/quant.py:29-39; svd_linear.py:124-139) and written as packed rows
 *   codes[(n / R) * sc_g + (row0 + m) * sc_l + ...]  (bytes; a row = R * bits / 8 bytes, the layout of palu_quantize_pack)
 *   meta [(n / R) * sm_g + (row0 + m) * sm_l + {0, 1}] = (scale, zero) fp16 (elements)
 * bit-identical to palu_lowrank_project_gemm followed by palu_quantize_pack.  bits = 3 / 4.  The fused tile needs whole groups
 * per workgroup (32 NI | R, R | 128 NI, N % (128 NI) == 0 for an NI in {1, 2, 3}) and M >= 512, N >= 256, K >= 512:
 * palu_lowrank_project_gemm_q_supported says whether a shape qualifies; otherwise PALU_ERR_UNSUPPORTED (run the two calls). */
int palu_lowrank_project_gemm_q(const void* x, int64_t ldx, const void* w, int64_t ldw,
                                void* codes, int64_t sc_g, int64_t sc_l, void* meta, int64_t sm_g, int64_t sm_l,
                                int M, int N, int K, int R, int row0, int bits, palu_stream_t stream);
int palu_lowrank_project_gemm_q_supported(int M, int N, int K, int R, int bits);

/* ------------------------------------------------------------------------------------------
 * In-place rotary embedding of x [H, T, D] fp16 (strides sx_h, sx_t; D contiguous), row t at position
 * pos0 + t, with the rounding points of the reference's prompt branch (HF 4.37.2 rotary_emb +
 * apply_rotary_pos_emb as called at kernel/palu_attention.py:204-205): fp32 cos/sin table cast to fp16,
 * x*cos + rotate_half(x)*sin evaluated in fp16.  inv_freq: D/2 fp32 values (palu_rope_inv_freq_host).
 */
int palu_rope_f16(void* x, int64_t sx_h, int64_t sx_t, int H, int T, int D, int pos0, const float* inv_freq,
                  palu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Prefill attention over the latent value cache: the q_len > 1 branch of
 * LlamaPaluAttention.forward (kernel/palu_attention.py:196-206 scores, :229-238 mask + softmax,
 * :246-255 latent P.V and head concat), flash-style -- the [Tq x Tk] score matrix of :205 is never
 * materialised:
 *   out[t, h*Rv + c] = sum_j softmax_j(q[h,t,:].k[h,j,:] * scale, j <= past + t if causal) * V_lat[h/gs, j, c]
 * q:  [H, Tq, D] RoPE'd queries (element strides sq_h, sq_t; D contiguous), row t = position past + t
 * k:  [H, Tk, D] reconstructed + RoPE'd keys (sk_h, sk_t)
 * vt: [G, Rv, Tk] latent values TRANSPOSED (kv contiguous; strides sv_g, sv_c), each row zero-padded
 *     to a multiple of 64 positions (sv_c >= round_up(Tk, 64)) -- a transient prefill workspace
 * out:[Tq, H*Rv] fp16 (row stride so_t) = the operand of the fused o_proj (:257)
 * causal != 0: the standard causal mask of the HF caller (additive -inf above the diagonal);
 * causal == 0: no mask (what :229 does when attention_mask is None).  fp32 online softmax.
 * D must be 128, Rv % 32 == 0.
 */
int palu_prefill_attn_f16(const void* q, int64_t sq_h, int64_t sq_t, const void* k, int64_t sk_h, int64_t sk_t,
                          const void* vt, int64_t sv_g, int64_t sv_c, void* out, int64_t so_t, int H, int G,
                          int D, int Tq, int Tk, int Rv, int past, int causal, float scale,
                          palu_stream_t stream);
/* The same over kv PANELS, for prompt passes whose reconstructed keys / transposed values must not exist all at once (the
 * reference materialises the full K and the [H, T, T] scores, palu_attention.py:199-205): k / vt hold the kv positions
 * [kv0, kv0 + Tk) only; past_rel = (absolute position of query row 0) - kv0, negative when the panel starts behind the first
 * query.  The online-softmax state of every (head, query) travels between the launches in fp32: state_o [H][Tq][Rv]
 * (un-normalised context), state_ml [Rv / 32][H][Tq][8] (running maximum + partial sums, one slice per column block a launch
 * may split a query tile into); palu_prefill_state_bytes(H, Tq, Rv, 0 / 1)
 * gives their sizes.  first != 0: start from the empty state (the buffers need no initialisation); last != 0: normalise and
 * write `out` instead of the state.  The panels of one (query chunk, head set) run in ascending kv order on one stream. */
int palu_prefill_attn_panel_f16(const void* q, int64_t sq_h, int64_t sq_t, const void* k, int64_t sk_h, int64_t sk_t,
                                const void* vt, int64_t sv_g, int64_t sv_c, void* out, int64_t so_t, int H, int G,
                                int D, int Tq, int Tk, int Rv, int past_rel, int causal, float scale,
                                void* state_o, void* state_ml, int first, int last, palu_stream_t stream);
size_t palu_prefill_state_bytes(int H, int Tq, int Rv, int which);
/* Prompt attention straight from the latent caches (round 6, csrc/prefill_lat.hip): the keys of every 64-position
 * kv tile are rebuilt INSIDE the kernel, K~ = RoPE(X_k . B_h) (kernel/palu_attention.py:67-77, :199-205: reconstruct GEMM rounded to
 * fp16, HF rotary with fp16 cos / sin and fp16 products), and the latent values are read in the cache's row-major layout -- no
 * [H, kv, D] key workspace and no transposed value copy.  q [H][Tq][128] rotated queries (first one at absolute position `past`),
 * xk [G][>= Tk][128] / xv [G][>= Tk][Rv] fp16 latent caches (row l = position l), bt = B^T [H][128 d][Rk] contiguous (the rows
 * of U_h), cs = the rotary cache of the key positions 0 .. Tk - 1, [pos][2][64] fp16 (cos row, sin row), built once by
 * palu_rope_cs_table_build (palu_rope_cs_table_bytes(npos) bytes); out [Tq][H * Rv] fp16.  Needs head_dim 128, rank_k / G in {64, 128} with
 * rank_v / G in {128, 192, 256, 384}, or 32 with 64 / 96 (palu_prefill_attn_lat_supported); causal as palu_prefill_attn_f16. */
size_t palu_rope_cs_table_bytes(int npos);
int palu_rope_cs_table_build(const float* inv_freq, int pos0, int npos, void* table, palu_stream_t stream);
int palu_prefill_attn_lat_supported(int H, int G, int D, int Rk, int Rv);
/* the same per row format: bits = 16 (fp16 rows, palu_prefill_attn_lat_f16), 4 or 3 (packed rows, palu_prefill_attn_lat_q) */
int palu_prefill_attn_lat_supported_bits(int H, int G, int D, int Rk, int Rv, int bits);
int palu_prefill_attn_lat_f16(const void* q, int64_t sq_h, int64_t sq_t, const void* xk, int64_t sxk_g, int64_t sxk_l,
                              const void* xv, int64_t sxv_g, int64_t sxv_l, const void* bt, const void* cs, void* out,
                              int64_t so_t, int H, int G, int D, int Tq, int Tk, int Rk, int Rv, int past, int causal,
                              float scale, palu_stream_t stream);
/* The same over PACKED 4-bit or 3-bit latent caches (the reference's README.md:24 TODO; values defined by
 * palu/model/modules/quant.py:37-39): codes [G][>= Tk][R bits / 8] bytes (byte strides s*c_g, s*c_l), meta [G][>= Tk][2] fp16 (scale, zero) per (token, group) row (element
 * strides) -- palu_quantize_pack's layout.  The codes are de-quantised inside the kernel with palu_unpack_dequant's arithmetic (keys
 * in the rebuild's registers, values into the kernel's half-tile image): no fp16 copy of the cache exists.  bt_perm = B^T
 * [H][128][Rk] with the columns of every group of 8 in the order 0 4 1 5 2 6 3 7 (the order the nibbles leave a dword; the 3-bit form extracts its pairs in the same
 * order).  bits = 4: rank_v / G a multiple of 64; bits = 3: rank_k / G = 128 and rank_v / G in {128, 256, 384}. */
int palu_prefill_attn_lat_q(const void* q, int64_t sq_h, int64_t sq_t, const void* k_codes, int64_t skc_g, int64_t skc_l,
                            const void* k_meta, int64_t skm_g, int64_t skm_l, const void* v_codes, int64_t svc_g,
                            int64_t svc_l, const void* v_meta, int64_t svm_g, int64_t svm_l, const void* bt_perm,
                            const void* cs, void* out, int64_t so_t, int H, int G, int D, int Tq, int Tk, int Rk, int Rv,
                            int bits, int past, int causal, float scale, palu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One-shot peer-to-peer exchange for the head-group-parallel decode step (the reference is single-GPU and
 * has no collective).  The step's one collective moves 3 KiB (all-gather of a rank's context slice) or 16 KiB (all-reduce
 * of the [hidden] fp32 partials) per rank: one kernel per rank writes its slice straight into every peer's exchange
 * buffer, raises a flag there, waits for the peers' flags in its own buffer (bounded spin) and copies / sums the slots.
 * Setup (allocates, synchronises -- NOT part of a step): every rank allocates a buffer of palu_exchange_bytes(nranks,
 * slot_bytes), exports a handle (palu_exchange_handle_bytes() host bytes) that the other ranks import (hipIpc*; within one
 * process the pointer itself is used), and uploads the n base pointers, own included and in rank order, as a device array.
 * Steps (capturable, no allocation, no host sync): palu_exchange_allgather (out [nranks][bytes]) /
 * palu_exchange_allreduce_f32 (out [bytes], the fp32 sum in rank order: identical on every rank).  Every rank must issue the
 * same sequence of exchanges on a buffer.  palu_exchange_status reads the buffer's control words after a synchronisation:
 * exchanges done and the epoch of a wait that timed out (0 = none; a peer that never arrives becomes an error, not a hang).
 */
size_t palu_exchange_bytes(int nranks, size_t slot_bytes);
int palu_exchange_alloc(size_t bytes, void** ptr);
int palu_exchange_free(void* ptr);
size_t palu_exchange_handle_bytes(void);
int palu_exchange_export(void* ptr, void* handle_out_host);
int palu_exchange_import(const void* handle_host, void** ptr_out);
int palu_exchange_close(void* imported_ptr);
int palu_exchange_allgather(const void* src, size_t bytes, const void* peers_dev, int rank, int nranks, size_t slot_bytes,
                            void* out, palu_stream_t stream);
int palu_exchange_allreduce_f32(const void* src, size_t bytes, const void* peers_dev, int rank, int nranks,
                                size_t slot_bytes, void* out, palu_stream_t stream);
int palu_exchange_status(const void* buffer, unsigned* epoch_host, unsigned* error_host);

#ifdef __cplusplus
}
#endif
#endif /* PALU_HIP_H */
