// Batch-1 projections of the decode step (HBM-bound GEMVs), one wave per PAIR of output rows.
//
//   decode_qkv : q = W_q h, k_lat = VT_k h, v_lat = VT_v h   (kernel/palu_attention.py:164-168)
//                + RoPE of q at the token position (:214-215) + in-place append of the new latent
//                rows to the pre-allocated caches (replaces DynamicCache.update's torch.cat, :193)
//   gemv       : y = W x  (o_proj with U_v folded in, :257)
//
// Weights are streamed once with 16-byte loads straight into registers (no LDS round trip: nothing
// is shared between waves), x lives in LDS, products use v_dot2_f32_f16 with fp32 accumulation,
// one wave-level shuffle reduction per row pair.  For q the two rows of a wave are (i, i+64) of one
// head so that the rotation is applied in registers before the single fp16 rounding.
#include "abx_fold.h"

namespace {

constexpr int GV_THREADS = 256;   // 4 waves -> 8 output rows per workgroup

static __device__ __forceinline__ float dot8(u32x4 w, u32x4 x, float acc) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    // copy the lanes to scalars first: __builtin_bit_cast straight from a vector-element lvalue
    // miscompiles (always element 0) with this hipcc
    const unsigned we = w[e], xe = x[e];
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, we), __builtin_bit_cast(h16x2, xe), acc, false);
  }
  return acc;
}

// dot products of rows r0, r1 (length K, K % 8 == 0) with the LDS-resident vector xs
static __device__ __forceinline__ void row_pair_dot(const h16* __restrict__ r0, const h16* __restrict__ r1,
                                                    const h16* xs, int K, int lane, float* y0, float* y1) {
  float a0 = 0.f, a1 = 0.f;
  const int nchunk = K >> 3;
  int c = lane;
  for (; c + 192 < nchunk; c += 256) {   // 4 chunks per row in flight
    u32x4 w0[4], w1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      w0[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(r0) + c + 64 * u);
      w1[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(r1) + c + 64 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      u32x4 xv = *(reinterpret_cast<const u32x4*>(xs) + c + 64 * u);
      a0 = dot8(w0[u], xv, a0);
      a1 = dot8(w1[u], xv, a1);
    }
  }
  for (; c < nchunk; c += 64) {
    u32x4 xv = *(reinterpret_cast<const u32x4*>(xs) + c);
    a0 = dot8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(r0) + c), xv, a0);
    a1 = dot8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(r1) + c), xv, a1);
  }
  *y0 = wave_sum(a0);
  *y1 = wave_sum(a1);
}

static __device__ __forceinline__ void stage_x(const h16* __restrict__ x, h16* xs, int K, int tid) {
  for (int c = tid; c < (K >> 3); c += GV_THREADS)
    *(reinterpret_cast<u32x4*>(xs) + c) = *(reinterpret_cast<const u32x4*>(x) + c);
  __syncthreads();
}

// bias (optional, nn.Linear(bias=True): kernel/palu_attention.py:142-145 with config.attention_bias): added to the fp32
// accumulator before the one rounding, as a GEMM epilogue would
template <typename OUT>
__global__ __launch_bounds__(GV_THREADS) void gemv_kernel(const h16* __restrict__ W, int64_t ldw,
                                                          const h16* __restrict__ x, OUT* __restrict__ y, int N,
                                                          int K, const h16* __restrict__ bias) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  h16* xs = reinterpret_cast<h16*>(smem_raw);
  stage_x(x, xs, K, threadIdx.x);
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * (GV_THREADS / 64) + (threadIdx.x >> 6);
  const int n0 = 2 * pair;
  if (n0 >= N) return;
  const int n1 = min(n0 + 1, N - 1);
  float y0, y1;
  row_pair_dot(W + (int64_t)n0 * ldw, W + (int64_t)n1 * ldw, xs, K, lane, &y0, &y1);
  if (lane == 0) {
    if (bias) {
      y0 += (float)bias[n0];
      y1 += (float)bias[n1];
    }
    y[n0] = (OUT)y0;
    if (n0 + 1 < N) y[n0 + 1] = (OUT)y1;
  }
}

// act[n] = silu(W_gate[n] . x) * (W_up[n] . x): the two up-projections of a gated MLP for one token in one pass over x
// (whole-model decode, SURVEY 8(f) N2 -- not part of the attention module).  One wave per output element: its gate row and
// its up row.  fp16 semantics of the torch composition: both dot products are rounded to fp16 (two nn.Linear outputs), silu
// is evaluated in fp32 on the rounded gate and rounded, the product is rounded.
__global__ __launch_bounds__(GV_THREADS) void gemv_silu_mul_kernel(const h16* __restrict__ Wg, int64_t ldg,
                                                                   const h16* __restrict__ Wu, int64_t ldu,
                                                                   const h16* __restrict__ x, h16* __restrict__ y, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  h16* xs = reinterpret_cast<h16*>(smem_raw);
  stage_x(x, xs, K, threadIdx.x);
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (GV_THREADS / 64) + (threadIdx.x >> 6);
  if (n >= N) return;
  float g, u;
  row_pair_dot(Wg + (int64_t)n * ldg, Wu + (int64_t)n * ldu, xs, K, lane, &g, &u);
  if (lane == 0) {
    const float g16 = (float)(h16)g, u16 = (float)(h16)u;
    const float s16 = (float)(h16)(g16 / (1.0f + __expf(-g16)));
    y[n] = (h16)(s16 * u16);
  }
}

// RMSNorm of ONE token (transformers LlamaRMSNorm: x * rsqrt(mean(x^2) + eps) in fp32, rounded to fp16, times the fp16
// weight) in one launch instead of the seven of the torch composition -- whole-model decode.
__global__ __launch_bounds__(GV_THREADS) void rmsnorm_row_kernel(const h16* __restrict__ x, const h16* __restrict__ w,
                                                                 h16* __restrict__ y, int K, float eps) {
  __shared__ float part[GV_THREADS / 64];
  const int tid = threadIdx.x;
  float ss = 0.f;
  for (int c = tid; c < (K >> 3); c += GV_THREADS) {
    const h16x8 v = *(reinterpret_cast<const h16x8*>(x) + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf((float)v[e], (float)v[e], ss);
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) part[tid >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < GV_THREADS / 64; ++i) tot += part[i];
  const float rs = rsqrtf(tot / (float)K + eps);
  for (int c = tid; c < (K >> 3); c += GV_THREADS) {
    const h16x8 v = *(reinterpret_cast<const h16x8*>(x) + c);
    const h16x8 g = *(reinterpret_cast<const h16x8*>(w) + c);
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (h16)((float)g[e] * (float)(h16)((float)v[e] * rs));
    *(reinterpret_cast<h16x8*>(y) + c) = o;
  }
}

struct QkvParams {
  const h16 *wq, *vtk, *vtv, *x;
  int64_t ldq, ldk, ldv;
  h16* q_out;                 // [H*D] rotated query
  h16 *k_cache, *v_cache;     // [G, Lmax, R*] caches, row `row` receives the new latent
  int64_t sk_g, sk_l, sv_g, sv_l;
  const float* inv_freq;
  int H, D, K, rank_k, rank_v, Rk, Rv;
  int pos, row;
  const h16* q_bias;          // optional [H*D]: q_proj.bias, added before the rotation
  // optional tail of the q waves (abx_fold.h): fold the rotated (q_i, q_{i+64}) into the two-band fragments of B for the
  // position-split score kernel that follows in the step (H = 4 G, D = 128, Rk = 16 fold_nks)
  const u32x4* fold_b;        // two-band fragments of B (palu_abx_prepare_b's second part)
  u32x4* qfold;               // folded fragments out, [G][16 fold_nks KB]; null = no fold
  int fold_nks;
};

// cos/sin of the oracle's fp32-rounded angle fl32(pos * f)  (kernel/pytorch_reference.py:5-6)
static __device__ __forceinline__ void rope_cs(int pos, float f, float* c, float* s) {
  const float lf = (float)pos;
  const float ang = lf * f;
  const double x = (double)ang;   // the rounded product IS the angle here: reduce it exactly in fp64
  const double nd = __builtin_rint(x * 0.6366197723675814);
  double rd = __builtin_fma(-nd, 1.5707963267948966, x);
  rd = __builtin_fma(-nd, 6.123233995736766e-17, rd);
  const float r = (float)rd, r2 = r * r;
  float sp = fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f);
  sp = fmaf(r * r2, sp, r);
  float cp = fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
  cp = fmaf(r2 * r2, cp, fmaf(-0.5f, r2, 1.0f));
  const int q = (int)(long long)nd & 3;
  const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}

__global__ __launch_bounds__(GV_THREADS) void decode_qkv_kernel(QkvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  h16* xs = reinterpret_cast<h16*>(smem_raw);
  stage_x(p.x, xs, p.K, threadIdx.x);
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * (GV_THREADS / 64) + (threadIdx.x >> 6);
  const int half = p.D / 2;
  const int nq = p.H * half, nk = p.rank_k / 2, nv = p.rank_v / 2;
  float y0, y1;
  if (pair < nq) {
    const int h = pair / half, i = pair - h * half;
    const h16* r0 = p.wq + (int64_t)(h * p.D + i) * p.ldq;
    // (the 512 B of B fragments this pair folds are requested in front of the weight stream: their L2 / HBM latency is
    //  gone by the time the dot products are reduced)
    AbxFoldSrc fsrc;
    if (p.qfold) fsrc = abx_fold_load(p.fold_b, p.H / 4, p.fold_nks, h >> 2, h & 3, i, lane);
    row_pair_dot(r0, r0 + (int64_t)half * p.ldq, xs, p.K, lane, &y0, &y1);
    float c, s;
    if (p.q_bias) {
      // HF adds the bias inside the fp16 linear (one rounding of W x + b) and rotates the fp16 result
      y0 += (float)p.q_bias[h * p.D + i];
      y1 += (float)p.q_bias[h * p.D + i + half];
    }
    rope_cs(p.pos, p.inv_freq[i], &c, &s);
    const h16 qa = (h16)(y0 * c - y1 * s), qb = (h16)(y1 * c + y0 * s);     // (every lane: wave_sum leaves the sums in all of them)
    if (lane == 0) {
      p.q_out[h * p.D + i] = qa;
      p.q_out[h * p.D + i + half] = qb;
    }
    if (p.qfold) abx_fold_store(fsrc, p.qfold, p.fold_nks, h >> 2, h & 3, i, qa, qb, lane);
  } else if (pair < nq + nk) {
    const int n0 = 2 * (pair - nq);
    const h16* r0 = p.vtk + (int64_t)n0 * p.ldk;
    row_pair_dot(r0, r0 + p.ldk, xs, p.K, lane, &y0, &y1);
    if (lane == 0) {
      const int g = n0 / p.Rk, r = n0 - g * p.Rk;
      h16* d = p.k_cache + (int64_t)g * p.sk_g + (int64_t)p.row * p.sk_l + r;
      *reinterpret_cast<h16x2*>(d) = h16x2{(h16)y0, (h16)y1};
    }
  } else if (pair < nq + nk + nv) {
    const int n0 = 2 * (pair - nq - nk);
    const h16* r0 = p.vtv + (int64_t)n0 * p.ldv;
    row_pair_dot(r0, r0 + p.ldv, xs, p.K, lane, &y0, &y1);
    if (lane == 0) {
      const int g = n0 / p.Rv, r = n0 - g * p.Rv;
      h16* d = p.v_cache + (int64_t)g * p.sv_g + (int64_t)p.row * p.sv_l + r;
      *reinterpret_cast<h16x2*>(d) = h16x2{(h16)y0, (h16)y1};
    }
  }
}

}  // namespace

extern "C" int palu_gemv_f16(const void* W, int64_t ldw, const void* x, void* y, int N, int K, palu_stream_t stream) {
  return palu_gemv_bias_f16(W, ldw, x, nullptr, y, N, K, stream);
}

// y = W x + bias (bias [N] fp16 or null): o_proj of a model with attention_bias (kernel/palu_attention.py:145)
extern "C" int palu_gemv_bias_f16(const void* W, int64_t ldw, const void* x, const void* bias, void* y, int N, int K,
                                  palu_stream_t stream) {
  PALU_REQUIRE(W && x && y && N > 0 && K > 0, PALU_ERR_ARG, "gemv: bad arguments");
  PALU_REQUIRE(K % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)x & 15) == 0, PALU_ERR_ARG,
               "gemv: K, ldw must be multiples of 8 and W, x 16-byte aligned");
  PALU_REQUIRE((size_t)K * 2 <= 64 * 1024, PALU_ERR_UNSUPPORTED, "gemv: K too large for the LDS-resident vector");
  const int pairs = (N + 1) / 2;
  const int blocks = (pairs + GV_THREADS / 64 - 1) / (GV_THREADS / 64);
  hipLaunchKernelGGL(gemv_kernel<h16>, dim3(blocks), dim3(GV_THREADS), (size_t)K * 2, (hipStream_t)stream, (const h16*)W,
                     ldw, (const h16*)x, (h16*)y, N, K, (const h16*)bias);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

extern "C" int palu_gemv_silu_mul_f16(const void* Wg, int64_t ldg, const void* Wu, int64_t ldu, const void* x, void* y, int N,
                                      int K, palu_stream_t stream) {
  PALU_REQUIRE(Wg && Wu && x && y && N > 0 && K > 0, PALU_ERR_ARG, "gemv_silu_mul: bad arguments");
  PALU_REQUIRE(K % 8 == 0 && ldg % 8 == 0 && ldu % 8 == 0 && (((uintptr_t)Wg | (uintptr_t)Wu | (uintptr_t)x) & 15) == 0,
               PALU_ERR_ARG, "gemv_silu_mul: K, ldg, ldu must be multiples of 8 and the operands 16-byte aligned");
  PALU_REQUIRE((size_t)K * 2 <= 64 * 1024, PALU_ERR_UNSUPPORTED, "gemv_silu_mul: K too large for the LDS-resident vector");
  const int blocks = (N + GV_THREADS / 64 - 1) / (GV_THREADS / 64);
  hipLaunchKernelGGL(gemv_silu_mul_kernel, dim3(blocks), dim3(GV_THREADS), (size_t)K * 2, (hipStream_t)stream, (const h16*)Wg,
                     ldg, (const h16*)Wu, ldu, (const h16*)x, (h16*)y, N, K);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

extern "C" int palu_rmsnorm_row_f16(const void* x, const void* w, void* y, int K, float eps, palu_stream_t stream) {
  PALU_REQUIRE(x && w && y && K > 0 && K % 8 == 0, PALU_ERR_ARG, "rmsnorm_row: K must be a positive multiple of 8");
  PALU_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0, PALU_ERR_ARG, "rmsnorm_row: 16-byte aligned operands");
  hipLaunchKernelGGL(rmsnorm_row_kernel, dim3(1), dim3(GV_THREADS), 0, (hipStream_t)stream, (const h16*)x, (const h16*)w, (h16*)y,
                     K, eps);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

// the same product with the fp32 accumulator written out unrounded: the partial sums of a column-sharded o_proj
// (each rank multiplies its own context slice by its column block; the ranks all-reduce [N] fp32 and round once)
extern "C" int palu_gemv_f16_acc32(const void* W, int64_t ldw, const void* x, float* y, int N, int K,
                                   palu_stream_t stream) {
  PALU_REQUIRE(W && x && y && N > 0 && K > 0, PALU_ERR_ARG, "gemv_acc32: bad arguments");
  PALU_REQUIRE(K % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)x & 15) == 0, PALU_ERR_ARG,
               "gemv_acc32: K, ldw must be multiples of 8 and W, x 16-byte aligned");
  PALU_REQUIRE((size_t)K * 2 <= 64 * 1024, PALU_ERR_UNSUPPORTED, "gemv_acc32: K too large for the LDS-resident vector");
  const int pairs = (N + 1) / 2;
  const int blocks = (pairs + GV_THREADS / 64 - 1) / (GV_THREADS / 64);
  hipLaunchKernelGGL(gemv_kernel<float>, dim3(blocks), dim3(GV_THREADS), (size_t)K * 2, (hipStream_t)stream,
                     (const h16*)W, ldw, (const h16*)x, y, N, K, (const h16*)nullptr);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

extern "C" int palu_decode_qkv_f16(const void* wq, int64_t ldq, const void* vtk, int64_t ldk, const void* vtv,
                                   int64_t ldv, const void* x, void* q_out, void* k_cache, int64_t sk_g, int64_t sk_l,
                                   void* v_cache, int64_t sv_g, int64_t sv_l, const float* inv_freq, int H, int D,
                                   int hidden, int G, int Rk, int Rv, int pos, int row, palu_stream_t stream) {
  return palu_decode_qkv_bias_f16(wq, ldq, nullptr, vtk, ldk, vtv, ldv, x, q_out, k_cache, sk_g, sk_l, v_cache, sv_g, sv_l,
                                  inv_freq, H, D, hidden, G, Rk, Rv, pos, row, stream);
}

// the same with q_proj.bias ([H*D] fp16 or null).  The latent projections VT have no bias (kernel/palu_attention.py:33),
// and the decode branch never applies the biases of U (:207-219 reads the latents only), so q and o_proj are the two
// places config.attention_bias reaches in a decode step.
extern "C" int palu_decode_qkv_bias_f16(const void* wq, int64_t ldq, const void* q_bias, const void* vtk, int64_t ldk,
                                        const void* vtv, int64_t ldv, const void* x, void* q_out, void* k_cache,
                                        int64_t sk_g, int64_t sk_l, void* v_cache, int64_t sv_g, int64_t sv_l,
                                        const float* inv_freq, int H, int D, int hidden, int G, int Rk, int Rv, int pos,
                                        int row, palu_stream_t stream) {
  return palu_decode_qkv_fold_f16(wq, ldq, q_bias, vtk, ldk, vtv, ldv, x, q_out, k_cache, sk_g, sk_l, v_cache, sv_g, sv_l,
                                  inv_freq, H, D, hidden, G, Rk, Rv, pos, row, nullptr, nullptr, stream);
}

// The same launch with the query fold of the position-split score kernel in the tail of its q waves (abx_fold.h): `bfrag`
// = palu_abx_prepare_b's fragments of B, `qfold` = palu_abx_fold_bytes(H, G, Rk) bytes that palu_abx_rope_pf_f16 consumes
// later in the same stream.  bfrag = qfold = null: no fold (palu_decode_qkv_bias_f16).
extern "C" int palu_decode_qkv_fold_f16(const void* wq, int64_t ldq, const void* q_bias, const void* vtk, int64_t ldk,
                                        const void* vtv, int64_t ldv, const void* x, void* q_out, void* k_cache,
                                        int64_t sk_g, int64_t sk_l, void* v_cache, int64_t sv_g, int64_t sv_l,
                                        const float* inv_freq, int H, int D, int hidden, int G, int Rk, int Rv, int pos,
                                        int row, const void* bfrag, void* qfold, palu_stream_t stream) {
  const void* fold_b = nullptr;
  if (bfrag || qfold) {
    PALU_REQUIRE(bfrag && qfold && palu_abx_fold_bytes(H, G, Rk) != 0 && D == 128, PALU_ERR_UNSUPPORTED,
                 "decode_qkv_fold: the fold needs 4 heads per group, head_dim 128 and rank_k / G in {32, 64, 128} (H=%d G=%d D=%d Rk=%d)",
                 H, G, D, Rk);
    PALU_REQUIRE(((uintptr_t)qfold & 15) == 0, PALU_ERR_ARG, "decode_qkv_fold: qfold must be 16-byte aligned");
    fold_b = palu_abx_two_band_frags(bfrag, H, G, Rk);
    PALU_REQUIRE(fold_b, PALU_ERR_ARG, "decode_qkv_fold: no two-band fragments for this shape");
  }
  PALU_REQUIRE(wq && vtk && vtv && x && q_out && k_cache && v_cache && inv_freq, PALU_ERR_ARG, "decode_qkv: null pointer");
  PALU_REQUIRE(H > 0 && G > 0 && D > 0 && D % 2 == 0 && Rk % 2 == 0 && Rv % 2 == 0 && hidden % 8 == 0 && pos >= 0 &&
                   row >= 0,
               PALU_ERR_ARG, "decode_qkv: bad shape");
  PALU_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ((uintptr_t)wq & 15) == 0 && ((uintptr_t)vtk & 15) == 0 &&
                   ((uintptr_t)vtv & 15) == 0 && ((uintptr_t)x & 15) == 0,
               PALU_ERR_ARG, "decode_qkv: weights and x must be 16-byte aligned with row strides % 8 == 0");
  PALU_REQUIRE((size_t)hidden * 2 <= 64 * 1024, PALU_ERR_UNSUPPORTED, "decode_qkv: hidden too large");
  QkvParams p;
  p.wq = (const h16*)wq; p.vtk = (const h16*)vtk; p.vtv = (const h16*)vtv; p.x = (const h16*)x;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv;
  p.q_out = (h16*)q_out;
  p.k_cache = (h16*)k_cache; p.v_cache = (h16*)v_cache;
  p.sk_g = sk_g; p.sk_l = sk_l; p.sv_g = sv_g; p.sv_l = sv_l;
  p.inv_freq = inv_freq;
  p.H = H; p.D = D; p.K = hidden; p.rank_k = G * Rk; p.rank_v = G * Rv; p.Rk = Rk; p.Rv = Rv;
  p.pos = pos; p.row = row;
  p.q_bias = (const h16*)q_bias;
  p.fold_b = (const u32x4*)fold_b;
  p.qfold = (u32x4*)qfold;
  p.fold_nks = Rk / 16;
  const int pairs = H * D / 2 + p.rank_k / 2 + p.rank_v / 2;
  const int blocks = (pairs + GV_THREADS / 64 - 1) / (GV_THREADS / 64);
  hipLaunchKernelGGL(decode_qkv_kernel, dim3(blocks), dim3(GV_THREADS), (size_t)hidden * 2, (hipStream_t)stream, p);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}
