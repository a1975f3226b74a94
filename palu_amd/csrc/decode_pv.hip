// Softmax + latent-space P.V of the decode step, as split-L streaming kernels.
//
// Replaces kernel/palu_attention.py:219 (/sqrt(D)), :229-234 (mask), :238 (softmax fp32 -> fp16)
// and :246-251 (attn[1,G,gs,L] @ V_lat[1,G,L,Rv]).  This is the genuinely HBM-bound part of the
// step (the V latents are 3x the K latents and are touched once): no MFMA, no LDS staging of V --
// every lane streams 16-byte row chunks straight into registers (deep unroll, late waits) and the
// gs heads of a group share each V row.  Split-L with a log-sum-exp merge (flash-decoding):
//   pv_partial : WG = (group g, L-range) -> local max m, local sum S, partial sum_l e^(x-m) V[l,:]
//   pv_combine : merges the splits, normalises, rounds once to fp16
//   probs      : optional attention weights (output_attentions=True), fp16 like :238
// x = fp16(fp16(score)/sqrt(D)) [+ mask], the rounding points of the reference's fp16 tensors.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "palu_common.h"
#include "pv_mfma.h"

namespace {

constexpr int PV_THREADS = 256;

struct PvParams {
  const h16* scores;   // [H, L] raw abx output
  int64_t ss_h;
  const h16* mask;     // [L] additive or null
  const h16* v;        // [G, L, Rv]
  int64_t sv_g, sv_l;
  float* part;         // [G][nsplit][gs][Rv]
  float* ml;           // [G][nsplit][gs][2] (max, sum)
  int G, gs, L, Rv, nsplit, rps;
  float inv_scale;     // sqrt(D): the reference DIVIDES by it (palu_attention.py:219)
};

static __device__ __forceinline__ float scaled_logit(h16 s, float inv_scale, const h16* mask, int l) {
  // fp16 tensor / python float -> fp32 divide, rounded to fp16 (torch semantics); then + mask in fp16
  h16 x = (h16)((float)s / inv_scale);
  if (mask) x = (h16)((float)x + (float)mask[l]);
  return (float)x;
}

static __device__ __forceinline__ float block_max(float v, float* sh, int tid) {
  v = wave_max(v);
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
static __device__ __forceinline__ float block_sum(float v, float* sh, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// Sweep 1 of the softmax statistics for all GS heads of a group: the raw scores (and mask values) of up to PA_UN rows
// per thread are requested together, so a range costs ceil(rows / (PA_UN * threads)) memory latencies instead of one
// per row.  Writes the scaled logits to pl[h][i] and returns the running maxima.
constexpr int PA_UN = 4;
// one sweep = PA_UN x PV_THREADS positions: the loads (scores of the GS heads + the mask) ...
template <int GS>
static __device__ __forceinline__ void logits_load(const h16* scores, int64_t ss_h, const h16* mask, int g, int l0, int n, int k0,
                                                   h16 (&sc)[PA_UN][GS], h16 (&mk)[PA_UN], int tid) {
  const int nlast = max(n - 1, 0);
#pragma unroll
  for (int u = 0; u < PA_UN; ++u) {
    const int ic = min(tid + (k0 + u) * PV_THREADS, nlast);
#pragma unroll
    for (int h = 0; h < GS; ++h) sc[u][h] = scores[(int64_t)(g * GS + h) * ss_h + l0 + ic];
    mk[u] = mask ? mask[l0 + ic] : (h16)0.f;
  }
}
// ... and what is done with them: x = fp16(fp16(s) / sqrt(D)) (+ mask), the logits to LDS, the thread's running maxima
template <int GS>
static __device__ __forceinline__ void logits_apply(const h16 (&sc)[PA_UN][GS], const h16 (&mk)[PA_UN], bool masked, int n, int k0, int rps,
                                                    float inv_scale, float* pl, float (&mloc)[GS], int tid) {
#pragma unroll
  for (int u = 0; u < PA_UN; ++u) {
    const int i = tid + (k0 + u) * PV_THREADS;
    if (i < n) {
#pragma unroll
      for (int h = 0; h < GS; ++h) {
        // fp16 tensor / python float -> fp32 divide, rounded to fp16 (torch semantics); then + mask in fp16
        h16 x16 = (h16)((float)sc[u][h] / inv_scale);
        if (masked) x16 = (h16)((float)x16 + (float)mk[u]);
        const float x = (float)x16;
        pl[h * rps + i] = x;
        mloc[h] = fmaxf(mloc[h], x);
      }
    }
  }
}
template <int GS>
static __device__ __forceinline__ void logits_sweep(const h16* scores, int64_t ss_h, const h16* mask, int g, int l0, int n,
                                                    int rps, float inv_scale, float* pl, float (&mloc)[GS], int tid, int k_begin = 0) {
  for (int k0 = k_begin; k0 * PV_THREADS < n; k0 += PA_UN) {
    h16 sc[PA_UN][GS], mk[PA_UN];
    logits_load<GS>(scores, ss_h, mask, g, l0, n, k0, sc, mk, tid);
    logits_apply<GS>(sc, mk, mask != nullptr, n, k0, rps, inv_scale, pl, mloc, tid);
  }
}

// the same for GS values at once: one pair of barriers for all heads of the group
template <int GS, bool MAX>
static __device__ __forceinline__ void block_reduce(float (&v)[GS], float (*sh)[4], int tid) {
#pragma unroll
  for (int h = 0; h < GS; ++h) v[h] = MAX ? wave_max(v[h]) : wave_sum(v[h]);
  __syncthreads();
  if ((tid & 63) == 0) {
#pragma unroll
    for (int h = 0; h < GS; ++h) sh[h][tid >> 6] = v[h];
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < GS; ++h)
    v[h] = MAX ? fmaxf(fmaxf(sh[h][0], sh[h][1]), fmaxf(sh[h][2], sh[h][3])) : (sh[h][0] + sh[h][1]) + (sh[h][2] + sh[h][3]);
}

template <int GS>
__global__ __launch_bounds__(PV_THREADS) void pv_partial_kernel(PvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* pl = reinterpret_cast<float*>(smem_raw);  // [GS][rps] logits -> probabilities; later the reduce buffer
  __shared__ float sh[4];

  const int tid = threadIdx.x;
  const int g = blockIdx.x % p.G;
  const int split = blockIdx.x / p.G;
  const int l0 = split * p.rps;
  const int n = max(0, min(p.L - l0, p.rps));
  float* ml = p.ml + ((size_t)(g * p.nsplit + split) * GS) * 2;
  float* part = p.part + (size_t)(g * p.nsplit + split) * GS * p.Rv;

  // thread = (row group rg, 16-byte column chunk cc) of the V stream
  const int cpr = p.Rv >> 3;            // 16-byte chunks per row
  const int rpp = PV_THREADS / cpr;     // rows per pass
  const int rg = tid / cpr, cc = tid - rg * cpr;
  const bool streamer = rg < rpp;
  constexpr int U = 4;                  // rows per batch; two batches in flight (software pipeline)
  const h16* vb = p.v + (int64_t)g * p.sv_g + (int64_t)l0 * p.sv_l + cc * 8;
  const int nlast = max(n - 1, 0);
  // rows beyond the range are clamped (re-read, weight 0) so that the loop body is branch-free
  auto load_batch = [&](u32x4 (&raw)[U], int i) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      raw[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (int64_t)min(i + u * rpp, nlast) * p.sv_l));
  };
  // A wave's loads return in issue order: the range's scores (L2 / MALL: the score kernel has just written them) are requested
  // FIRST, then two batches of V rows (HBM).  The other way round the statistics waited for the V rows' round trip and HBM sat
  // idle while they were computed; now the stream runs from the first cycle and phase A works in its shadow.
  h16 sc0[PA_UN][GS], mk0[PA_UN];
  logits_load<GS>(p.scores, p.ss_h, p.mask, g, l0, n, 0, sc0, mk0, tid);
  u32x4 rawA[U], rawB[U];
  if (streamer) {
    load_batch(rawA, rg);
    load_batch(rawB, rg + U * rpp);
  }

  // ---- phase A: logits, local max, probabilities, local sum (per head of the group)
  // all heads of the group in one sweep: one block reduction (two barriers) for the maxima, one for the sums
  __shared__ float shg[GS][4];
  float mloc[GS], sloc[GS];
#pragma unroll
  for (int h = 0; h < GS; ++h) mloc[h] = -INFINITY;
  logits_apply<GS>(sc0, mk0, p.mask != nullptr, n, 0, p.rps, p.inv_scale, pl, mloc, tid);
  logits_sweep<GS>(p.scores, p.ss_h, p.mask, g, l0, n, p.rps, p.inv_scale, pl, mloc, tid, PA_UN);   // (ranges above 1024 rows)
  block_reduce<GS, true>(mloc, shg, tid);
#pragma unroll
  for (int h = 0; h < GS; ++h) sloc[h] = 0.f;
  for (int i = tid; i < p.rps; i += PV_THREADS) {
#pragma unroll
    for (int h = 0; h < GS; ++h) {
      float e = 0.f;                                   // rows past the range (clamped re-reads) weigh nothing
      if (i < n) e = (mloc[h] == -INFINITY) ? 0.f : __expf(pl[h * p.rps + i] - mloc[h]);
      pl[h * p.rps + i] = e;
      sloc[h] += e;
    }
  }
  block_reduce<GS, false>(sloc, shg, tid);
  if (tid == 0) {
#pragma unroll
    for (int h = 0; h < GS; ++h) {
      ml[2 * h] = mloc[h];
      ml[2 * h + 1] = sloc[h];
    }
  }
  __syncthreads();

  // ---- phase B: stream the V rows, two batches of U rows in flight per thread
  float acc[GS][8];
#pragma unroll
  for (int h = 0; h < GS; ++h)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[h][j] = 0.f;
  auto consume = [&](const u32x4 (&raw)[U], int i) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      h16x8 v8 = __builtin_bit_cast(h16x8, raw[u]);
      float vf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) vf[j] = (float)v8[j];
      const int row = min(i + u * rpp, p.rps - 1);
#pragma unroll
      for (int h = 0; h < GS; ++h) {
        float ph = (i + u * rpp < n) ? pl[h * p.rps + row] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[h][j] = fmaf(ph, vf[j], acc[h][j]);
      }
    }
  };
  if (streamer) {
    const int stride = U * rpp;
    int i = rg;   // rawA holds the batch whose first row is i, rawB the next one (both issued before phase A)
    for (;;) {
      consume(rawA, i);
      i += stride;
      if (i >= n) break;
      load_batch(rawA, i + stride);   // one batch ahead; clamped beyond the range (weight 0 in consume)
      consume(rawB, i);
      i += stride;
      if (i >= n) break;
      load_batch(rawB, i + stride);
    }
  }
  __syncthreads();   // everyone is done reading the probabilities: reuse the buffer for the reduction

  // ---- phase C: sum the row groups -> partial context [GS][Rv], one head at a time: the reduce buffer [rpp][Rv] stays
  //      below 8 KB (a workgroup of this kernel then fits in the LDS a score-kernel workgroup leaves free on its CU)
  float* redb = pl;   // [rpp][Rv]
#pragma unroll
  for (int h = 0; h < GS; ++h) {
    if (h > 0) __syncthreads();
    if (rg < rpp) {
      float* d = redb + (size_t)rg * p.Rv + cc * 8;
      *reinterpret_cast<f32x4*>(d) = f32x4{acc[h][0], acc[h][1], acc[h][2], acc[h][3]};
      *reinterpret_cast<f32x4*>(d + 4) = f32x4{acc[h][4], acc[h][5], acc[h][6], acc[h][7]};
    }
    __syncthreads();
    for (int o = tid; o < p.Rv; o += PV_THREADS) {
      float s = 0.f;
      for (int r = 0; r < rpp; ++r) s += redb[(size_t)r * p.Rv + o];
      part[h * p.Rv + o] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Quantised V latents (3/4-bit codes + per-row (scale, zero), quant.hip layout).  Same split-L scheme;
// a thread owns a 32-code chunk of the row (16 B at 4 bit, 12 B at 3 bit).  Dequantisation is folded
// into the accumulation:  sum_l p_l (c_l - z_l) s_l = sum_l (p_l s_l) c_l  -  sum_l p_l s_l z_l
// so the inner loop is: extract code, int->float, one FMA per head; the second term is one FMA per row.
struct PvQParams {
  const h16* scores;
  int64_t ss_h;
  const h16* mask;
  const unsigned char* codes;  // [G, L, Rv*bits/8]
  int64_t sc_g, sc_l;          // bytes
  const h16* meta;             // [G, L, 2]
  int64_t sm_g, sm_l;          // elements
  float* part;
  float* ml;
  int G, gs, L, Rv, nsplit, rps;
  float inv_scale;
  int exp_flags;   // 8: timeline dump of pv_partial_qr_kernel (PALU_PVQ_TIMELINE_DUMP, tools/time_pvq.py)
  int qr_nsl, qr_ncw, qr_s;   // register-direct kernel: column slices, chunks per slice, row sets per unit
  unsigned qr_park_off;       // LDS offset of the parked per-lane partial sums [8 waves][3][64 lanes] f32x4
  int qr_prio;                // 1: waves 4-7 run the unit loop at raised priority
  float rcp_scale;            // 1 / inv_scale when the 3-instruction quotient is exact for every fp16 score (else 0)
};

// one thread owns a 16-code column chunk (8 bytes at 4 bit, 6 bytes at 3 bit) and walks the rows in PAIRS:
// (1024+c_A, 1024+c_B) is formed as an fp16 pair by byte permutes and multiplied with the fp16 weight pair
// (w_A, w_B) of each head by ONE v_dot2_f32_f16 (fp32 accumulate): 2 code-MACs per instruction.
//   sum_l p_l s_l (c_l - z_l) = sum_l w_l (1024 + c_l)  -  sum_l w_l (1024 + z_l),   w_l = fp16(p_l s_l)
template <int BITS>
struct QRow {
  unsigned w0, w1;    // the 16 codes (4 bit: 64 bits; 3 bit: the 48 bits start at bit `bsh` of w1:w0)
};

template <int GS, int BITS>
__global__ __launch_bounds__(PV_THREADS) void pv_partial_q_kernel(PvQParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* pl = reinterpret_cast<float*>(smem_raw);                 // [GS][rps] logits -> e^(x-m); later the reduce buffer
  h16* wl = reinterpret_cast<h16*>(pl + (size_t)GS * p.rps);      // [GS][rps] fp16 weights w = e^(x-m) * scale_row
  __shared__ float sh[4];
  const int tid = threadIdx.x;
  const int g = blockIdx.x % p.G;
  const int split = blockIdx.x / p.G;
  const int l0 = split * p.rps;
  const int n = max(0, min(p.L - l0, p.rps));
  float* ml = p.ml + ((size_t)(g * p.nsplit + split) * GS) * 2;
  float* part = p.part + (size_t)(g * p.nsplit + split) * GS * p.Rv;

  const int cpr = p.Rv >> 4;            // 16-code chunks per row
  const int rpp = PV_THREADS / cpr;     // row PAIRS per pass
  const int rg = tid / cpr, cc = tid - rg * cpr;
  const bool streamer = rg < rpp;
  constexpr int NP = 2;                 // row pairs per batch; two batches in flight
  const int boff = (BITS == 4) ? cc * 8 : ((cc * 6) & ~3);     // dword-aligned byte offset of the chunk
  const int bsh = (BITS == 4) ? 0 : ((cc * 6) & 3) * 8;        // 3 bit: 0 or 16 bits into the first dword
  const unsigned char* cb = p.codes + (int64_t)g * p.sc_g + (int64_t)l0 * p.sc_l + boff;
  const h16* mb = p.meta + (int64_t)g * p.sm_g + (int64_t)l0 * p.sm_l;
  const int nlast = max(n - 1, 0);
  // adjacent rows (2*pi, 2*pi+1) form a pair: their weights are one aligned fp16x2 in LDS
  auto load_pair = [&](QRow<BITS> (&raw)[2 * NP], int slot, int pi) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int row = min(2 * pi + e, nlast);
      const unsigned* src = reinterpret_cast<const unsigned*>(cb + (int64_t)row * p.sc_l);
      raw[2 * slot + e].w0 = __builtin_nontemporal_load(src);
      raw[2 * slot + e].w1 = __builtin_nontemporal_load(src + 1);
    }
  };
  auto load_batch = [&](QRow<BITS> (&raw)[2 * NP], int pi) {
#pragma unroll
    for (int u = 0; u < NP; ++u) load_pair(raw, u, pi + u * rpp);
  };
  QRow<BITS> rawA[2 * NP], rawB[2 * NP];
  if (streamer) load_batch(rawA, rg);

  // ---- phase A (all heads per sweep):  x -> max -> e^(x-m) -> sum;  w = fp16(e * scale_row) -> LDS;
  //      corr = sum w (1024 + zero_row)
  __shared__ float shg[GS][4];
  float mloc[GS], sloc[GS], corr[GS];
#pragma unroll
  for (int h = 0; h < GS; ++h) mloc[h] = -INFINITY;
  // the (scale, zero) pairs of this thread's rows are requested before the score sweep (rps <= 2048: 8 rows at most)
  constexpr int MAXR = 2048 / PV_THREADS;
  unsigned metar[MAXR];
#pragma unroll
  for (int k = 0; k < MAXR; ++k)
    metar[k] = (k * PV_THREADS < n) ? *reinterpret_cast<const unsigned*>(mb + (int64_t)min(tid + k * PV_THREADS, nlast) * p.sm_l) : 0u;
  logits_sweep<GS>(p.scores, p.ss_h, p.mask, g, l0, n, p.rps, p.inv_scale, pl, mloc, tid);
  block_reduce<GS, true>(mloc, shg, tid);
#pragma unroll
  for (int h = 0; h < GS; ++h) {
    sloc[h] = 0.f;
    corr[h] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < MAXR; ++k) {
    const int i = tid + k * PV_THREADS;
    if (i >= p.rps) break;
    const h16x2 m2 = __builtin_bit_cast(h16x2, metar[k]);
#pragma unroll
    for (int h = 0; h < GS; ++h) {
      h16 wq = (h16)0.f;
      if (i < n) {
        const float e = (mloc[h] == -INFINITY) ? 0.f : __expf(pl[h * p.rps + i] - mloc[h]);
        sloc[h] += e;
        wq = (h16)(e * (float)m2[0]);
        corr[h] = fmaf((float)wq, 1024.f + (float)m2[1], corr[h]);
      }
      wl[h * p.rps + i] = wq;
    }
  }
  block_reduce<GS, false>(sloc, shg, tid);
  block_reduce<GS, false>(corr, shg, tid);
  if (tid == 0) {
#pragma unroll
    for (int h = 0; h < GS; ++h) {
      ml[2 * h] = mloc[h];
      ml[2 * h + 1] = sloc[h];
    }
  }
  __syncthreads();

  float acc[GS][16];
#pragma unroll
  for (int h = 0; h < GS; ++h)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[h][j] = 0.f;

  // (1024 + c_A, 1024 + c_B) fp16 pairs of the 16 columns of a row pair
  auto pair_words = [&](const QRow<BITS>& ra, const QRow<BITS>& rb, unsigned (&pw)[16]) {
    if (BITS == 4) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const unsigned wa = d ? ra.w1 : ra.w0, wb = d ? rb.w1 : rb.w0;
        const unsigned ae = wa & 0x0F0F0F0Fu, ao = (wa >> 4) & 0x0F0F0F0Fu;   // even / odd codes as bytes
        const unsigned be = wb & 0x0F0F0F0Fu, bo = (wb >> 4) & 0x0F0F0F0Fu;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned sel = 0x0C000C00u | ((4u + k) << 16) | (unsigned)k;   // bytes [A_k, 0, B_k, 0]
          pw[8 * d + 2 * k] = __builtin_amdgcn_perm(be, ae, sel) | 0x64006400u;
          pw[8 * d + 2 * k + 1] = __builtin_amdgcn_perm(bo, ao, sel) | 0x64006400u;
        }
      }
    } else {
      unsigned ga[2], gb[2];
      {
        const unsigned lo = bsh ? __builtin_amdgcn_alignbit(ra.w1, ra.w0, 16) : ra.w0;
        const unsigned hi = bsh ? (ra.w1 >> 16) : ra.w1;
        ga[0] = lo;
        ga[1] = __builtin_amdgcn_alignbit(hi, lo, 24);
      }
      {
        const unsigned lo = bsh ? __builtin_amdgcn_alignbit(rb.w1, rb.w0, 16) : rb.w0;
        const unsigned hi = bsh ? (rb.w1 >> 16) : rb.w1;
        gb[0] = lo;
        gb[1] = __builtin_amdgcn_alignbit(hi, lo, 24);
      }
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned a = ((ga[d] >> (3 * e)) & 7u) | 0x64006400u;
          const int shl = 16 - 3 * e;                                    // move code e of B to bits 16..18
          const unsigned bsft = shl >= 0 ? (gb[d] << shl) : (gb[d] >> (-shl));
          pw[8 * d + e] = (bsft & 0x70000u) | a;
        }
    }
  };
  auto consume = [&](const QRow<BITS> (&raw)[2 * NP], int pi) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int row2 = min(2 * (pi + u * rpp), p.rps - 2);     // rows beyond the range carry weight 0 in wl
      h16x2 w2[GS];
#pragma unroll
      for (int h = 0; h < GS; ++h) {
        const unsigned wv = *reinterpret_cast<const unsigned*>(wl + h * p.rps + row2);
        w2[h] = (2 * (pi + u * rpp) < p.rps) ? __builtin_bit_cast(h16x2, wv) : h16x2{(h16)0.f, (h16)0.f};
      }
      unsigned pw[16];
      pair_words(raw[2 * u], raw[2 * u + 1], pw);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const unsigned pj = pw[j];
        const h16x2 cp = __builtin_bit_cast(h16x2, pj);
#pragma unroll
        for (int h = 0; h < GS; ++h) acc[h][j] = __builtin_amdgcn_fdot2(cp, w2[h], acc[h][j], false);
      }
    }
  };
  if (streamer) {
    const int stride = NP * rpp;          // in row pairs
    int pi = rg;
    for (;;) {
      load_batch(rawB, pi + stride);
      consume(rawA, pi);
      pi += stride;
      if (2 * pi >= n) break;
      load_batch(rawA, pi + stride);
      consume(rawB, pi);
      pi += stride;
      if (2 * pi >= n) break;
    }
  }
  // ---- cross-row-group reduction, one head per pass through a [threads][16] LDS buffer
  float* redb = pl;
#pragma unroll
  for (int h = 0; h < GS; ++h) {
    __syncthreads();
    if (streamer) {
#pragma unroll
      for (int j = 0; j < 16; j += 4)
        *reinterpret_cast<f32x4*>(redb + (size_t)tid * 16 + j) = f32x4{acc[h][j], acc[h][j + 1], acc[h][j + 2], acc[h][j + 3]};
    }
    __syncthreads();
    for (int r = tid; r < p.Rv; r += PV_THREADS) {
      const int c = r >> 4, j = r & 15;
      float s = -corr[h];
      for (int q = 0; q < rpp; ++q) s += redb[(size_t)(q * cpr + c) * 16 + j];
      part[h * p.Rv + r] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Quantised V latents, register-direct (pv_partial_qr_kernel): packed rows go HBM -> VGPRs -> MFMA operands with no LDS, no
// barrier and no transpose read on the V path.  (Its predecessor wrote an fp16 image of the codes to LDS and read it back
// transposed: a ds_write + barrier + ds_read_tr round trip per 32 rows, latency-bound; 48 instead of 29 us at C3.)
// The MFMA contracts over ROWS, and which row sits in which k-slot is ours to choose as long as both operands agree:
//   v_mfma_f32_16x16x32_f16: D[m, n] += sum_k A[m, k] B[k, n];   lane = (m or n = lane % 16, q = lane / 16) holds
//   the 8 k-slots e = 0..7 of k-group q.  k-slot (q, e) <-> row 4e + q of a 32-row set, so
//   B: lane (n, q) loads ITS OWN 8 rows x one 32-code chunk (8 dwordx3 / dwordx4 buffer loads; instruction e covers
//      rows 4e..4e+3 of the set: whole consecutive rows, coalesced) and turns them into 32 operands (one per code
//      column j of the chunk) with byte permutes that pair rows (2p, 2p+1) plus one v_and_or per code pair: the code
//      stays where it is inside the 16-bit half and the fp16 exponent is chosen for that bit position --
//      ((w & (7 << s)) | fp16(2^(10-s)))  ==  2^(10-s) + code  for s <= 7 -- so most codes need no shift
//      (0.75-0.81 VALU per code instead of 1.5 + the LDS traffic);
//   A: lane (m, q) reads the 8 weights w = fp16(e^(x-m) * scale_row) of its rows with one ds_read_b128 (the weight
//      rows of a 64-row batch sit in the wave's LDS patch in k-slot order).
// n = (row set s, chunk c), m = (row set s, head h): with Rv/32 = 12 chunks one set of 32 rows fills 12 of the 16 n
// lanes, with <= 8 chunks two sets share an MFMA; D[(s',h), (s,c)] is meaningful for s == s' and ignored otherwise
// (D lane (n, q) holds rows m = 4q..4q+3 = the heads of row set q).  One accumulator per code column of the chunk lives
// in registers for the whole range; the per-column offsets 2^(10-s) and the zero points leave at the end:
// out = acc - sum_l w_l z_l - off_j sum_l w_l.  Rows wider than 512 codes are cut into column slices handled by different
// waves.  Statistics are computed online in 64-row batches inside the unit loop (see there); the only workgroup barrier
// precedes the merge of the waves.  BITS = 16 runs plain fp16 rows through the same structure (8-column chunks, the
// byte permute is the whole decode).
template <int BITS>
struct QrDecode {
  // offsets of the decoded columns (fp16 value = off + code), in MFMA order j % 8 (3 bit) or j % 4 (4 bit)
  static __device__ __forceinline__ float off(int j) {
    if (BITS == 3) {
      const int k = j & 7;
      return (k == 0 || k == 3 || k == 6) ? 1024.f : (k == 1 || k == 4 || k == 7) ? 128.f : (k == 2) ? 16.f : 8.f;
    }
    if (BITS == 16) return 0.f;
    return (j & 1) ? 64.f : 1024.f;
  }
};

// wave-wide maximum / sum by DPP (no LDS): quad permutes, row mirrors, row broadcasts; lane 63 ends up with the result
static __device__ __forceinline__ float wave_max_dpp(float v) {
#define PALU_DPP_MAX(CTRL, ROWMASK)                                                                                    \
  v = fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), \
                                                                       CTRL, ROWMASK, 0xF, false)))
  PALU_DPP_MAX(0xB1, 0xF);    // quad_perm [1,0,3,2]
  PALU_DPP_MAX(0x4E, 0xF);    // quad_perm [2,3,0,1]
  PALU_DPP_MAX(0x141, 0xF);   // row_half_mirror
  PALU_DPP_MAX(0x140, 0xF);   // row_mirror
  PALU_DPP_MAX(0x142, 0xA);   // row_bcast:15 -> rows 1 and 3
  PALU_DPP_MAX(0x143, 0xC);   // row_bcast:31 -> rows 2 and 3
#undef PALU_DPP_MAX
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
static __device__ __forceinline__ float wave_sum_dpp(float v) {
#define PALU_DPP_ADD(CTRL, ROWMASK)                                                                                    \
  v += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, ROWMASK, 0xF, true))
  PALU_DPP_ADD(0xB1, 0xF);    // quad_perm [1,0,3,2]
  PALU_DPP_ADD(0x4E, 0xF);    // quad_perm [2,3,0,1]
  PALU_DPP_ADD(0x141, 0xF);   // row_half_mirror
  PALU_DPP_ADD(0x140, 0xF);   // row_mirror
  PALU_DPP_ADD(0x142, 0xA);   // row_bcast:15 -> rows 1 and 3
  PALU_DPP_ADD(0x143, 0xC);   // row_bcast:31 -> rows 2 and 3
#undef PALU_DPP_ADD
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
// (a & mask) | magic in ONE instruction: gfx9 VOP3 takes no literals and one scalar operand, and with literal operands
// hipcc selects v_and_b32 + v_or_b32 (twice the VALU work).  The masks are therefore handed to the compiler as opaque
// SGPR values and the magics as opaque VGPR values: the plain C expression then selects v_and_or_b32 v, v, s, v.
// (Writing the instruction as inline asm is NOT an option: the hazard recogniser does not see an asm as a VALU write and
//  leaves out the wait states an MFMA reading the result needs -- measured: sporadic NaNs in one code column.)
static __device__ __forceinline__ unsigned and_or(unsigned a, unsigned mask_sgpr, unsigned magic_vgpr) {
  return (a & mask_sgpr) | magic_vgpr;
}
static __device__ __forceinline__ unsigned vgpr_const(unsigned c) {
  unsigned d;
  asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "s"(c));
  return d;
}
static __device__ __forceinline__ unsigned sgpr_const(unsigned c) {
  unsigned d;
  asm volatile("s_mov_b32 %0, %1" : "=s"(d) : "i"(c));
  return d;
}

template <int GS, int BITS, int S, int CWT = 32>
__global__ __launch_bounds__(512) void pv_partial_qr_kernel(PvQParams p) {
  // BITS = 3 / 4: packed codes, 32 per chunk.  BITS = 16: plain fp16 latents, 8 per chunk (16 bytes) -- the same kernel
  // without a decode step: 8 accumulators instead of 32 leave room for four register sets of units in flight.
  // CWT = 24 (4 bit only: 12-byte chunks) is for rows whose 32-code chunks would leave a quarter of the MFMA lanes idle:
  // 192 columns = 6 x 32 = 8 x 24 with two row sets (C4).  (Measured and dropped for 3 bit -- 9-byte chunks loaded as
  // unaligned dwordx3, 384 = 16 x 24 columns: 26.7 us against 26.4 us at C3.  The unit loop runs at 5.8 TB/s either way.)
  constexpr bool FP16 = BITS == 16;
  static_assert(CWT == 32 || (CWT == 24 && BITS == 4), "24-code chunks: 4 bit only");
  constexpr int CW = FP16 ? 8 : CWT;                     // columns per chunk = MFMAs per unit = accumulators
  constexpr int CHB = FP16 ? 16 : CW * BITS / 8;         // bytes per chunk: 12 / 16 (32 codes), 12 (24 codes at 4 bit)
  constexpr int RDW = (BITS == 3 || CWT == 24) ? 3 : 4;  // dwords a lane loads per row
  constexpr int NW = 8, NSET = FP16 ? 4 : 2, NJ = CW;    // (packed: a third set spills next to 128 accumulator registers;
                                                         //  next to the 96 of 24-code chunks it fits and measured 1 us slower)
  constexpr int NJP = NJ + 1;                // red: code columns + 1 pad (chunks on different banks)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const unsigned smem_lds = (unsigned)reinterpret_cast<uintptr_t>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x % p.G;
  const int split = blockIdx.x / p.G;
  const int l0 = split * p.rps;
  const int n = max(0, min(p.L - l0, p.rps));
  long long tstamp[5] = {0, 0, 0, 0, 0};
  const bool stamps = (p.exp_flags & 8) != 0;               // timeline dump (tools/time_pvq.py)
  if (stamps) tstamp[0] = wall_clock64();

  // ---- geometry (uniform): column slices of <= 16 chunks, S row sets per unit, every wave owns rpw consecutive rows
  const int nch = p.Rv / CW;
  const int nsl = p.qr_nsl, ncw = p.qr_ncw;                      // slices, chunks per slice (host plan; S = p.qr_s row sets)
  constexpr int RU = 32 * S;
  const int sl = wv % nsl, wph = wv / nsl, nws = NW / nsl;       // this wave: slice, row phase; waves per slice
  // rows per wave: a multiple of RU, any number of batches.  (The two waves of a SIMD do not run at the same speed -- the
  // one dispatched first leaves the loop at 12 us, the other at 18 us -- but shifting rows from one to the other, 32..192
  // of 256, changes nothing: the unit loop as a whole streams at 5.8 TB/s, profiles/r04_pv_q_experiments.txt)
  const int rpw = p.rps / nws;
  const int r0 = wph * rpw;
  const int nw = max(0, min(n - r0, rpw));                       // valid rows of this wave
  const int nwlast = max(nw - 1, 0);
  const int nunit = (nw + RU - 1) / RU;
  const int m = lane & 15, q = lane >> 4;
  float* ml = p.ml + ((size_t)(g * p.nsplit + split) * GS) * 2;
  float* part = p.part + (size_t)(g * p.nsplit + split) * GS * p.Rv;
  // LDS: per-wave weight rows of one batch [4][64 + 8] fp16 | per-wave red [S * ncw][33][4 heads] fp32 (the accumulators on their way
  // out) | stat [NW][GS][4] fp32 (max, sum, zero-point term, weight sum)
  float* red_all = reinterpret_cast<float*>(smem_raw + (size_t)NW * 4 * (64 + 8) * sizeof(h16));
  float* red = red_all + (size_t)wv * (16 * NJP * 4);
  float* stat = red_all + (size_t)NW * (16 * NJP * 4);
  // V operand lanes: n = (row set sA, chunk cA)
  const int sA = m / ncw, cA = m - sA * ncw, chunk = sl * ncw + cA;
  const bool actA = sA < S && chunk < nch;
  const unsigned long long cbase = reinterpret_cast<unsigned long long>(p.codes + (int64_t)g * p.sc_g);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(cbase), 0, (int)((int64_t)(p.L - 1) * p.sc_l + (int64_t)p.Rv * BITS / 8), 0x00020000);
  // per-lane byte offset of (row l0 + r0 + 32 sA + q, chunk); inactive lanes sit past the descriptor (zeros, no traffic)
  const unsigned sc_l = (unsigned)p.sc_l;
  const unsigned voff0 = actA ? (unsigned)(l0 + r0 + 32 * sA + q) * sc_l + (unsigned)(chunk * CHB) : 0x80000000u;

  unsigned raw[NSET][8][RDW];
  auto load_unit = [&](unsigned (&r)[8][RDW], int u) {
    // (units past this wave's range belong to the next split: sent past the descriptor, they cost no traffic)
    const unsigned vo = u < nunit ? voff0 + (unsigned)(u * RU) * sc_l : 0x80000000u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (RDW == 3) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b96(rs, vo + (unsigned)(4 * e) * sc_l, 0, 0);
        r[e][0] = v[0]; r[e][1] = v[1]; r[e][2] = v[2];
      } else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + (unsigned)(4 * e) * sc_l, 0, 0);
        r[e][0] = v[0]; r[e][1] = v[1]; r[e][2] = v[2]; r[e][3 % RDW] = v[3];
      }
    }
  };

  // ---- softmax statistics ONLINE, in batches of 64 rows of this wave's range (no workgroup barrier: a wave
  //      is its own split until the epilogue merges the waves with exp(m_wave - m_range)).  Per batch: scaled logits ->
  //      does any row beat the running maximum (a ballot)?  If so (wave-uniform, rare after the first batches): wave
  //      maximum by DPP, accumulators and partial sums rescaled -> weights fp16(e^(x-m) * scale_row) into the wave's LDS patch in
  //      k-slot order -> the batch's units run on the matrix cores.  The raw scores / (scale, zero) of batch b+1 are
  //      requested before the units of batch b run, into the registers batch b has just vacated; the V units stream through
  //      two register sets all along, so HBM never waits for the statistics and a range may have any number of rows.
  constexpr int UB = 64 / RU;                                // units per 64-row batch: 2 (S = 1) or 1 (S = 2)
  const int nbatch = (nw + 63) >> 6;
  // per-lane partial sums (sum e, sum w z, sum w; 4 heads each) live in LDS between batches: 12 registers that the unit
  // loop needs more (with them in VGPRs hipcc spilled an accumulator inside the loop)
  typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
  const unsigned park = smem_lds + p.qr_park_off + (unsigned)((wv * 3 * 64 + lane) * sizeof(f32x4));
#pragma unroll
  for (int t = 0; t < 3; ++t) *(lds_f32x4*)(uintptr_t)(park + t * 64 * sizeof(f32x4)) = f32x4{0.f, 0.f, 0.f, 0.f};
  float mrun[GS];
#pragma unroll
  for (int h = 0; h < GS; ++h) mrun[h] = -INFINITY;
  h16 sc[GS], mk;
  unsigned mt = 0x00003C00u;                                 // (scale 1, zero 0): fp16 latents have no meta
  const h16* mb = FP16 ? nullptr : p.meta + (int64_t)g * p.sm_g + (int64_t)(l0 + r0) * p.sm_l;
  const h16* mkp = p.mask ? p.mask : p.scores + (int64_t)g * GS * p.ss_h;   // branch-free: a dummy row when there is no mask
  const bool has_mask = p.mask != nullptr;
  auto load_scores = [&](int b) {
    const int ic = min(b * 64 + lane, nwlast);
#pragma unroll
    for (int h = 0; h < GS; ++h) sc[h] = p.scores[(int64_t)(g * GS + h) * p.ss_h + l0 + r0 + ic];
    mk = mkp[l0 + r0 + ic];
    if (!FP16) mt = *reinterpret_cast<const unsigned*>(mb + (int64_t)ic * p.sm_l);
  };
  if (nw > 0) load_scores(0);                                // (spare waves of a group's last range touch no memory)
#pragma unroll
  for (int s = 0; s < NSET; ++s) load_unit(raw[s], s);       // requested BEHIND the small score loads: they return first

  // ---- P is the A operand (m = 4 * row set + head), the decoded codes the B operand (n = V lane):
  //      D lane (n, q) holds rows m = 4q .. 4q+3 = the 4 heads of row set q -> meaningful where q == sA(n)
  f32x4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int sP = m >> 2, hP = m & 3;
  const bool actP = sP < S && hP < GS;
  const int WS = 64 + 8;                                     // one batch of weights per head
  const unsigned wl_wave = smem_lds + (unsigned)(wv * 4 * WS * sizeof(h16));
  h16* wl = reinterpret_cast<h16*>(smem_raw) + (size_t)wv * 4 * WS;
  const unsigned wl_lane = wl_wave + (unsigned)(((actP ? hP : 0) * WS + (actP ? sP : 0) * 32 + q * 8) * sizeof(h16));
  typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
  unsigned M1024 = 0, M128 = 0, M16 = 0, M8 = 0, M64 = 0, K07 = 0, K38 = 0, K1C0 = 0, K380 = 0, K0F = 0, KF0 = 0;
  if (BITS == 3) {
    M1024 = vgpr_const(0x64006400u); M128 = vgpr_const(0x58005800u); M16 = vgpr_const(0x4C004C00u); M8 = vgpr_const(0x48004800u);
    K07 = sgpr_const(0x00070007u); K38 = sgpr_const(0x00380038u); K1C0 = sgpr_const(0x01C001C0u); K380 = sgpr_const(0x03800380u);
  } else if (BITS == 4) {
    M1024 = vgpr_const(0x64006400u); M64 = vgpr_const(0x54005400u);
    K0F = sgpr_const(0x000F000Fu); KF0 = sgpr_const(0x00F000F0u);
  }
  // One unit: row-pair windows (byte permutes) -> code pairs as fp16 pairs (v_and_or_b32) -> MFMA, one block of 8 (4)
  // code columns at a time.  (Column-major -- 4 operand registers live instead of 32 -- measured slower: every MFMA then
  // waits on the v_and_or just in front of it.)
  // (Requesting the set's next unit right after the windows are formed -- the packed words are dead from there -- was
  //  tried: it keeps two units in flight all the time, but the 32 window registers on top of both sets spill.)
  auto consume = [&](const unsigned (&r)[8][RDW], int u) {
    constexpr int NB = BITS == 3 ? CW / 8 : CW / 4, NWIN = BITS == 3 ? 2 : 1, NK = BITS == 3 ? 8 : 4;
    const u32x4 pw = *(const lds_u32x4*)(uintptr_t)(wl_lane + (unsigned)(u * RU * sizeof(h16)));   // u: unit inside the batch
    const h16x8 pop = __builtin_bit_cast(h16x8, pw);
    if (FP16) {
      // dword d of a row holds columns 2d, 2d+1: one byte permute per (row pair, column) is the whole "decode"
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        unsigned lo[4], hi[4];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          lo[pr] = __builtin_amdgcn_perm(r[2 * pr + 1][d % RDW], r[2 * pr][d % RDW], 0x05040100u);
          hi[pr] = __builtin_amdgcn_perm(r[2 * pr + 1][d % RDW], r[2 * pr][d % RDW], 0x07060302u);
        }
        acc[(2 * d) % NJ] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
            pop, __builtin_bit_cast(h16x8, u32x4{lo[0], lo[1], lo[2], lo[3]}), acc[(2 * d) % NJ], 0, 0, 0);
        acc[(2 * d + 1) % NJ] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
            pop, __builtin_bit_cast(h16x8, u32x4{hi[0], hi[1], hi[2], hi[3]}), acc[(2 * d + 1) % NJ], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      unsigned W[4][NWIN], ws1[4], ws2[4];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        if (BITS == 3) {
          // code column 8b + k of the chunk sits at bit 3k of the 3 bytes 3b .. 3b+2 of the row: bring those bytes to
          // the front of a dword per row, then W1 = bytes (0,1) and W2 = bytes (1,2) of rows (2p, 2p+1) side by side
          unsigned x0, x1;
          if (b == 0) { x0 = r[2 * pr][0]; x1 = r[2 * pr + 1][0]; }
          else if (b == 1) {
            x0 = __builtin_amdgcn_alignbit(r[2 * pr][1], r[2 * pr][0], 24);
            x1 = __builtin_amdgcn_alignbit(r[2 * pr + 1][1], r[2 * pr + 1][0], 24);
          } else if (b == 2) {
            x0 = __builtin_amdgcn_alignbit(r[2 * pr][2], r[2 * pr][1], 16);
            x1 = __builtin_amdgcn_alignbit(r[2 * pr + 1][2], r[2 * pr + 1][1], 16);
          } else { x0 = r[2 * pr][2]; x1 = r[2 * pr + 1][2]; }   // bytes 9..11 = bytes 1..3 of dword 2
          W[pr][0] = __builtin_amdgcn_perm(x1, x0, b == 3 ? 0x06050201u : 0x05040100u);
          W[pr][NWIN - 1] = __builtin_amdgcn_perm(x1, x0, b == 3 ? 0x07060302u : 0x06050201u);
        } else {
          // code column 4b + k sits at bit 4k of bytes 2b, 2b+1 of the row
          W[pr][0] = __builtin_amdgcn_perm(r[2 * pr + 1][(b >> 1) % RDW], r[2 * pr][(b >> 1) % RDW],
                                           (b & 1) ? 0x07060302u : 0x05040100u);
        }
        ws1[pr] = W[pr][0] >> (BITS == 3 ? 9 : 8);
        ws2[pr] = W[pr][NWIN - 1] >> 10;
      }
      unsigned o[4][NK];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const unsigned w1 = W[pr][0], w2 = W[pr][NWIN - 1];
        if (BITS == 3) {
          o[pr][0] = and_or(w1, K07, M1024);
          o[pr][1] = and_or(w1, K38, M128);
          o[pr][2] = and_or(w1, K1C0, M16);
          o[pr][3] = and_or(ws1[pr], K07, M1024);
          o[pr][4 % NK] = and_or(ws1[pr], K38, M128);
          o[pr][5 % NK] = and_or(w2, K380, M8);
          o[pr][6 % NK] = and_or(ws2[pr], K07, M1024);
          o[pr][7 % NK] = and_or(ws2[pr], K38, M128);
        } else {
          o[pr][0] = and_or(w1, K0F, M1024);
          o[pr][1] = and_or(w1, KF0, M64);
          o[pr][2] = and_or(ws1[pr], K0F, M1024);
          o[pr][3] = and_or(ws1[pr], KF0, M64);
        }
      }
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const h16x8 vop = __builtin_bit_cast(h16x8, u32x4{o[0][k], o[1][k], o[2][k], o[3][k]});
        acc[(NK * b + k) % NJ] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pop, vop, acc[(NK * b + k) % NJ], 0, 0, 0);
      }
    }
  };

  // statistics of batch b (its raw scores are in sc / mk / mt), weights into the LDS patch; then batch b+1 is requested.
  // Branch-free except for the (wave-uniform, rare) rescale.
  auto batch_stats = [&](int b) {
    const bool ok = b * 64 + lane < nw;
    const float mkf = has_mask ? (float)mk : 0.f;            // x + 0 leaves the fp16 logit as it is
    float xl[GS], mb_[GS];
    bool raise = false;
#pragma unroll
    for (int h = 0; h < GS; ++h) {
      // fp16 tensor / python float -> fp32 divide, rounded to fp16 (torch semantics); then + mask in fp16.  The IEEE
      // quotient costs ~12 instructions; q0 = x r, q1 = fma(fma(-q0, d, x), r, q0) gives the same bits for EVERY fp16 x
      // (the host checks that for the divisor of the launch -- pv_exact_rcp, all 63488 finite inputs; true for sqrt(128) --
      // and takes the older kernels otherwise)
      const float xs = (float)sc[h];
      const float q0 = xs * p.rcp_scale;
      const h16 x16 = (h16)fmaf(fmaf(-q0, p.inv_scale, xs), p.rcp_scale, q0);
      xl[h] = ok ? (float)(h16)((float)x16 + mkf) : -INFINITY;
      raise = raise || __builtin_amdgcn_ballot_w64(xl[h] > mrun[h]) != 0;
    }
    f32x4 ps = *(const lds_f32x4*)(uintptr_t)(park), pz = *(const lds_f32x4*)(uintptr_t)(park + 64 * sizeof(f32x4)),
          pw = *(const lds_f32x4*)(uintptr_t)(park + 128 * sizeof(f32x4));
    if (raise) {                                             // wave-uniform: some row of the batch beats a running maximum
      // (the wave-wide maximum -- 6 DPP steps per head plus their hazard wait states, ~140 instructions per batch -- is
      //  only evaluated in here: after the first batches of a range almost never)
      f32x4 al = f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int h = 0; h < GS; ++h) {
        mb_[h] = wave_max_dpp(xl[h]);
        const float mn = fmaxf(mrun[h], mb_[h]);
        al[h] = mrun[h] == -INFINITY ? 0.f : __expf(mrun[h] - mn);
        mrun[h] = mn;
      }
      ps *= al;
      pz *= al;
      pw *= al;
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] *= al;
    }
    const h16x2 m2 = __builtin_bit_cast(h16x2, mt);
    const int slot = ((lane & 32)) + ((lane & 3) << 3) + ((lane & 31) >> 2);   // row 32c + 4e + q -> k-slot order 32c + 8q + e
#pragma unroll
    for (int h = 0; h < GS; ++h) {
      const float e = xl[h] == -INFINITY ? 0.f : __expf(xl[h] - mrun[h]);      // (x > -inf implies m >= x > -inf)
      const h16 wq = (h16)(e * (float)m2[0]);
      ps[h] += e;
      pw[h] += (float)wq;
      pz[h] = fmaf((float)wq, (float)m2[1], pz[h]);
      wl[h * WS + slot] = wq;
    }
    *(lds_f32x4*)(uintptr_t)(park) = ps;
    *(lds_f32x4*)(uintptr_t)(park + 64 * sizeof(f32x4)) = pz;
    *(lds_f32x4*)(uintptr_t)(park + 128 * sizeof(f32x4)) = pw;
    load_scores(b + 1);                                      // (clamped to the range: harmless after the last batch)
  };
  if (stamps) tstamp[1] = wall_clock64();
  // unit u of the range sits in register set u % NSET (the loop is unrolled by NSET: the set index is a compile-time
  // constant); a batch of statistics precedes the first of its UB units
  // the second-dispatched wave of every SIMD loses issue arbitration to the older one (measured: waves 0-3 left this loop
  // at 11 us, waves 4-7 at 17.5 us).  Static priority for the younger half (PALU_PVQ_PRIO=1, default) is worth 3 %
  // (28.6 vs 29.5 us at C3, same box); taking turns per iteration (=2) is worth nothing.
  if (p.qr_prio == 1 && wv >= NW / 2) __builtin_amdgcn_s_setprio(1);
  for (int base = 0; base < nunit; base += NSET) {
    if (p.qr_prio == 2) {                       // the two waves of a SIMD take turns
      if (((base / NSET) + (wv >= NW / 2 ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
      const int u = base + s;
      if (u < nunit) {
        if ((u & (UB - 1)) == 0) batch_stats(u / UB);
        consume(raw[s], u & (UB - 1));
      }
      load_unit(raw[s], u + NSET);
    }
  }
  if (p.qr_prio) __builtin_amdgcn_s_setprio(0);
  if (stamps) tstamp[2] = wall_clock64();
  float mloc[GS], sloc[GS], cz[GS], cw[GS];
  {
    const f32x4 ps = *(const lds_f32x4*)(uintptr_t)(park), pz = *(const lds_f32x4*)(uintptr_t)(park + 64 * sizeof(f32x4)),
                pw = *(const lds_f32x4*)(uintptr_t)(park + 128 * sizeof(f32x4));
#pragma unroll
    for (int h = 0; h < GS; ++h) {
      mloc[h] = mrun[h];
      sloc[h] = ps[h];
      cz[h] = pz[h];
      cw[h] = pw[h];
    }
  }

  // ---- merge of the waves: every wave parks its accumulators and statistics in its own LDS patch, ONE barrier (the only
  //      one of the kernel: waves reach it at their own pace, and with one workgroup per CU a waiting wave costs nothing),
  //      then all waves share the columns of the sum.  (A barrier-free "last arrival merges" variant was measured: a lone
  //      wave issues ~1 instruction per 8 cycles and needed 6-8 us for what 8 waves do in well under 1.)
#pragma unroll
  for (int h = 0; h < GS; ++h) {
    sloc[h] = wave_sum_dpp(sloc[h]);
    cz[h] = wave_sum_dpp(cz[h]);
    cw[h] = wave_sum_dpp(cw[h]);
  }
  if (lane == 0) {
#pragma unroll
    for (int h = 0; h < GS; ++h)
      *reinterpret_cast<f32x4*>(stat + (wv * GS + h) * 4) = nw > 0 ? f32x4{mloc[h], sloc[h], cz[h], cw[h]}
                                                                     : f32x4{-INFINITY, 0.f, 0.f, 0.f};
  }
  // red [row set * ncw + chunk][33 (32 code columns + 1 pad: chunks on different banks)][4 heads]
  if (actA && q == sA) {
    f32x4* dst = reinterpret_cast<f32x4*>(red) + (size_t)(sA * ncw + cA) * NJP;
#pragma unroll
    for (int j = 0; j < NJ; ++j) dst[j] = acc[j];
  }
  if (stamps) tstamp[3] = wall_clock64();
  __syncthreads();
  {
    // factors exp(m_wave - m_range) and the scalar terms.  Straight-line code (selects, no branches): the LDS reads are
    // independent and are waited for once.  Waves without rows parked (-inf, 0, 0, 0): factor 0, and what is read from
    // their (zeroed) accumulators is 0 as well.
    f32x4 st[NW][GS];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int h = 0; h < GS; ++h)
        st[w][h] = *reinterpret_cast<const f32x4*>(stat + ((min(w, nws - 1) * nsl + sl) * GS + h) * 4);
    float mr[GS], fz[GS], fw[GS], fs[GS], f[NW][GS];
#pragma unroll
    for (int h = 0; h < GS; ++h) {
      mr[h] = -INFINITY;
#pragma unroll
      for (int w = 0; w < NW; ++w) mr[h] = fmaxf(mr[h], w < nws ? st[w][h][0] : -INFINITY);
      fz[h] = fw[h] = fs[h] = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const bool live = w < nws && st[w][h][0] != -INFINITY;
        f[w][h] = live ? __expf(st[w][h][0] - mr[h]) : 0.f;
        fs[h] = fmaf(f[w][h], st[w][h][1], fs[h]);
        fz[h] = fmaf(f[w][h], st[w][h][2], fz[h]);
        fw[h] = fmaf(f[w][h], st[w][h][3], fw[h]);
      }
    }
    const int ncol = min(ncw, nch - sl * ncw) * CW;                  // columns of this slice, shared by its nws waves
    for (int c = wph * 64 + lane; c < ncol; c += 64 * nws) {
      const int c2 = c / CW, jq = c % CW;
      const float off = QrDecode<BITS>::off(jq);
      f32x4 a[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) a[w] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < S; ++s) {                     // (uniform trip count; the NW reads of one pass are independent)
        f32x4 t[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w)
          t[w] = reinterpret_cast<const f32x4*>(red_all + (size_t)(min(w, nws - 1) * nsl + sl) * (16 * NJP * 4))[(size_t)(s * ncw + c2) * NJP + jq];
#pragma unroll
        for (int w = 0; w < NW; ++w) a[w] += t[w];
      }
#pragma unroll
      for (int h = 0; h < GS; ++h) {
        float o = -fz[h] - off * fw[h];
#pragma unroll
        for (int w = 0; w < NW; ++w) o = fmaf(f[w][h], w < nws ? a[w][h] : 0.f, o);
        part[(size_t)h * p.Rv + sl * ncw * CW + c] = o;
      }
    }
    if (tid == 0) {
#pragma unroll
      for (int h = 0; h < GS; ++h) {
        ml[2 * h] = mr[h];
        ml[2 * h + 1] = fs[h];
      }
    }
  }
  if (stamps && lane == 0) {           // timeline dump behind the workspace proper (tools/time_pvq.py allocates it)
    tstamp[4] = wall_clock64();
    long long* dbg = reinterpret_cast<long long*>(p.ml + (size_t)p.G * GS * p.nsplit * 2) + ((size_t)blockIdx.x * NW + wv) * 5;
#pragma unroll
    for (int t = 0; t < 5; ++t) dbg[t] = tstamp[t];
  }
}

struct CombineParams {
  const float* part;
  const float* ml;
  h16* ctx;        // [H, Rv] (row stride ctx_ld)
  float* stats;    // [H][2] global (max, sum) -- consumed by the probs kernel
  int G, gs, Rv, nsplit;
  int ctx_ld;      // elements between the context rows of consecutive heads (Rv unless a column group is written, _qg)
};


// grid H * ceil(Rv / CL) workgroups of 512 threads: the 8 waves share the splits of one head for CL context columns
// (8 independent loads in flight per lane: the merge is latency-, not bandwidth-bound), LDS sum at the end.  CL = 64: one
// column per lane.  CL = 16 (few heads -- a one-group shard of an 8-GPU head-group sharding has H = 4, i.e. 24 workgroups
// at 64 columns each for up to ~1000 ranges): the four 16-lane groups of a wave take every fourth batch of splits, four
// times the workgroups and four times the loads in flight per column.
constexpr int CB_WAVES = 8;
template <int CL>
__global__ __launch_bounds__(64 * CB_WAVES) void pv_combine_kernel(CombineParams p) {
  constexpr int NS = 64 / CL;                       // split streams per wave
  __shared__ float red[CB_WAVES][64];
  __shared__ float redt[CB_WAVES][NS];
  // 1-D grid; workgroup b -> (group, head in group, column block) with group = b % G:
  // the partials of latent group g were written by workgroups with id % G == g (pv_partial*, decode_fused), i.e. with
  // G = 8 on XCD g -- the merge reads them from that XCD's L2 instead of from memory (placement: speed only)
  const int g = blockIdx.x % p.G;
  const int rest = blockIdx.x / p.G;
  const int hh = rest % p.gs, yb = rest / p.gs;
  const int h = g * p.gs + hh;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cl = lane % CL, sub = lane / CL;
  const int r = yb * CL + cl;
  const float* ml = p.ml + ((size_t)g * p.nsplit * p.gs + hh) * 2;
  const size_t ml_stride = (size_t)p.gs * 2;
  const float* part = p.part + ((size_t)g * p.nsplit * p.gs + hh) * p.Rv + min(r, p.Rv - 1);
  const size_t pstride = (size_t)p.gs * p.Rv;
  float acc = 0.f, tot = 0.f;
  constexpr int UN = 16;
  // first batch of this stream's splits is requested BEFORE the global maximum is reduced: the partial rows do not depend
  // on it, and the merge is one memory round trip instead of two (it is latency-bound: 4.7 us for ~1.5 MB)
  float m[UN], sm[UN], pv[UN];
  auto load_batch = [&](int s0) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int s = min(s0 + u, p.nsplit - 1);
      m[u] = ml[s * ml_stride];
      sm[u] = ml[s * ml_stride + 1];
      pv[u] = part[s * pstride];
    }
  };
  const int stream = wv * NS + sub;
  load_batch(stream * UN);
  // global max over the splits (every wave computes it: nsplit is small)
  float M = -INFINITY;
  // (four loads in flight per lane: with one latent group per launch there are ~1000 splits, 16 dependent round trips otherwise)
  for (int s = lane; s < p.nsplit; s += 256) {
    const float m0 = ml[s * ml_stride], m1 = ml[min(s + 64, p.nsplit - 1) * ml_stride];
    const float m2 = ml[min(s + 128, p.nsplit - 1) * ml_stride], m3 = ml[min(s + 192, p.nsplit - 1) * ml_stride];
    M = fmaxf(fmaxf(M, fmaxf(m0, m1)), fmaxf(m2, m3));
  }
  M = wave_max(M);
  for (int s0 = stream * UN; s0 < p.nsplit; s0 += CB_WAVES * NS * UN) {
    if (s0 != stream * UN) load_batch(s0);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float wgt = (s0 + u < p.nsplit && m[u] != -INFINITY) ? __expf(m[u] - M) : 0.f;
      acc = fmaf(wgt, pv[u], acc);
      tot = fmaf(wgt, sm[u], tot);
    }
  }
  red[wv][lane] = acc;
  if (cl == 0) redt[wv][sub] = tot;
  __syncthreads();
  if (wv == 0 && lane < CL) {
    float a = 0.f, t = 0.f;
#pragma unroll
    for (int k = 0; k < CB_WAVES; ++k)
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        a += red[k][q * CL + lane];
        t += redt[k][q];
      }
    if (yb == 0 && lane == 0) {
      p.stats[2 * h] = M;
      p.stats[2 * h + 1] = t;
    }
    if (r < p.Rv) p.ctx[(size_t)h * p.ctx_ld + r] = (h16)(a / t);
  }
}

static void pv_combine_dispatch(const CombineParams& c, int H, int Rv, hipStream_t s) {
  // few heads (one or two latent groups per launch): 16 columns per workgroup, four split streams per wave
  if ((int64_t)H * ((Rv + 63) / 64) < 96)
    hipLaunchKernelGGL(pv_combine_kernel<16>, dim3(H * ((Rv + 15) / 16)), dim3(64 * CB_WAVES), 0, s, c);
  else
    hipLaunchKernelGGL(pv_combine_kernel<64>, dim3(H * ((Rv + 63) / 64)), dim3(64 * CB_WAVES), 0, s, c);
}

// attention weights: softmax(x, fp32).to(fp16)  (palu_attention.py:238)
__global__ void probs_kernel(const h16* scores, int64_t ss_h, const h16* mask, const float* stats, h16* probs,
                             int64_t sp_h, int L, float inv_scale) {
  const int h = blockIdx.y;
  const float M = stats[2 * h], S = stats[2 * h + 1];
  for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x) {
    float x = scaled_logit(scores[h * ss_h + l], inv_scale, mask, l);
    probs[h * sp_h + l] = (h16)(__expf(x - M) / S);
  }
}

int pv_wgs_per_cu() {
  static int per_cu = 0;
  if (per_cu == 0) {
    const char* e = palu_exp_env("PALU_PV_WGS_PER_CU");
    per_cu = e ? atoi(e) : 4;
    if (per_cu < 1) per_cu = 1;
  }
  return per_cu;
}

// split target: ~PALU_PV_WGS_PER_CU (default 4) workgroups per CU in flight, whole multiples of the CU count
// (every CU gets the same number of workgroups: no straggler round)
// 1 / d if  q0 = x * r,  q1 = fma(fma(-q0, d, x), r, q0)  equals the IEEE quotient x / d for every finite fp16 x, else 0
// (checked once per divisor on the host, all 63488 inputs)
float pv_exact_rcp(float d) {
  static thread_local float last_d = 0.f, last_r = 0.f;   // per thread: a consistent (divisor, result) pair without a lock
  if (d == last_d) return last_r;
  const float r = 1.0f / d;
  bool ok = d > 0.f;
  for (unsigned bits = 0; ok && bits < 65536u; ++bits) {
    const unsigned short hb = (unsigned short)bits;
    _Float16 xh;
    memcpy(&xh, &hb, 2);
    const float x = (float)xh;
    if (!(x - x == 0.f)) continue;                       // inf / nan
    const float q0 = x * r;
    const float q1 = fmaf(fmaf(-q0, d, x), r, q0);
    const float ref = x / d;
    if (memcmp(&q1, &ref, 4) != 0 && !(q1 == 0.f && ref == 0.f)) ok = false;
  }
  last_d = d;
  last_r = ok ? r : 0.f;
  return last_r;
}

int pv_qr_wgs() {
  static int wgs = 0;
  if (wgs == 0) {
    const char* e = palu_exp_env("PALU_PVQ_WGS");
    wgs = e ? atoi(e) : 1;
    if (wgs < 1) wgs = 1;
  }
  return wgs;
}

long long pv_split_target(int G) {
  long long t = ((long long)pv_wgs_per_cu() * palu_num_cus() + G - 1) / G;
  return t < 1 ? 1 : t;
}

int pv_rows_per_split(int G, int L) {
  // 8-row granularity; at most 2048 rows (LDS), at least 128
  const long long nsplit = pv_split_target(G);
  long long rps = (L + nsplit - 1) / nsplit;
  rps = (rps + 7) / 8 * 8;
  if (rps < 128) rps = 128;
  if (rps > 2048) rps = 2048;
  return (int)rps;
}

// Upper bound of the split count over EVERY L <= Lcap.  ceil(L / rps(L)) is not monotone in L (rps is rounded up to
// 8 and clamped), so a workspace sized from nsplit(Lcap) can be too small for a shorter fill level: with
// rps = clamp(round8(ceil(L/T)), 128, 2048) the count is <= min(T, ceil(L/128)) while rps < 2048 and ceil(L/2048) after.
// The fused decode kernel (decode_fused.hip) splits into min(CUs / G, ceil(L/64)) ranges: covered by min(T, ceil(L/64)).
int pv_nsplit_bound(int G, int Lcap, int Rv) {
  const long long T = pv_split_target(G);
  long long a = ((long long)Lcap + 63) / 64;
  if (a > T) a = T;
  const long long b = ((long long)Lcap + 2047) / 2048;
  long long r = a > b ? a : b;
  // the register-direct quantised kernel (pv_partial_qr_kernel): one round of Pr = CUs / G ranges, each at least one
  // unit (32 rows) per wave of a slice
  // (the plain-fp16 form of that kernel has 8-column chunks, i.e. the most column slices and the fewest waves per range:
  //  the bound is taken for it)
  int nsl = 1;
  while (nsl * 16 < Rv / 8 && nsl < 8) nsl *= 2;
  const long long nws = 8 / nsl;
  const long long Pr = ((long long)pv_qr_wgs() * palu_num_cus() + G - 1) / G;
  long long c = ((long long)Lcap + 32 * nws - 1) / (32 * nws);
  if (c > Pr) c = Pr;
  return (int)(r > c ? r : c);
}

}  // namespace

// split merge shared with the fused decode kernel (decode_fused.hip): ws = stats [H][2] (padded) | part [H][ns][Rv] | ml [H][ns][2]
int palu_pv_combine_launch(float* ws, void* ctx, int H, int G, int Rv, int ns, hipStream_t s, int ctx_ld) {
  CombineParams c;
  c.part = ws + pv_ws_stats_floats(H);
  c.ml = c.part + (size_t)H * ns * Rv;
  c.ctx = (h16*)ctx;
  c.stats = ws;
  c.G = G; c.gs = H / G; c.Rv = Rv; c.nsplit = ns;
  c.ctx_ld = ctx_ld > 0 ? ctx_ld : Rv;
  pv_combine_dispatch(c, H, Rv, s);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

extern "C" int palu_pv_nsplit(int G, int L) {
  if (L <= 0 || G <= 0) return 0;
  int rps = pv_rows_per_split(G, L);
  return (L + rps - 1) / rps;
}

extern "C" size_t palu_pv_workspace_bytes(int H, int G, int L, int Rv) {
  if (L <= 0 || G <= 0) return 0;
  // part [H][ns][Rv] + ml [H][ns][2] + stats [H][2], fp32, for the largest split count any L' <= L can produce
  // (this kernel's and the fused decode kernel's, which never exceeds CUs / G <= the split target)
  const int ns = pv_nsplit_bound(G, L, Rv);
  return ((size_t)H * ns * (Rv + 2) + pv_ws_stats_floats(H)) * sizeof(float);
}

static int pv_qr_plan(int G, int L, int Rv, int cw, int* nsl_o, int* ncw_o, int* S_o, int* rps_o);
static int pv_qr_chunk_width(int Rv, int bits);
// split count of the register-direct kernel for (G, L, Rv, bits = 3 / 4 / 16): what palu_pv_workspace_bytes has to cover
// (introspection for tests and tools; 0 for shapes that kernel does not take)
extern "C" int palu_pv_direct_nsplit(int G, int L, int Rv, int bits) {
  if (G <= 0 || L <= 0 || Rv <= 0 || !(bits == 3 || bits == 4 || bits == 16)) return 0;
  const int cw = pv_qr_chunk_width(Rv, bits);
  if (cw == 0) return 0;
  int nsl, ncw, S, rps;
  return pv_qr_plan(G, L, Rv, cw, &nsl, &ncw, &S, &rps);
}

extern "C" size_t palu_pv_stats_offset(int H, int G, int L, int Rv) {
  (void)H; (void)G; (void)L; (void)Rv;
  return 0;     // the global (max, sum) pairs lead the workspace, whichever kernel and split count ran
}


// The register-direct matrix-core kernel for packed (bits = 3, 4) or plain fp16 (bits = 16) latent rows: gs in {1,2,4},
// whole 32-code (8-column) chunks, dword (16-byte) aligned rows, a group's rows within 2 GiB, and a divisor whose fast
// quotient is exact.  Returns PV_QR_NOT_TAKEN when the shape is left to the older kernels.
constexpr int PV_QR_NOT_TAKEN = 1;
#define PV_DIRECT_AUTO(G, L) ((G) == 1)
// geometry of a register-direct launch: column slices of <= 16 chunks (cw columns each), S row sets per unit, and one
// round of 8-wave workgroups (256 VGPRs per wave: one workgroup per CU): CUs / G ranges per group, each cut into
// nws = 8 / slices wave ranges of whole units (32 S rows); the statistics are computed online, so a wave range may have
// any number of rows.  Returns the number of ranges.
// columns per chunk of a register-direct launch: 8 (fp16), 32 or 24 (packed) -- whichever fills more of the 16 MFMA lanes
// (row sets x chunks of a slice) -- or 0 when the kernel does not take the shape
static int pv_qr_lanes(int Rv, int cw) {
  const int nch = Rv / cw;
  int nsl = 1;
  while (nsl * 16 < nch) nsl *= 2;
  const int ncw = (nch + nsl - 1) / nsl;
  int S = 16 / ncw;
  if (S > 2) S = 2;
  return S * ncw;
}
static int pv_qr_chunk_width(int Rv, int bits) {
  if (bits == 16) return (Rv % 8 == 0 && Rv / 8 <= 128) ? 8 : 0;
  const bool ok32 = Rv % 32 == 0 && Rv / 32 <= 128, ok24 = bits == 4 && Rv % 24 == 0 && Rv / 24 <= 128;
  if (ok24 && (!ok32 || pv_qr_lanes(Rv, 24) > pv_qr_lanes(Rv, 32))) return 24;
  return ok32 ? 32 : 0;
}
static int pv_qr_plan(int G, int L, int Rv, int cw, int* nsl_o, int* ncw_o, int* S_o, int* rps_o) {
  const int nch = Rv / cw;
  int nsl = 1;
  while (nsl * 16 < nch) nsl *= 2;                      // 1, 2, 4, 8 column slices of <= 16 chunks
  const int ncw = (nch + nsl - 1) / nsl;
  int S = 16 / ncw;
  if (S > 2) S = 2;                                     // (a unit is at most one 64-row batch of statistics)
  const int RU = 32 * S, nws = 8 / nsl;
  const long long P = ((long long)pv_qr_wgs() * palu_num_cus() * nws + G - 1) / G;
  long long rpw = (L + P - 1) / P;
  rpw = (rpw + RU - 1) / RU * RU;
  const int rps = (int)rpw * nws;
  *nsl_o = nsl; *ncw_o = ncw; *S_o = S; *rps_o = rps;
  return (L + rps - 1) / rps;
}
static int pv_qr_launch(const void* scores, int64_t ss_h, const void* mask, const void* rows, int64_t sc_g, int64_t sc_l,
                        const void* meta, int64_t sm_g, int64_t sm_l, void* ctx, void* probs, int64_t sp_h,
                        void* workspace, int H, int G, int L, int Rv, int bits, float sqrt_d, hipStream_t s, int ctx_ld = 0) {
  static int qr_enabled = -1, qr16_enabled = -1;
  if (qr_enabled < 0) {
    const char* e = getenv("PALU_PVQ_DIRECT");
    qr_enabled = e ? atoi(e) : 1;
    // plain fp16 rows through the same kernel: correct (the fp16 parity tests pass with it) and 2 % faster back to back
    // (76.1 vs 78.7 us at C2), but 2.5 us slower behind the score kernel inside the step (bench.py, same box:
    // 182.6 vs 177.4 us per step) -- both kernels stream at the ~6.3 TB/s this part sustains, the VALU kernel's four
    // small workgroups per CU ramp up and drain better.  Opt-in: PALU_PV_DIRECT=1.
    const char* e16 = getenv("PALU_PV_DIRECT");
    qr16_enabled = e16 ? atoi(e16) : 2;                 // 2 = by shape (below)
  }
  const int gs = H / G;
  int cw = pv_qr_chunk_width(Rv, bits);
  const bool al16 = ((uintptr_t)rows & 15) == 0 && sc_g % 16 == 0 && sc_l % 16 == 0;
  if (cw == 32 && bits == 4 && !al16) cw = 0;                          // 16-byte chunks are loaded as aligned dwordx4
  const long long row_bytes = (long long)Rv * bits / 8;
  const long long span = (long long)(L - 1) * sc_l + row_bytes;
  // fp16 rows, no setting: ONE latent group per launch (a rank of the head-group sharding, BASELINE config 5) takes this kernel --
  // 40.4 vs 49.6 us at 256 k positions (profiles/r06_pv_forms.txt): the VALU kernel's ~1000 workgroups then all walk one group's rows
  // in 264-row ranges, this one runs a single round of 256 workgroups with 8 wave ranges each
  const bool on16 = qr16_enabled == 2 ? PV_DIRECT_AUTO(G, L) : qr16_enabled != 0;
  bool ok = (bits == 16 ? on16 : qr_enabled != 0) && (gs == 1 || gs == 2 || gs == 4) && cw != 0 &&
            span + 4096ll * sc_l < 0x7FFFFFFFll && ((uintptr_t)rows & 3) == 0 && sc_g % 4 == 0 && sc_l % 4 == 0 &&
            (bits != 16 || al16) && pv_exact_rcp(sqrt_d) != 0.f;
  if (!ok) return PV_QR_NOT_TAKEN;
  int nsl, ncw, S, rps;
  const int ns = pv_qr_plan(G, L, Rv, cw, &nsl, &ncw, &S, &rps);   // ranges (one workgroup each) = splits seen by pv_combine
  if (ns > pv_nsplit_bound(G, L, Rv)) return PV_QR_NOT_TAKEN;   // (cannot happen for a workspace sized by palu_pv_workspace_bytes)
  float* ws = (float*)workspace;
  PvQParams p;
  p.scores = (const h16*)scores; p.ss_h = ss_h; p.mask = (const h16*)mask;
  p.codes = (const unsigned char*)rows; p.sc_g = sc_g; p.sc_l = sc_l;
  p.meta = (const h16*)meta; p.sm_g = sm_g; p.sm_l = sm_l;
  p.part = ws + pv_ws_stats_floats(H);
  p.ml = p.part + (size_t)H * ns * Rv;
  float* stats = ws;
  p.G = G; p.gs = gs; p.L = L; p.Rv = Rv; p.nsplit = ns; p.rps = rps;
  p.inv_scale = sqrt_d;
  {
    static int ex = -1;
    if (ex < 0) {
      // PALU_PVQ_TIMELINE_DUMP=<bytes the caller added behind the workspace>: every wave dumps 5 wall-clock stamps there
      // (tools/time_pvq.py).  The explicit size keeps an accidental setting from writing past a normal workspace.
      const char* e = palu_exp_env("PALU_PVQ_TIMELINE_DUMP");
      ex = (e && atoll(e) >= (long long)G * ns * 8 * 5 * 8) ? 8 : 0;
    }
    p.exp_flags = ex;
  }
  p.qr_nsl = nsl; p.qr_ncw = ncw; p.qr_s = S;
  p.rcp_scale = pv_exact_rcp(sqrt_d);
  {
    static int prio = -1;
    if (prio < 0) {
      const char* e = palu_exp_env("PALU_PVQ_PRIO");
      prio = e ? atoi(e) : 1;
    }
    p.qr_prio = prio;
  }
  const int njp = cw + 1;
  size_t ldsr = (size_t)8 * 4 * (64 + 8) * sizeof(h16) + (size_t)8 * 16 * njp * 4 * sizeof(float) +
                (size_t)8 * gs * 4 * sizeof(float);
  ldsr = (ldsr + 15) / 16 * 16;
  p.qr_park_off = (unsigned)ldsr;
  ldsr += (size_t)8 * 3 * 64 * 4 * sizeof(float);
  dim3 gridr(G * ns), blockr(512);
#define PALU_PVQR2(GSV, BV, CWV)                                                                            \
  {                                                                                                         \
    if (S == 1) hipLaunchKernelGGL((pv_partial_qr_kernel<GSV, BV, 1, CWV>), gridr, blockr, ldsr, s, p);     \
    else hipLaunchKernelGGL((pv_partial_qr_kernel<GSV, BV, 2, CWV>), gridr, blockr, ldsr, s, p);            \
  }
#define PALU_PVQR(GSV)                                      \
  {                                                         \
    if (bits == 16) PALU_PVQR2(GSV, 16, 32)                 \
    else if (bits == 4 && cw == 24) PALU_PVQR2(GSV, 4, 24)  \
    else if (bits == 4) PALU_PVQR2(GSV, 4, 32)              \
    else PALU_PVQR2(GSV, 3, 32)                             \
  }
  switch (gs) {
    case 1: PALU_PVQR(1) break;
    case 2: PALU_PVQR(2) break;
    default: PALU_PVQR(4) break;
  }
#undef PALU_PVQR
#undef PALU_PVQR2
  PALU_LAUNCH_CHECK();
  int rcq = palu_pv_combine_launch(ws, ctx, H, G, Rv, ns, s, ctx_ld);
  if (rcq) return rcq;
  if (probs) {
    int bx = (L + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(probs_kernel, dim3(bx, H), dim3(256), 0, s, (const h16*)scores, ss_h, (const h16*)mask,
                       (const float*)stats, (h16*)probs, sp_h, L, sqrt_d);
    PALU_LAUNCH_CHECK();
  }
  return PALU_OK;
}

extern "C" int palu_softmax_pv_f16(const void* scores, int64_t ss_h, const void* mask, const void* v, int64_t sv_g,
                                   int64_t sv_l, void* ctx, void* probs, int64_t sp_h, void* workspace, int H, int G,
                                   int L, int Rv, float sqrt_d, palu_stream_t stream) {
  const int ctx_ld = 0;   // context rows are Rv apart
  PALU_REQUIRE(H > 0 && G > 0 && H % G == 0 && L > 0 && Rv > 0, PALU_ERR_ARG, "softmax_pv: bad shape");
  PALU_REQUIRE(scores && v && ctx && workspace, PALU_ERR_ARG, "softmax_pv: null pointer");
  const int gs = H / G;
  PALU_REQUIRE(gs == 1 || gs == 2 || gs == 3 || gs == 4 || gs == 8, PALU_ERR_UNSUPPORTED,
               "softmax_pv: group size %d not supported (1,2,3,4,8)", gs);
  PALU_REQUIRE(Rv % 8 == 0 && Rv / 8 <= PV_THREADS, PALU_ERR_UNSUPPORTED, "softmax_pv: Rv must be a multiple of 8, <= 2048");
  PALU_REQUIRE(((uintptr_t)v & 15) == 0 && sv_g % 8 == 0 && sv_l % 8 == 0 && sv_l >= Rv, PALU_ERR_ARG,
               "softmax_pv: v rows must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  // matrix-core streaming kernel first (pv_partial_qr_kernel with plain fp16 rows); the VALU kernel below takes gs = 8 etc.
  {
    const int rcq = pv_qr_launch(scores, ss_h, mask, v, sv_g * 2, sv_l * 2, nullptr, 0, 0, ctx, probs, sp_h, workspace, H, G, L,
                                 Rv, 16, sqrt_d, s);
    if (rcq != PV_QR_NOT_TAKEN) return rcq;
  }
  const int rps = pv_rows_per_split(G, L);
  const int ns = (L + rps - 1) / rps;
  float* ws = (float*)workspace;
  PvParams p;
  p.scores = (const h16*)scores; p.ss_h = ss_h; p.mask = (const h16*)mask;
  p.v = (const h16*)v; p.sv_g = sv_g; p.sv_l = sv_l;
  p.part = ws + pv_ws_stats_floats(H);
  p.ml = p.part + (size_t)H * ns * Rv;
  float* stats = ws;
  p.G = G; p.gs = gs; p.L = L; p.Rv = Rv; p.nsplit = ns; p.rps = rps;
  p.inv_scale = sqrt_d;
  size_t lds = (size_t)gs * rps * sizeof(float);
  size_t lds_red = (size_t)(PV_THREADS / (Rv / 8)) * Rv * sizeof(float);
  if (lds_red > lds) lds = lds_red;
  PALU_REQUIRE(lds <= 64 * 1024, PALU_ERR_UNSUPPORTED, "softmax_pv: LDS budget exceeded");
  dim3 grid(G * ns), block(PV_THREADS);
  switch (gs) {
    case 1: hipLaunchKernelGGL(pv_partial_kernel<1>, grid, block, lds, s, p); break;
    case 2: hipLaunchKernelGGL(pv_partial_kernel<2>, grid, block, lds, s, p); break;
    case 3: hipLaunchKernelGGL(pv_partial_kernel<3>, grid, block, lds, s, p); break;   // what the single-kernel core and abx take too
    case 4: hipLaunchKernelGGL(pv_partial_kernel<4>, grid, block, lds, s, p); break;
    default: hipLaunchKernelGGL(pv_partial_kernel<8>, grid, block, lds, s, p); break;
  }
  PALU_LAUNCH_CHECK();
  CombineParams c;
  c.part = p.part; c.ml = p.ml; c.ctx = (h16*)ctx; c.stats = stats;
  c.G = G; c.gs = gs; c.Rv = Rv; c.nsplit = ns;
  c.ctx_ld = ctx_ld > 0 ? ctx_ld : Rv;
  pv_combine_dispatch(c, H, Rv, s);
  PALU_LAUNCH_CHECK();
  if (probs) {
    int bx = (L + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(probs_kernel, dim3(bx, H), dim3(256), 0, s, (const h16*)scores, ss_h, (const h16*)mask,
                       (const float*)stats, (h16*)probs, sp_h, L, sqrt_d);
    PALU_LAUNCH_CHECK();
  }
  return PALU_OK;
}

// ctx_ld: elements between the context rows of consecutive heads (0 = Rv; palu_softmax_pv_qg writes column groups in place)
static int softmax_pv_q_impl(const void* scores, int64_t ss_h, const void* mask, const void* codes, int64_t sc_g,
                             int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l, void* ctx, void* probs,
                             int64_t sp_h, void* workspace, int H, int G, int L, int Rv, int bits, float sqrt_d,
                             palu_stream_t stream, int ctx_ld);

extern "C" int palu_softmax_pv_q(const void* scores, int64_t ss_h, const void* mask, const void* codes, int64_t sc_g,
                                 int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l, void* ctx, void* probs,
                                 int64_t sp_h, void* workspace, int H, int G, int L, int Rv, int bits, float sqrt_d,
                                 palu_stream_t stream) {
  return softmax_pv_q_impl(scores, ss_h, mask, codes, sc_g, sc_l, meta, sm_g, sm_l, ctx, probs, sp_h, workspace, H, G, L, Rv,
                           bits, sqrt_d, stream, 0);
}

static int softmax_pv_q_impl(const void* scores, int64_t ss_h, const void* mask, const void* codes, int64_t sc_g,
                             int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l, void* ctx, void* probs,
                             int64_t sp_h, void* workspace, int H, int G, int L, int Rv, int bits, float sqrt_d,
                             palu_stream_t stream, int ctx_ld) {
  PALU_REQUIRE(H > 0 && G > 0 && H % G == 0 && L > 0 && Rv > 0, PALU_ERR_ARG, "softmax_pv_q: bad shape");
  PALU_REQUIRE(scores && codes && meta && ctx && workspace, PALU_ERR_ARG, "softmax_pv_q: null pointer");
  PALU_REQUIRE(bits == 3 || bits == 4, PALU_ERR_UNSUPPORTED, "softmax_pv_q: bits must be 3 or 4");
  const int gs = H / G;
  PALU_REQUIRE(gs == 1 || gs == 2 || gs == 3 || gs == 4 || gs == 8, PALU_ERR_UNSUPPORTED,
               "softmax_pv_q: group size %d not supported (1,2,3,4,8)", gs);
  PALU_REQUIRE(Rv % 32 == 0 && Rv / 16 <= PV_THREADS, PALU_ERR_UNSUPPORTED, "softmax_pv_q: Rv must be a multiple of 32, <= 4096");
  PALU_REQUIRE(((uintptr_t)codes & 3) == 0 && sc_g % 4 == 0 && sc_l % 4 == 0 && ((uintptr_t)meta & 3) == 0 &&
                   sm_g % 2 == 0 && sm_l % 2 == 0,
               PALU_ERR_ARG, "softmax_pv_q: packed rows / meta must be 4-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  int rps = pv_rows_per_split(G, L);
  // register-direct matrix-core kernel first (pv_partial_qr_kernel); the VALU kernel below takes what it does not
  {
    const int rcq = pv_qr_launch(scores, ss_h, mask, codes, sc_g, sc_l, meta, sm_g, sm_l, ctx, probs, sp_h, workspace, H, G, L,
                                 Rv, bits, sqrt_d, s, ctx_ld);
    if (rcq != PV_QR_NOT_TAKEN) return rcq;
  }
  const int ns = (L + rps - 1) / rps;
  float* ws = (float*)workspace;
  PvQParams p;
  p.scores = (const h16*)scores; p.ss_h = ss_h; p.mask = (const h16*)mask;
  p.codes = (const unsigned char*)codes; p.sc_g = sc_g; p.sc_l = sc_l;
  p.meta = (const h16*)meta; p.sm_g = sm_g; p.sm_l = sm_l;
  p.part = ws + pv_ws_stats_floats(H);
  p.ml = p.part + (size_t)H * ns * Rv;
  float* stats = ws;
  p.G = G; p.gs = gs; p.L = L; p.Rv = Rv; p.nsplit = ns; p.rps = rps;
  p.inv_scale = sqrt_d;
  p.exp_flags = 0;
  size_t lds = (size_t)gs * rps * (sizeof(float) + sizeof(h16));
  if (lds < (size_t)PV_THREADS * 16 * sizeof(float)) lds = (size_t)PV_THREADS * 16 * sizeof(float);
  PALU_REQUIRE(lds <= 160 * 1024, PALU_ERR_UNSUPPORTED, "softmax_pv_q: LDS budget exceeded");
  dim3 grid(G * ns), block(PV_THREADS);
  // gs = 3 / 8 (query heads per latent group of GQA models: kv group size x n_rep) exist only in this VALU kernel; the
  // register-direct kernel above takes gs in {1, 2, 4}
#define PALU_PVQ1(GSV, BV)                                                                          \
  {                                                                                                 \
    auto kern = pv_partial_q_kernel<GSV, BV>;                                                       \
    const int rca = lds > 64 * 1024 ? palu_func_max_lds(reinterpret_cast<const void*>(kern), (int)lds) : PALU_OK; \
    if (rca) return rca;                                                                            \
    hipLaunchKernelGGL(kern, grid, block, lds, s, p);                                               \
  }
#define PALU_PVQ(GSV) \
  if (bits == 4) PALU_PVQ1(GSV, 4) else PALU_PVQ1(GSV, 3)
  switch (gs) {
    case 1: PALU_PVQ(1); break;
    case 2: PALU_PVQ(2); break;
    case 3: PALU_PVQ(3); break;
    case 8: PALU_PVQ(8); break;
    default: PALU_PVQ(4); break;
  }
#undef PALU_PVQ
#undef PALU_PVQ1
  PALU_LAUNCH_CHECK();
  CombineParams c;
  c.part = p.part; c.ml = p.ml; c.ctx = (h16*)ctx; c.stats = stats;
  c.G = G; c.gs = gs; c.Rv = Rv; c.nsplit = ns;
  c.ctx_ld = ctx_ld > 0 ? ctx_ld : Rv;
  pv_combine_dispatch(c, H, Rv, s);
  PALU_LAUNCH_CHECK();
  if (probs) {
    int bx = (L + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(probs_kernel, dim3(bx, H), dim3(256), 0, s, (const h16*)scores, ss_h, (const h16*)mask,
                       (const float*)stats, (h16*)probs, sp_h, L, sqrt_d);
    PALU_LAUNCH_CHECK();
  }
  return PALU_OK;
}

// Quantised latents with per-COLUMN-GROUP (scale, zero) pairs -- quantize_tensor(..., group_size > 0), quant.py:11-13, the
// --lt_group_size option: a row of Rv codes carries Rv / group_size metas.  The packed code stream of such a row is the
// concatenation of its groups' streams (group_size % 8 == 0), i.e. identical to the whole-row layout; only the meta differs:
// meta [G, L, Rv / group_size, 2].  Each column group is a P.V problem of width group_size with its own row scales: the
// kernels above run once per group on offset pointers and write their [H, group_size] slice of ctx in place.
extern "C" int palu_softmax_pv_qg(const void* scores, int64_t ss_h, const void* mask, const void* codes, int64_t sc_g,
                                  int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l, void* ctx, void* probs,
                                  int64_t sp_h, void* workspace, int H, int G, int L, int Rv, int bits, int group_size,
                                  float sqrt_d, palu_stream_t stream) {
  if (group_size <= 0 || group_size == Rv)
    return palu_softmax_pv_q(scores, ss_h, mask, codes, sc_g, sc_l, meta, sm_g, sm_l, ctx, probs, sp_h, workspace, H, G, L, Rv,
                             bits, sqrt_d, stream);
  PALU_REQUIRE(Rv % group_size == 0 && group_size % 32 == 0, PALU_ERR_UNSUPPORTED,
               "softmax_pv_qg: group_size %d must divide Rv %d and be a multiple of 32", group_size, Rv);
  PALU_REQUIRE(bits == 3 || bits == 4, PALU_ERR_UNSUPPORTED, "softmax_pv_qg: bits must be 3 or 4");
  PALU_REQUIRE(sm_l >= 2 * (Rv / group_size), PALU_ERR_ARG, "softmax_pv_qg: meta rows hold Rv / group_size (scale, zero) pairs");
  const int ng = Rv / group_size;
  const int gbytes = group_size * bits / 8;
  int rc = PALU_OK;
  for (int q = 0; q < ng && rc == PALU_OK; ++q)
    rc = softmax_pv_q_impl(scores, ss_h, mask, (const char*)codes + (size_t)q * gbytes, sc_g, sc_l, (const h16*)meta + 2 * q,
                           sm_g, sm_l, (h16*)ctx + (size_t)q * group_size, q == 0 ? probs : nullptr, sp_h, workspace, H, G, L,
                           group_size, bits, sqrt_d, stream, Rv);
  return rc;
}
