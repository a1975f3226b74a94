// abx_rope: fused  K = X.B  ->  RoPE  ->  q.K^T   for the low-rank latent key cache.
//
// Replaces the reference's only GPU kernel, Triton `_abx_fwd` (kernel/abx_rope.py:44-111) and
// its launcher `abx` (:114-150); numerics follow the PyTorch oracle `torch_abx` (:152-171) with
// fp32 kept through RoPE and the q-dot (one fp16 rounding at the store).
//
// MI355X design (not a translation of the Triton tiling):
//   * one 512-thread workgroup (8 waves, 2 per SIMD) per CU, persistent over a contiguous
//     range of 128-row tiles of one latent group g: the X tile is read from HBM exactly once
//     and shared by all heads of the group through LDS;
//   * the reconstruction is a dense [L x R].[R x gs*D] GEMM (arithmetic intensity gs*D = 512
//     flop/byte > machine ridge), so it runs on MFMA: v_mfma_f32_32x32x16_f16 with
//       A = rows of B^T held in REGISTERS for the whole kernel (B is a weight: wave w owns the
//           8 RoPE pairs {8w..8w+7, 64+8w..64+8w+7} of every head of the block), pre-laid-out by
//           abx_prepare_b so the prologue is 16-byte lane-linear loads,
//       B = X rows read from LDS with one ds_read_b128 per 16-deep k-step (XOR-swizzled rows,
//           conflict free), shared by all heads;
//   * the M-rows of each MFMA are ordered (pair, head, half) so that a lane ends up holding
//     k[i] and k[i+64] of 4 RoPE pairs x all heads for ONE position: the rotation coefficients
//     are computed once per (position, pair) and reused by every head, the d-reduction is
//     in-lane, then one cross-half shuffle and a cross-wave LDS sum;
//   * RoPE angles follow the oracle exactly: angle = fl32(l * inv_freq) (kernel/
//     pytorch_reference.py:5-6).  cos/sin of the exact product l*inv_freq are carried by a
//     rotation recurrence (+32 positions per step) and corrected to the fp32-rounded angle by a
//     second-order expansion in the (exactly computed) rounding residual.
#include "palu_common.h"

namespace {

constexpr int TL = 128;        // rows (cache positions) per tile
constexpr int NTHREADS = 512;  // 8 waves
constexpr int HEAD_DIM = 128;

struct AbxParams {
  const h16* a;
  int64_t sa_h, sa_d;
  const u32x4* bfrag;
  const h16* x;
  int64_t sx_g, sx_l;
  h16* out;
  int64_t so_h;
  const float* inv_freq;
  int H, G, gs, HB, L, R, pos0;
  int nch;       // workgroups per (group, head-block)
  int nt_total;  // number of 128-row tiles covering L
  int nkc;       // 128-column chunks of R (chunked kernel only)
};

// heads per workgroup = 2*NMB; each MFMA M-block carries 2 heads x 8 pairs x {i, i+64}
inline int abx_nmb(int gs) { return gs >= 3 ? 2 : 1; }

// ---------------------------------------------------------------------------------------------
// B [H,R,D] -> MFMA A-operand fragments.
// u32x4 index = ((((gb*8 + w)*NMB + mb)*NKS + ks)*64 + lane);  lane = m + 32*hi holds row m of the
// M-block, k = 16*ks + 8*hi .. +7.  Row m  <->  u = m&1 (0: d=i, 1: d=i+64), t = (m>>1)&1 (head
// 2*mb+t of the block), pair = m>>2 (i = 8*w + pair).  Invalid heads / r >= R are zero.
__global__ void abx_prepare_b_kernel(const h16* __restrict__ b, int64_t sb_h, int64_t sb_r, int64_t sb_d,
                                     int H, int G, int R, int nmb, int hb_per_g, int nks,
                                     u32x4* __restrict__ out, int64_t total) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int lane = (int)(idx & 63);
  int64_t t = idx >> 6;
  int ks = (int)(t % nks); t /= nks;
  int mb = (int)(t % nmb); t /= nmb;
  int w = (int)(t % 8); t /= 8;
  int gb = (int)t;
  int g = gb / hb_per_g, hb = gb % hb_per_g;
  int m = lane & 31, hi = lane >> 5;
  int u = m & 1, tt = (m >> 1) & 1, pair = m >> 2;
  int gs = H / G;
  int hloc = hb * (2 * nmb) + 2 * mb + tt;
  bool valid = hloc < gs;
  int h = g * gs + hloc;
  int d = 8 * w + pair + 64 * u;
  h16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int r = 16 * ks + 8 * hi + e;
    v[e] = (valid && r < R) ? b[h * sb_h + r * sb_r + d * sb_d] : (h16)0.f;
  }
  out[idx] = *reinterpret_cast<u32x4*>(&v);
}

// ---------------------------------------------------------------------------------------------
template <int NKS>
struct LdsGeom {
  static constexpr int CPR = 2 * NKS;              // 16-byte chunks per LDS row
  static constexpr int RB = 32 * NKS;              // LDS row bytes (power of two: NKS in {2,4,8})
  static constexpr int TILE_BYTES = TL * RB;
  static constexpr int SPT = TL * CPR / NTHREADS;  // staging slots per thread
  static constexpr int RPB = 256 / RB;             // rows per 256-byte bank row
  static constexpr int SH = (RPB == 4) ? 2 : (RPB == 2 ? 1 : 0);
  static constexpr int MASK = CPR - 1;
  // chunk position of global chunk c in LDS row `row` (XOR swizzle, an involution)
  static __device__ __forceinline__ int swz(int row, int c) { return c ^ ((row >> SH) & MASK); }
};

constexpr int abx_smem_bytes(int nks) { return 2 * TL * 32 * nks + 2 * 8 * 4 * TL * (int)sizeof(float); }

template <int NKS, int NMB, bool CHUNKED>
__global__ __launch_bounds__(NTHREADS, 2) void abx_rope_kernel(AbxParams p) {
  using Geo = LdsGeom<NKS>;
  constexpr int HPW = 2 * NMB;
  constexpr int NACC = CHUNKED ? 4 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem + 2 * Geo::TILE_BYTES);  // [2][8][4][TL]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hi = lane >> 5;

  const int ngb = p.G * p.HB;
  const int gb = blockIdx.x % ngb;
  const int cidx = blockIdx.x / ngb;
  const int g = gb / p.HB, hb = gb % p.HB;

  // contiguous tile range of this workgroup
  const int base = p.nt_total / p.nch, rem = p.nt_total % p.nch;
  const int tile0 = cidx * base + min(cidx, rem);
  const int ntile = base + (cidx < rem ? 1 : 0);
  if (ntile <= 0) return;
  const int NKC = CHUNKED ? p.nkc : 1;
  const int nunit = ntile * NKC;
  const int nks_tot = NKS * NKC;

  const h16* xg = p.x + (int64_t)g * p.sx_g;

  // ---- staging slots of this thread: LDS slot s = tid + 512*k  ->  (row, chunk position)
  int st_row[Geo::SPT], st_col[Geo::SPT];
#pragma unroll
  for (int k = 0; k < Geo::SPT; ++k) {
    int s = tid + NTHREADS * k;
    int row = s / Geo::CPR, pp = s % Geo::CPR;
    st_row[k] = row;
    st_col[k] = Geo::swz(row, pp) * 8;  // global column (elements) inside the 16*NKS-wide chunk
  }
  u32x4 pf[Geo::SPT];
  auto load_unit = [&](int u) {
    int tt = u / NKC, kc = u - tt * NKC;
    int row0 = (tile0 + tt) * TL;
#pragma unroll
    for (int k = 0; k < Geo::SPT; ++k) {
      int l = min(row0 + st_row[k], p.L - 1);
      int col = kc * (16 * NKS) + st_col[k];
      const u32x4* src = reinterpret_cast<const u32x4*>(xg + (int64_t)l * p.sx_l + col);
      if (CHUNKED && col >= p.R) {
        pf[k] = u32x4{0u, 0u, 0u, 0u};
      } else {
        pf[k] = __builtin_nontemporal_load(src);
      }
    }
  };
  auto store_unit = [&](int buf) {
    char* dst = smem + buf * Geo::TILE_BYTES;
#pragma unroll
    for (int k = 0; k < Geo::SPT; ++k)
      *reinterpret_cast<u32x4*>(dst + (size_t)(tid + NTHREADS * k) * 16) = pf[k];
  };

  load_unit(0);

  // ---- B fragments (registers for the whole kernel unless CHUNKED)
  const u32x4* bf_base = p.bfrag + ((int64_t)(gb * 8 + w) * NMB) * nks_tot * 64 + lane;
  h16x8 bf[NMB][NKS];
  if (!CHUNKED) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        u32x4 v = bf_base[(int64_t)(mb * nks_tot + ks) * 64];
        bf[mb][ks] = *reinterpret_cast<h16x8*>(&v);
      }
  }

  // ---- query values of this lane's 4 pairs x HPW heads: (a[h][i], a[h][i+64]), i = 8w + 2j + hi
  float q1[HPW][4], q2[HPW][4];
#pragma unroll
  for (int s = 0; s < HPW; ++s) {
    int hloc = hb * HPW + s;
    bool valid = hloc < p.gs;
    int h = p.G > 0 ? g * p.gs + (valid ? hloc : 0) : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int i = 8 * w + 2 * j + hi;
      float v1 = (float)p.a[h * p.sa_h + i * p.sa_d];
      float v2 = (float)p.a[h * p.sa_h + (i + 64) * p.sa_d];
      q1[s][j] = valid ? v1 : 0.f;
      q2[s][j] = valid ? v2 : 0.f;
    }
  }

  // ---- RoPE state of this lane: position l = pos0 + row0 + n (+32 per block), 4 pairs
  float fr[4], rc[4], rs[4], cs[4], sn[4];
  float lf = (float)(p.pos0 + tile0 * TL + n);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    fr[j] = p.inv_freq[8 * w + 2 * j + hi];
    float ang = lf * fr[j];
    float lo = fmaf(lf, fr[j], -ang);  // exact: l*f = ang + lo
    float so, co;
    sincosf(ang, &so, &co);
    float hh = 0.5f * lo * lo;
    cs[j] = fmaf(-hh, co, fmaf(-lo, so, co));  // cos(ang + lo)
    sn[j] = fmaf(-hh, so, fmaf(lo, co, so));   // sin(ang + lo)
    sincosf(32.0f * fr[j], &rs[j], &rc[j]);
  }

  f32x16 acc[NACC][NMB];

  store_unit(0);
  if (nunit > 1) load_unit(1);

  auto reduce_store = [&](int tt) {
    // thread -> (head slot, position): sum the 8 waves' partials, round once to fp16
    int slot = tid >> 7, pos = tid & 127;
    if (slot < HPW) {
      const float* r = red + (size_t)(tt & 1) * (8 * 4 * TL) + slot * TL + pos;
      float s = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) s += r[ww * 4 * TL];
      int l = (tile0 + tt) * TL + pos;
      int hloc = hb * HPW + slot;
      if (l < p.L && hloc < p.gs) p.out[(int64_t)(g * p.gs + hloc) * p.so_h + l] = (h16)s;
    }
  };

  auto epilogue_block = [&](int tt, int blk, f32x16 (&ac)[NMB]) {
    float part[HPW];
#pragma unroll
    for (int s = 0; s < HPW; ++s) part[s] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // coefficients at the oracle's fp32-rounded angle
      float ang = lf * fr[j];
      float lo = fmaf(lf, fr[j], -ang);  // exact angle = ang + lo  ->  want cos/sin(exact - lo)
      float hh = 0.5f * lo * lo;
      float cc = fmaf(-hh, cs[j], fmaf(lo, sn[j], cs[j]));
      float ss = fmaf(-hh, sn[j], fmaf(-lo, cs[j], sn[j]));
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float k1 = ac[mb][4 * j + 2 * t], k2 = ac[mb][4 * j + 2 * t + 1];
          int s = 2 * mb + t;
          float t1 = fmaf(q2[s][j], k2, q1[s][j] * k1);
          float t2 = fmaf(-q1[s][j], k2, q2[s][j] * k1);
          part[s] = fmaf(cc, t1, fmaf(ss, t2, part[s]));
        }
      // advance the exact-angle state by 32 positions
      float c2 = fmaf(-sn[j], rs[j], cs[j] * rc[j]);
      float s2 = fmaf(cs[j], rs[j], sn[j] * rc[j]);
      cs[j] = c2;
      sn[j] = s2;
    }
    lf += 32.0f;
    float* rdst = red + (size_t)(tt & 1) * (8 * 4 * TL) + (size_t)w * (4 * TL) + blk * 32 + n;
#pragma unroll
    for (int s = 0; s < HPW; ++s) {
      float v = part[s] + __shfl_xor(part[s], 32, 64);
      if (hi == 0) rdst[s * TL] = v;
    }
  };

  for (int u = 0; u < nunit; ++u) {
    const int tt = u / NKC, kc = u - tt * NKC;
    __syncthreads();
    if (u + 1 < nunit) store_unit((u + 1) & 1);
    if (u + 2 < nunit) load_unit(u + 2);
    if (kc == 0 && tt > 0) reduce_store(tt - 1);

    if (CHUNKED) {
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          u32x4 v = bf_base[(int64_t)(mb * nks_tot + kc * NKS + ks) * 64];
          bf[mb][ks] = *reinterpret_cast<h16x8*>(&v);
        }
    }

    const char* xs = smem + (u & 1) * Geo::TILE_BYTES;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const int ai = CHUNKED ? blk : 0;
      if (!CHUNKED || kc == 0) {
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[ai][mb][e] = 0.f;
      }
      const int row = blk * 32 + n;
      const char* xrow = xs + row * Geo::RB;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int c = Geo::swz(row, 2 * ks + hi);
        h16x8 xf = *reinterpret_cast<const h16x8*>(xrow + c * 16);
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
          acc[ai][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[mb][ks], xf, acc[ai][mb], 0, 0, 0);
      }
      if (!CHUNKED) epilogue_block(tt, blk, acc[0]);
    }
    if (CHUNKED && kc == NKC - 1) {
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) epilogue_block(tt, blk, acc[blk]);
    }
  }
  __syncthreads();
  reduce_store(ntile - 1);
}

template <int NKS, int NMB, bool CHUNKED>
int launch_abx(const AbxParams& p, int nwg, hipStream_t stream) {
  auto kern = abx_rope_kernel<NKS, NMB, CHUNKED>;
  constexpr int smem = abx_smem_bytes(NKS);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      palu_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", smem, hipGetErrorString(e));
      return PALU_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), smem, stream, p);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

struct AbxPlan {
  int gs, nmb, hpw, hb, nks_tot, nkc;
  bool chunked;
};

bool abx_plan(int H, int G, int R, AbxPlan* pl) {
  if (H <= 0 || G <= 0 || H % G != 0 || R <= 0 || R % 8 != 0) return false;
  pl->gs = H / G;
  pl->nmb = abx_nmb(pl->gs);
  pl->hpw = 2 * pl->nmb;
  pl->hb = (pl->gs + pl->hpw - 1) / pl->hpw;
  pl->chunked = !(R == 32 || R == 64 || R == 128);
  pl->nkc = pl->chunked ? (R + 127) / 128 : 1;
  pl->nks_tot = pl->chunked ? 8 * pl->nkc : R / 16;
  return true;
}

}  // namespace

extern "C" size_t palu_abx_bfrag_bytes(int H, int G, int R) {
  AbxPlan pl;
  if (!abx_plan(H, G, R, &pl)) return 0;
  return (size_t)G * pl.hb * 8 * pl.nmb * pl.nks_tot * 64 * sizeof(u32x4);
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
  return PALU_OK;
}

extern "C" int palu_abx_rope_f16(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, const void* x,
                                 int64_t sx_g, int64_t sx_l, void* out, int64_t so_h, int H, int G, int L, int R,
                                 int D, const float* inv_freq, int pos0, palu_stream_t stream) {
  AbxPlan pl;
  PALU_REQUIRE(abx_plan(H, G, R, &pl), PALU_ERR_ARG, "abx: bad shape H=%d G=%d R=%d", H, G, R);
  PALU_REQUIRE(D == HEAD_DIM, PALU_ERR_UNSUPPORTED, "abx: head_dim must be 128 (got %d)", D);
  PALU_REQUIRE(L >= 0, PALU_ERR_ARG, "abx: negative L");
  if (L == 0) return PALU_OK;
  PALU_REQUIRE(a && bfrag && x && out && inv_freq, PALU_ERR_ARG, "abx: null pointer");
  PALU_REQUIRE(((uintptr_t)x & 15) == 0 && sx_g % 8 == 0 && sx_l % 8 == 0 && sx_l >= R, PALU_ERR_ARG,
               "abx: x rows must be 16-byte aligned and contiguous (sx_g=%lld sx_l=%lld R=%d)", (long long)sx_g,
               (long long)sx_l, R);
  PALU_REQUIRE(((uintptr_t)bfrag & 15) == 0, PALU_ERR_ARG, "abx: bfrag must be 16-byte aligned");
  PALU_REQUIRE((int64_t)pos0 + L < (1 << 24), PALU_ERR_UNSUPPORTED, "abx: positions must stay below 2^24");

  AbxParams p;
  p.a = (const h16*)a; p.sa_h = sa_h; p.sa_d = sa_d;
  p.bfrag = (const u32x4*)bfrag;
  p.x = (const h16*)x; p.sx_g = sx_g; p.sx_l = sx_l;
  p.out = (h16*)out; p.so_h = so_h;
  p.inv_freq = inv_freq;
  p.H = H; p.G = G; p.gs = pl.gs; p.HB = pl.hb; p.L = L; p.R = R; p.pos0 = pos0;
  p.nt_total = (L + TL - 1) / TL;
  p.nkc = pl.nkc;
  int ngb = G * pl.hb;
  int target = palu_num_cus();               // one 8-wave workgroup per CU
  int nch = target / ngb;
  if (nch < 1) nch = 1;
  if (nch > p.nt_total) nch = p.nt_total;
  p.nch = nch;
  int nwg = nch * ngb;
  hipStream_t s = (hipStream_t)stream;
  if (pl.chunked) {
    return pl.nmb == 2 ? launch_abx<8, 2, true>(p, nwg, s) : launch_abx<8, 1, true>(p, nwg, s);
  }
  switch (R) {
    case 32: return pl.nmb == 2 ? launch_abx<2, 2, false>(p, nwg, s) : launch_abx<2, 1, false>(p, nwg, s);
    case 64: return pl.nmb == 2 ? launch_abx<4, 2, false>(p, nwg, s) : launch_abx<4, 1, false>(p, nwg, s);
    default: return pl.nmb == 2 ? launch_abx<8, 2, false>(p, nwg, s) : launch_abx<8, 1, false>(p, nwg, s);
  }
}
