// abx_rope kernels (shared by abx_rope.hip [fp16 latents] and abx_rope_q.hip [3/4-bit latents])
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
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "abx_fold.h"

// two-band score kernel (abx_rope2.hip): launches it when the shape, the positions and a registered coefficient table
// allow (gs = 4, R in {32, 64, 128}, pos0 % 128 == 0, pos0 + L <= 2^18, low-band angles below 2048 rad); returns
// PALU_ABX2_SKIP when the caller should run abx_rope_kernel instead.  `params` is an AbxParams with bfrag2 set.
#define PALU_ABX2_SKIP 1
int palu_abx2_try_launch(const void* params, int nwg, int bits, hipStream_t stream);
int palu_abx2_try_launch_windows(const void* params, int nwg, int bits, void* scratch, int64_t acc_ld, hipStream_t stream);
size_t palu_abx2_frag_bytes(int H, int G, int R);   // 0 when the shape is not covered
int palu_abx2_prepare_b(const void* b, int64_t sb_h, int64_t sb_r, int64_t sb_d, int H, int G, int R, void* frag2, hipStream_t stream);
// position-split form of the two-band kernel (abx_rope3.hip, abx_rope3_kernel.h): fp16 latents, one launch (no column
// windows); `params` as for palu_abx2_try_launch with the coefficient table filled in, nks = R / 16
int palu_abx3_launch(const void* params, int nks, hipStream_t stream);
// the query fold as its own launch (abx_fold.h): a [H, D] + the two-band fragments of B -> qfold [G][16 nks KB]
int palu_abx3_fold_launch(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag2, void* qfold, int H, int G, int nks,
                          hipStream_t stream);

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
  int nks_frag;  // k-steps per M-block in the fragment buffer (chunked kernel; 0 = 8 * nkc)
  unsigned long long* dbg;  // optional per-wave cycle stamps (timing build only)
  unsigned out_bytes;       // extent of `out` for the bounds-checked buffer store
  int prio_mode;            // 0 none, 1 static (waves 4-7), 2 alternating per half tile
  int exp_flags;            // experiments (timing only): bit0 = waves 4-7 skip the MFMA/epilogue work
  // quantised latents (QBITS > 0): packed codes [G, L, R*bits/8] + (scale, zero) fp16 pairs [G, L, 2]
  const unsigned char* xq;
  int64_t sq_g, sq_l;       // bytes
  const h16* xmeta;
  int64_t sm_g, sm_l;       // elements
  int qgroup;               // columns per (scale, zero) pair: 0 = one pair per row, else R / qgroup pairs (chunked / two-band kernels)
  int qcol0;                // two-band column windows with qgroup > 0: the window's first column of the row (selects the pair)
  int ncols;                // fast fp16 kernel: valid columns of the 16*NKS-column window (0 = all); the rest reads as zero
  // multi-pass use of the fast kernel (ranks above 128: one launch per 128-column window of x, fp32 accumulation):
  int ks0;                  // first fragment k-step of this pass (0)
  int win_pass;             // two-band column windows (abx_rope2.hip): 0 = first window (store), 1 = middle (add), 2 = last (add, round)
  float* acc;               // fp32 scores [H][acc_ld] of the ACC = 1 / 2 instantiations (store / atomic add of the partials)
  int64_t acc_ld;
  // two-band kernel (abx_rope2_kernel.h): its fragment layout (behind the fragments above in the same allocation), the
  // low-band coefficient table (palu_rope_table_build) and the table tile of position pos0
  const u32x4* bfrag2;
  const u32x4* rope_tab;
  int tab_tile0;
  // position-split kernel (abx_rope3_kernel.h): exact-angle RoPE start tables that follow the coefficient tiles in the same
  // table (abx2_rope_start_kernel): rope_t1 [tile][hi 2][q 16] (cos, sin)(128 tile f_i), rope_t2 [n 0..32][hi][q] (cos, sin)(n f_i)
  const float* rope_t1;
  const float* rope_t2;
  // the same kernel on PRE-FOLDED fragments (abx_fold.h; palu_abx_fold_f16 / palu_decode_qkv_fold_f16 wrote them earlier on the
  // stream): [G][16 NKS KB]; null = the kernel folds `a` into bfrag2 in its own prologue
  const u32x4* qfold;
};

// heads per workgroup = 2*NMB; each MFMA M-block carries 2 heads x 8 pairs x {i, i+64}
inline int abx_nmb(int gs) { return gs >= 3 ? 2 : 1; }

// ---------------------------------------------------------------------------------------------
// B [H,R,D] -> MFMA A-operand fragments.
// u32x4 index = ((((gb*8 + w)*NMB + mb)*NKS + ks)*64 + lane);  lane = m + 32*hi holds row m of the
// M-block, k = 16*ks + 8*hi .. +7.  Row m  <->  t = m&1 (head 2*mb+t of the block), u = (m>>1)&1 (0: d=i,
// 1: d=i+64), pair = m>>2 (i = 8*w + pair): after the MFMA a lane holds, per pair, the register quad
// (k_i[h0], k_i[h1], k_{i+64}[h0], k_{i+64}[h1]) -- head pairs adjacent, ready for packed-fp32 math.
// Invalid heads / r >= R are zero.
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
  int tt = m & 1, u = (m >> 1) & 1, pair = m >> 2;
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


// sin/cos of the EXACT product l*f (both fp32 values, product exact in fp64): two-term Cody-Waite
// reduction in fp64 (|n| < 2^24 here), fp32 minimax polynomials on [-pi/4, pi/4] (abs err ~1e-7).
static __device__ __forceinline__ void sincos_exact_product(float l, float f, float* s, float* c) {
  const double x = (double)l * (double)f;
  const double nd = __builtin_rint(x * 0.6366197723675814);
  double rd = __builtin_fma(-nd, 1.5707963267948966, x);
  rd = __builtin_fma(-nd, 6.123233995736766e-17, rd);
  const float r = (float)rd;
  const float r2 = r * r;
  float sp = fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f);
  sp = fmaf(r * r2, sp, r);
  float cp = fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
  cp = fmaf(r2 * r2, cp, fmaf(-0.5f, r2, 1.0f));
  const int q = (int)(long long)nd & 3;
  const float ss = (q & 1) ? cp : sp;
  const float cc = (q & 1) ? sp : cp;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}

constexpr int abx_smem_fast(int nks) { return 3 * TL * 32 * nks + 3 * 8 * 4 * TL * (int)sizeof(float); }
constexpr int abx_smem_bytes(int nks, int nred) { return 2 * TL * 32 * nks + nred * 8 * 4 * TL * (int)sizeof(float); }

// QBITS = 3 / 4: the latent rows are packed codes + (scale, zero) per row (quant.hip layout, like the fast kernel's
// QBITS path): every staging slot (a row's 8 consecutive columns) is dequantised in registers -- (code - zero) exact in
// fp16 via the 1024 + code trick, one fp16 multiply by the scale, i.e. bit-identical to quantize_tensor's output
// (quant.py:39) -- before it goes to the same LDS tile image.  Any R % 32 == 0 (3 bit) / R % 8 == 0 (4 bit): the ranks
// the Fisher rank search emits (palu/rank_search.py:11-17: multiples of 32 per group).
template <int NKS, int NMB, bool CHUNKED, int QBITS = 0>
__global__ __launch_bounds__(NTHREADS, 2) void abx_rope_generic_kernel(AbxParams p) {
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

  const h16* xg = QBITS ? nullptr : p.x + (int64_t)g * p.sx_g;

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
  const unsigned char* xqg = QBITS ? p.xq + (int64_t)g * p.sq_g : nullptr;
  const h16* xmg = QBITS ? p.xmeta + (int64_t)g * p.sm_g : nullptr;
  const int row_bytes_q = QBITS ? p.R * QBITS / 8 : 0;
  auto load_unit = [&](int u) {
    int tt = u / NKC, kc = u - tt * NKC;
    int row0 = (tile0 + tt) * TL;
#pragma unroll
    for (int k = 0; k < Geo::SPT; ++k) {
      int l = min(row0 + st_row[k], p.L - 1);
      int col = kc * (16 * NKS) + st_col[k];
      if (QBITS != 0) {
        // 8 codes of row l starting at column col (a multiple of 8): bits [QBITS * col, +8 * QBITS) of the packed row
        if (col >= p.R) {
          pf[k] = u32x4{0u, 0u, 0u, 0u};
        } else {
          const unsigned char* rowp = xqg + (int64_t)l * p.sq_l;
          unsigned grp;
          if (QBITS == 4) {
            grp = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(rowp + (col >> 1)));
          } else {
            const int bo = 3 * (col >> 3);                       // byte offset of the 24 bits; the row is dword aligned
            const int a = bo & ~3;
            const unsigned w0 = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(rowp + a));
            const unsigned w1 = (a + 4 < row_bytes_q) ? __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(rowp + a + 4)) : 0u;
            const unsigned long long w = ((unsigned long long)w1 << 32) | w0;
            grp = (unsigned)(w >> ((bo & 3) * 8));
          }
          // (scale, zero) of this slot's quantisation group: the whole row, or columns [qgroup * k, qgroup * (k + 1))
          // (quantize_tensor with group_size > 0, quant.py:11-13; qgroup % 8 == 0, so a slot never straddles two groups)
          const unsigned meta = *reinterpret_cast<const unsigned*>(xmg + (int64_t)l * p.sm_l + (p.qgroup > 0 ? 2 * (col / p.qgroup) : 0));
          const h16x2 m2 = __builtin_bit_cast(h16x2, meta);
          const h16x2 scale2 = h16x2{m2[0], m2[0]};
          const h16 nb = -((h16)1024.f + m2[1]);                 // exact: zero is an integer in [0, 15]
          const h16x2 negbias2 = h16x2{nb, nb};
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned c0 = (grp >> (QBITS * (2 * e))) & ((1u << QBITS) - 1);
            const unsigned c1 = (grp >> (QBITS * (2 * e + 1))) & ((1u << QBITS) - 1);
            const unsigned pw = 0x64006400u | c0 | (c1 << 16);   // (1024 + c0, 1024 + c1) as fp16
            h16x2 v = __builtin_bit_cast(h16x2, pw);
            v = (v + negbias2) * scale2;
            o[e] = __builtin_bit_cast(unsigned, v);
          }
          pf[k] = o;
        }
      } else {
        const u32x4* src = reinterpret_cast<const u32x4*>(xg + (int64_t)l * p.sx_l + col);
        if (CHUNKED && col >= p.R) {
          pf[k] = u32x4{0u, 0u, 0u, 0u};
        } else {
          pf[k] = __builtin_nontemporal_load(src);
        }
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
  // the fragment buffer is laid out by palu_abx_prepare_b for the (H, G, R) plan: nks_frag k-steps per M-block -- the
  // chunk count of this launch may cover more (R = 32 / 64 through this kernel: one zero-padded 128-column chunk over
  // 2 / 4 fragment k-steps), k-steps beyond it read as zero
  const int nks_frag = p.nks_frag > 0 ? p.nks_frag : nks_tot;
  const u32x4* bf_base = p.bfrag + ((int64_t)(gb * 8 + w) * NMB) * nks_frag * 64 + lane;
  h16x8 bf[NMB][NKS];
  if (!CHUNKED) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        u32x4 v = bf_base[(int64_t)(mb * nks_frag + ks) * 64];
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
          float k1 = ac[mb][4 * j + t], k2 = ac[mb][4 * j + 2 + t];
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
      // lanes n and n+32 hold complementary pairs: swap halves in-register (no LDS round trip);
      // both halves then hold the same sum and write the same word (benign duplicate store).
      unsigned pv = __float_as_uint(part[s]);
      auto sw = __builtin_amdgcn_permlane32_swap(pv, pv, false, false);
      rdst[s * TL] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
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
          u32x4 v = u32x4{0u, 0u, 0u, 0u};
          if (kc * NKS + ks < nks_frag) v = bf_base[(int64_t)(mb * nks_frag + kc * NKS + ks) * 64];
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


// ---------------------------------------------------------------------------------------------
// Fast path: R in {32, 64, 128} (NKS = R/16 k-steps), B fragments register-resident.
//
// FOLD: the query is folded into the B fragments once per launch (in registers):
//   P[r,i] = q_i B[r,i] + q_{i+64} B[r,i+64],  Q[r,i] = q_{i+64} B[r,i] - q_i B[r,i+64]
// so that the MFMA directly produces U = x.P and V = x.Q and the score is sum_i cos*U + sin*V
// (2 FMAs per pair and head instead of 6).  P and Q are rounded to fp16 (MFMA operands): one
// extra operand rounding, the same size as the oracle's own fp16 rounding of K (abx_rope.py:164).
//
// Software pipeline: the MFMAs of block b+1 and the RoPE/reduction epilogue of block b are
// independent instruction streams in one basic block (two accumulator sets), so the matrix pipe
// and the VALU overlap inside a wave as well as across the two waves of a SIMD.
// ORDER2: keep the lo^2/2 term of the angle correction (needed once positions exceed 2^18, see chunk t == 0).
// SHARED: all heads of a latent group use the SAME B (true-GQA checkpoints: the query heads of a group share one KV
// head, palu/model/svd_mistral/modeling_palu_mistral.py:37-59).  K is then reconstructed ONCE per group -- one M-block
// per wave (half of its rows unused) instead of one per head pair, a quarter of the useful MFMA work -- and the 2*NMB
// query heads are applied in the epilogue: rotate (k_i, k_i+64) once per pair, two FMAs per head.  The fragments come
// from palu_abx_prepare_b on the [G, R, D] shared factor (one head per group); FOLD is not used.
// ACC: 0 = scores rounded to fp16 and stored to `out` (single pass); 1 / 2 = this launch is one pass over a 128-column
// window of a wider rank: fp32 partial scores stored to (1) / added to (2) p.acc.
template <int NKS, int NMB, bool FOLD, bool TIMING = false, int QBITS = 0, bool ORDER2 = false, bool SHARED = false, int ACC = 0>
__global__ __launch_bounds__(NTHREADS, 2) void abx_rope_kernel(AbxParams p) {
  using Geo = LdsGeom<NKS>;
  static_assert(!SHARED || !FOLD, "the shared-B variant keeps q in the epilogue");
  constexpr int HPW = 2 * NMB;
  constexpr int NMM = SHARED ? 1 : NMB;      // M-blocks per wave on the matrix pipe
  constexpr int NRING = 3;                // X tiles resident in LDS
  constexpr int RED_STRIDE = 8 * 4 * TL;  // floats per red buffer: [8 waves][4 slots][TL]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // red[3][8 waves][4 heads][TL] partial sums follow the tile ring; LDS is addressed by 32-bit offsets
  // (address-space-3 pointers built from integers) so that every access is lane-constant + immediate
  const unsigned smem_lds = (unsigned)reinterpret_cast<uintptr_t>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hi = lane >> 5;
  int stamp_i = 0;
  auto stamp = [&]() {
    if (TIMING) {
      unsigned long long t = __builtin_readcyclecounter();
      if (lane == 0 && stamp_i < 64) p.dbg[((size_t)blockIdx.x * 8 + w) * 64 + stamp_i] = t;
      ++stamp_i;
    }
  };
  stamp();

  const int ngb = p.G * p.HB;
  const int gb = blockIdx.x % ngb;
  const int cidx = blockIdx.x / ngb;
  const int g = gb / p.HB, hb = gb % p.HB;

  // Full 128-row tiles are dealt out evenly; the partial tail tile (L % 128 rows, e.g. the one new row of a decode step
  // on a 64k prompt) goes to the LAST workgroup of the group, which is among those with the fewest full tiles, and costs
  // only the 32-row blocks it really has (tail_nb below): a ceil(L/128)-tile split made one workgroup per group carry a
  // whole extra tile (17 instead of 16 at C2: 6 % of the kernel) for a single row.
  const int nt_full = p.L / TL;
  const int base = nt_full / p.nch, rem = nt_full % p.nch;
  const int tile0 = cidx * base + min(cidx, rem);
  const bool has_tail = (p.L % TL) != 0 && cidx == p.nch - 1;
  const int ntile = base + (cidx < rem ? 1 : 0) + (has_tail ? 1 : 0);
  if (ntile <= 0) return;
  const int tail_nb = has_tail ? (p.L % TL + 31) / 32 : 4;   // 32-row blocks of this workgroup's last tile: 1..4

  const h16* xg = p.x + (int64_t)g * p.sx_g;

  // ---- staging by LDS-DMA (buffer_load_dwordx4 ... lds): wave w, piece k fills the 64 consecutive 16-byte
  //      LDS slots [512k + 64w, +64) of a tile, i.e. rows [k*RPP, (k+1)*RPP) of the tile.  The XOR swizzle is
  //      applied to the per-lane SOURCE offset (the DMA destination is lane-linear) and does not depend on k,
  //      so the lane offset is ONE constant VGPR and tile/piece select is a scalar offset: no vector ALU work
  //      per piece.  Rows past the end of the slab are out of range of the descriptor (read as zero / dropped;
  //      their scores are never stored).  Hidden from the compiler (inline asm), so the completion wait is
  //      ours: s_waitcnt vmcnt(0) before the barrier that publishes the tile.
  constexpr int RPP = NTHREADS / Geo::CPR;                      // rows per piece
  u32x4 xrs;
  {
    const unsigned long long xb = reinterpret_cast<unsigned long long>(xg);
    xrs[0] = __builtin_amdgcn_readfirstlane((unsigned)xb);
    xrs[1] = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32));
    const int ncols = (QBITS == 0 && p.ncols > 0 && p.ncols < 16 * NKS) ? p.ncols : 16 * NKS;
    xrs[2] = __builtin_amdgcn_readfirstlane((unsigned)(((int64_t)(p.L - 1) * p.sx_l + ncols) * 2));
    xrs[3] = 0x00020000u;
  }
  unsigned dma_voff = (unsigned)((tid / Geo::CPR) * p.sx_l * 2 + Geo::swz(tid / Geo::CPR, tid % Geo::CPR) * 16);
  if (QBITS == 0 && p.ncols > 0 && p.ncols < 16 * NKS) {
    // Rank below the 16*NKS-column window (e.g. R = 96 on the 128-column kernel; the B fragments are zero-padded by
    // palu_abx_prepare_b): lanes whose 16-byte source chunk lies beyond the row get an offset at the end of the
    // descriptor's range -- out of range whatever the scalar offset adds -- so they never read the NEXT row (which, for
    // the last cached row, is uninitialised memory: 0 x NaN would poison a score).  Their LDS slots are zeroed once here:
    // correct whether the hardware zero-fills or drops an out-of-range DMA lane.
    if (Geo::swz(tid / Geo::CPR, tid % Geo::CPR) * 8 >= p.ncols) dma_voff = xrs[2];
#pragma unroll
    for (int k = 0; k < NRING * Geo::SPT; ++k)
      *reinterpret_cast<u32x4*>(smem + (size_t)(tid + NTHREADS * k) * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
  }
  const unsigned row_bytes = __builtin_amdgcn_readfirstlane((unsigned)(p.sx_l * 2));
  auto dma_piece = [&](int tt, int slot, int k) {
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((tile0 + tt) * TL + k * RPP) * row_bytes);
    const unsigned dst = (unsigned)(slot * Geo::TILE_BYTES + (NTHREADS * k + 64 * w) * 16);
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(dst), "v"(dma_voff), "s"(xrs), "s"(soff)
        : "memory");
  };
  auto dma_tile = [&](int tt, int slot) {
#pragma unroll
    for (int k = 0; k < Geo::SPT; ++k) dma_piece(tt, slot, k);
  };
  auto dma_wait = [&]() {
    if (QBITS == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---- quantised staging (QBITS = 3/4): thread = (row = tid/4, quarter of the row); the packed codes of
  //      tile t+3 travel in registers while tiles t..t+2 sit in the LDS ring; they are dequantised exactly
  //      like quant.py:39 -- (code - zero) exact in fp16 via the 1024+code trick, one fp16 multiply by
  //      scale -- and written to the same swizzled fp16 tile image the MFMA loop reads.
  constexpr int CPQ = 4 * NKS;                               // codes per thread and tile (R/4)
  constexpr int NW = QBITS ? (CPQ * QBITS) / 32 : 1;         // packed dwords per thread and tile
  static_assert(QBITS == 0 || (CPQ * QBITS) % 32 == 0, "packed quarter rows must be whole dwords");
  unsigned qraw[NW];
  unsigned qmeta = 0;
  const unsigned char* xqg = QBITS ? p.xq + (int64_t)g * p.sq_g : nullptr;
  const h16* xmg = QBITS ? p.xmeta + (int64_t)g * p.sm_g : nullptr;
  auto load_q = [&](int tt) {
    const int row = tid >> 2, quarter = tid & 3;
    const int l = min((tile0 + tt) * TL + row, p.L - 1);
    const unsigned* src = reinterpret_cast<const unsigned*>(xqg + (int64_t)l * p.sq_l) + quarter * NW;
    // p.ncols < 16 * NKS: this launch covers a 128-column WINDOW of which only ncols are real (rank 96, or the last
    // window of rank 160): dwords beyond the window's packed bytes are not read (they may lie behind the allocation);
    // their codes read as 0 and dequantise to a finite value that meets zero-padded B fragments
    const int qwin_dw = (p.ncols > 0 && p.ncols < 16 * NKS) ? (p.ncols * QBITS) / 32 : 4 * NW;
#pragma unroll
    for (int k = 0; k < NW; ++k) qraw[k] = (quarter * NW + k < qwin_dw) ? __builtin_nontemporal_load(src + k) : 0u;
    qmeta = *reinterpret_cast<const unsigned*>(xmg + (int64_t)l * p.sm_l);
  };
  auto store_q = [&](int slot) {
    const int row = tid >> 2, quarter = tid & 3;
    const h16x2 m2 = __builtin_bit_cast(h16x2, qmeta);
    const h16x2 scale2 = h16x2{m2[0], m2[0]};
    const h16 nb = -((h16)1024.f + m2[1]);                  // exact: zero is an integer in [0, 15]
    const h16x2 negbias2 = h16x2{nb, nb};
    char* dst = smem + slot * Geo::TILE_BYTES + row * Geo::RB;
#pragma unroll
    for (int gq = 0; gq < CPQ / 8; ++gq) {
      unsigned grp;                                          // 8 codes, code e at bits [QBITS*e, +QBITS)
      if (QBITS == 4) {
        grp = qraw[gq];
      } else {
        grp = gq == 0 ? qraw[0]
            : gq == 1 ? __builtin_amdgcn_alignbit(qraw[1 % NW], qraw[0], 24)
            : gq == 2 ? __builtin_amdgcn_alignbit(qraw[2 % NW], qraw[1 % NW], 16)
                      : (qraw[2 % NW] >> 8);
      }
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned c0 = (grp >> (QBITS * (2 * e))) & ((1u << QBITS) - 1);
        const unsigned c1 = (grp >> (QBITS * (2 * e + 1))) & ((1u << QBITS) - 1);
        const unsigned pw = 0x64006400u | c0 | (c1 << 16);   // (1024 + c0, 1024 + c1) as fp16
        h16x2 v = __builtin_bit_cast(h16x2, pw);
        v = (v + negbias2) * scale2;
        o[e] = __builtin_bit_cast(unsigned, v);
      }
      const int c = quarter * (CPQ / 8) + gq;
      *reinterpret_cast<u32x4*>(dst + Geo::swz(row, c) * 16) = o;
    }
  };

  // Prologue order.  Loads return in issue order, so what the RoPE initialisation and the fold need besides the B
  // fragments -- this lane's four inverse frequencies and its query elements, a few bytes -- is requested FIRST: behind
  // the 24 KB of fragments and first tiles a wave issues below they would arrive last, and the ~300 VALU operations of
  // the RoPE initialisation (which depend on nothing else) could not start before the whole prologue ingest has landed.
  // With PALU_ABX_PIN_PROLOGUE both the initialisation and the fold are made to run in front of the first barrier
  // (left alone, hipcc sinks them behind it; see the pins below).  PALU_ABX_B_FIRST requests the fragments before the
  // first X tiles.  Measured (profiles/r03_abx_prologue_variants.txt).
#ifndef PALU_ABX_B_FIRST
#define PALU_ABX_B_FIRST 0
#endif
#ifndef PALU_ABX_PIN_PROLOGUE
#define PALU_ABX_PIN_PROLOGUE 0
#endif
#ifndef PALU_ABX_EARLY_SMALL_LOADS
#define PALU_ABX_EARLY_SMALL_LOADS 1
#endif
  float fr[4];
  h16 fold_qi[NMB], fold_qj[NMB];
  if (PALU_ABX_EARLY_SMALL_LOADS) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fr[j] = p.inv_freq[8 * w + 2 * j + hi];
    if (FOLD) {
      const int m = lane & 31;
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) {
        const int hloc = hb * HPW + 2 * mb + (m & 1);
        const bool valid = hloc < p.gs;
        const int h = g * p.gs + (valid ? hloc : 0);
        const int i = 8 * w + (m >> 2);
        fold_qi[mb] = valid ? p.a[h * p.sa_h + i * p.sa_d] : (h16)0.f;
        fold_qj[mb] = valid ? p.a[h * p.sa_h + (i + 64) * p.sa_d] : (h16)0.f;
      }
    }
  }
  auto first_tiles = [&]() {
    if (QBITS == 0) {
      dma_tile(0, 0);
      dma_tile(min(1, ntile - 1), 1);
    } else {
      load_q(0);
      store_q(0);
      load_q(min(1, ntile - 1));
      store_q(1);
      load_q(min(2, ntile - 1));
    }
  };
  if (!PALU_ABX_B_FIRST || QBITS != 0) first_tiles();

  // ---- B fragments (issued early; consumed by the fold / first MFMA)
  // (nks_frag: k-steps per M-block in the fragment buffer -- NKS, or 8 * chunks when this launch is one pass over a
  //  128-column window of a wider rank; ks0: first k-step of the window)
  const int nks_frag = p.nks_frag > 0 ? p.nks_frag : NKS;
  const u32x4* bf_base = p.bfrag + ((int64_t)(gb * 8 + w) * NMM) * nks_frag * 64 + (int64_t)p.ks0 * 64 + lane;
  h16x8 bf[NMM][NKS];
#pragma unroll
  for (int mb = 0; mb < NMM; ++mb)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4 v = bf_base[(int64_t)(mb * nks_frag + ks) * 64];
      bf[mb][ks] = *reinterpret_cast<h16x8*>(&v);
    }
  if (PALU_ABX_B_FIRST && QBITS == 0) first_tiles();

  stamp();  // 1: B loads issued
  // ---- RoPE state of this lane (C layout: position n, pairs i = 8w + 2j + hi), started one block
  //      early because the pipeline runs one (discarded) epilogue before the first real block.
  float rc[4], rs[4], cs[4], sn[4];
  float lf = (float)(p.pos0 + tile0 * TL + n - 32);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!PALU_ABX_EARLY_SMALL_LOADS) fr[j] = p.inv_freq[8 * w + 2 * j + hi];
    sincos_exact_product(lf, fr[j], &sn[j], &cs[j]);
    sincos_exact_product(32.0f, fr[j], &rs[j], &rc[j]);
#if PALU_ABX_PIN_PROLOGUE
    // pin: left alone, hipcc sinks this arithmetic (fp64 reductions included) and most of the fold below BEHIND the
    // barrier that publishes the first tiles -- ~640 VALU operations per wave on the critical path of every workgroup,
    // serialised between the two waves of a SIMD (tools/time_abx.py: first barrier left at 14.7k / 19.9k ticks instead
    // of ~10k).  An empty asm that takes the values as read-write operands makes them exist here, i.e. while the
    // prologue's loads are still in flight.
    asm volatile("" : "+v"(sn[j]), "+v"(cs[j]), "+v"(rs[j]), "+v"(rc[j]));
#endif
  }
  stamp();  // 2: rope init done
  // ---- query: either folded into the fragments (FOLD) or kept per (pair, head) for the epilogue
  float q1[(FOLD || SHARED) ? 1 : HPW][4], q2[(FOLD || SHARED) ? 1 : HPW][4];
  if (FOLD) {
    // A-fragment lane = row m of the M-block: t = m&1, u = (m>>1)&1, pair = m>>2.
    //   row u=0 (B[:,i])    <- P = q_i B[:,i] + q_{i+64} B[:,i+64]
    //   row u=1 (B[:,i+64]) <- Q = q_{i+64} B[:,i] - q_i B[:,i+64]
    // i.e. new = c_own*own + q_{i+64}*partner with c_own = +/-q_i; partner row = lane^2 (DPP).
    // v_dot2_f32_f16: both products exact in fp32, one rounding to fp16 at the end.
    const int m = lane & 31;
    const int t = m & 1, u = (m >> 1) & 1, pair = m >> 2;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      int hloc = hb * HPW + 2 * mb + t;
      bool valid = hloc < p.gs;
      int h = g * p.gs + (valid ? hloc : 0);
      int i = 8 * w + pair;
      h16 qi, qj;
      if (PALU_ABX_EARLY_SMALL_LOADS) {
        qi = fold_qi[mb];
        qj = fold_qj[mb];
      } else {
        qi = valid ? p.a[h * p.sa_h + i * p.sa_d] : (h16)0.f;
        qj = valid ? p.a[h * p.sa_h + (i + 64) * p.sa_d] : (h16)0.f;
      }
      h16x2 coef;
      coef[0] = u ? -qi : qi;
      coef[1] = qj;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        u32x4 own = __builtin_bit_cast(u32x4, bf[mb][ks]);
        u32x4 res;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned ow = own[e];
          unsigned par = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ow, 0x4E, 0xF, 0xF, false);  // lane^2 (quad_perm 2,3,0,1)
          unsigned lo2 = __builtin_amdgcn_perm(par, ow, 0x05040100u);  // (own.lo, par.lo)
          unsigned hi2 = __builtin_amdgcn_perm(par, ow, 0x07060302u);  // (own.hi, par.hi)
          float r0 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, lo2), coef, 0.f, false);
          float r1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, hi2), coef, 0.f, false);
          h16x2 r2;
          r2[0] = (h16)r0;
          r2[1] = (h16)r1;
          res[e] = __builtin_bit_cast(unsigned, r2);
        }
        bf[mb][ks] = __builtin_bit_cast(h16x8, res);
#if PALU_ABX_PIN_PROLOGUE
        asm volatile("" : "+v"(bf[mb][ks]));      // folded here, fragment by fragment as the loads land (see above)
#endif
      }
    }
  } else if (!SHARED) {
#pragma unroll
    for (int s = 0; s < HPW; ++s) {
      int hloc = hb * HPW + s;
      bool valid = hloc < p.gs;
      int h = g * p.gs + (valid ? hloc : 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int i = 8 * w + 2 * j + hi;
        float v1 = (float)p.a[h * p.sa_h + i * p.sa_d];
        float v2 = (float)p.a[h * p.sa_h + (i + 64) * p.sa_d];
        q1[s][j] = valid ? v1 : 0.f;
        q2[s][j] = valid ? v2 : 0.f;
      }
    }
  }
  // SHARED: the q-dot runs on the matrix cores too.  After the rotation a lane (position n, half hi) holds the 8 rotated
  // key components of ITS 4 pairs -- as fp16 exactly one B operand of v_mfma_f32_32x32x16_f16 (k-slot order is ours to
  // choose: slot 2j = component i_j, slot 2j+1 = component i_j + 64, i_j = 8w + 2j + hi).  The A operand carries q:
  // lane (m, hi) = head m of the block, the same 8 components; rows beyond the block's heads are zero.  One MFMA per
  // 32-position block replaces 8 of the 20 VALU operations per (position, pair).
  h16x8 qa;
  if (SHARED) {
    const int hloc = hb * HPW + n;                       // A row m = lane & 31
    const bool valid = n < HPW && hloc < p.gs;
    const int h = g * p.gs + (valid ? hloc : 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = 8 * w + 2 * j + hi;
      qa[2 * j] = valid ? p.a[h * p.sa_h + i * p.sa_d] : (h16)0.f;
      qa[2 * j + 1] = valid ? p.a[h * p.sa_h + (i + 64) * p.sa_d] : (h16)0.f;
    }
  }

  // scores leave through a buffer store: invalid (row >= L, padded head, pipeline warm-up) lanes get an
  // out-of-range offset that the hardware drops, so the store needs no branch inside the MFMA stream
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  // multi-pass: partial scores of this 128-column window go to an fp32 array instead (same [head][position] indexing);
  // the descriptor range (2 x the fp16 extent) drops the invalid lanes exactly like the fp16 store does
  u32x4 arsrc;
  {
    const unsigned long long ab = reinterpret_cast<unsigned long long>(p.acc);
    arsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)ab);
    arsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(ab >> 32));
    arsrc[2] = __builtin_amdgcn_readfirstlane((unsigned)((((int64_t)(p.H - 1) * p.acc_ld + p.L) * 4)));
    arsrc[3] = 0x00020000u;
  }

  // cross-wave reduction of tile tt (partials in red[rslot]) and the fp16 store
  auto reduce_store = [&](int tt, int rslot) {
    const int slot = tid >> 7, pos = tid & 127;
    unsigned r = smem_lds + (unsigned)(NRING * Geo::TILE_BYTES) +
                 (unsigned)((rslot * RED_STRIDE + (slot & 3) * TL + pos) * sizeof(float));
    asm volatile("" : "+v"(r));                 // keep the wave strides in the immediates
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww)
      s += *(const __attribute__((address_space(3))) float*)(uintptr_t)(r + (unsigned)(ww * 4 * TL * sizeof(float)));
    const int l = (tile0 + tt) * TL + pos;
    const int hloc = hb * HPW + slot;
    const bool ok = tt >= 0 && slot < HPW && l < p.L && hloc < p.gs;
    if (ACC == 0) {
      const unsigned off = ok ? (unsigned)(((int64_t)(g * p.gs + hloc) * p.so_h + l) * 2) : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (h16)s), orsrc, off, 0, 0);
    } else {
      const unsigned aoff = ok ? (unsigned)(((int64_t)(g * p.gs + hloc) * p.acc_ld + l) * 4) : 0xFFFFFFF0u;
      if (ACC == 1) {
        asm volatile("buffer_store_dword %0, %1, %2, 0 offen\n\ts_nop 0" ::"v"(s), "v"(aoff), "s"(arsrc) : "memory");
      } else {
        // every (head, position) is added to exactly once per pass, passes are stream-ordered: deterministic
        asm volatile("buffer_atomic_add_f32 %0, %1, %2, 0 offen\n\ts_nop 0" ::"v"(s), "v"(aoff), "s"(arsrc) : "memory");
      }
    }
  };

  // X fragments are prefetched XD k-steps ahead through a ring of XD registers-sets: the fragment of
  // k-step ks lives in xf[ks % XD] and is refilled with the fragment XD k-steps later (possibly of the
  // next block) right after its MFMAs have issued -> LDS latency never sits in front of an MFMA.
  // LDS byte address of the fragment (tile slot, block blk, k-step ks) = fa[ks] + blk*32*RB: the swizzle key
  // (row >> SH) & MASK does not depend on blk, so fa[ks] is one lane-constant VGPR per k-step that only moves
  // by the (scalar) ring-slot distance once per tile; blk rides in the instruction's immediate offset.
  constexpr int XD = NKS < 4 ? NKS : 4;
  h16x8 xf[XD];
  unsigned fa[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) fa[ks] = smem_lds + (unsigned)(n * Geo::RB + Geo::swz(n, 2 * ks + hi) * 16);
  auto read_frag = [&](int i, int blk) {
    return *(const __attribute__((address_space(3))) h16x8*)(uintptr_t)(fa[i] + (unsigned)(blk * 32 * Geo::RB));
  };
  // partial sums: after the half-swap below, lane (n, hi) holds head 2mb + hi of position n
  const unsigned red_lane = smem_lds + (unsigned)(NRING * Geo::TILE_BYTES) +
                            (unsigned)(((w * 4 + hi) * TL + n) * sizeof(float));

  // RoPE state of the epilogue lives in fr/rc/rs/cs/sn/lf (above).
  // ---- hand-interleaved region: the NKS*NMB MFMAs of block blk of the current tile and the epilogue of the
  //      PREVIOUS block (accumulators acP, written to red[erslot] as block eblk) are emitted alternately -- one
  //      epilogue chunk (4 VALU ops: coefficients of a pair | the pair's 2 FMAs x 2 heads of an M-block | state
  //      advance) per MFMA gap -- and pinned with sched_barrier(0): left alone, hipcc clusters 8 MFMAs then ~60
  //      VALU, and a wave's matrix and vector time simply add up (measured: 49 % MFMA issue for a lone wave).
  //      The SIMD issues about one instruction per 4-5 cycles whatever its type, so instruction COUNT is what
  //      bounds this kernel: addresses are lane constants + immediates, staging offsets are scalar.
  //      KIND 0 also carries the LDS-DMA pieces of tile stt into ring slot sslot (or the quantised staging),
  //      KIND 1 the reduce/store of tile stt (partials in red[sslot]); LAST = block 3: the fragment prefetch
  //      crosses into block 0 of the next tile (ring distance nd bytes); KIND 3 = drain (epilogue only).
  auto region = [&](auto kind_c, auto last_c, auto exact_c, f32x16 (&acN)[NMM], int blk, int erslot, int eblk,
                    const f32x16 (&acP)[NMM], int stt, int sslot, unsigned nd) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    constexpr bool EXACT = decltype(exact_c)::value;
    constexpr int GAPS = NKS * NMM;
    // chunks per pair j: [ang, lo, cc, ss] | per M-block: 2 FMAs x 2 heads | [advance cos, sin by 32 positions].
    // (Packed fp32 -- v_pk_fma_f32 over head pairs -- was measured and is an anti-lever beside MFMAs on gfx950:
    //  75.5 vs 69 us; hipcc itself unpacks half of them again in the MFMA shadow.)
    constexpr int CPP = 2 + NMB;
    constexpr int NC = 4 * CPP;
    constexpr int CPG = (NC + GAPS - 1) / GAPS;
#pragma unroll
    for (int mb = 0; mb < NMM; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acN[mb][e] = 0.f;
    float part[HPW];
#pragma unroll
    for (int s = 0; s < HPW; ++s) part[s] = 0.f;
    float cc = 0.f, ss = 0.f, kr1 = 0.f, kr2 = 0.f;
    h16x8 krb;
#pragma unroll
    for (int e = 0; e < 8; ++e) krb[e] = (h16)0.f;
    auto chunk = [&](int c) {
      const int j = c / CPP, t = c % CPP;
      if (t == 0) {
        // cos/sin at the oracle's fp32-rounded angle fl(l*f): exact angle = ang + lo.  First order in lo drops
        // lo^2/2 < 3.1e-5 for positions < 2^18 (|lo| <= half an ulp of the angle); the host selects ORDER2 beyond that
        // (palu_abx_rope_f16: pos0 + L > 262144), e.g. the harness's max_position_embeddings = 300000
        // Waves whose angles all stay below 1024 rad (waves 4..7 = pairs 32..63 at theta 1e4 up to 102k positions) have a
        // residual <= 2^-15 = 3.1e-5 rad, the size of the second-order term dropped here anyway: they use the exact-angle
        // state as it is (EXACT = false: 4 of the 16 instructions per pair less; selected per wave at the bottom).
        const float ang = lf * fr[j];
        const float lo = fmaf(lf, fr[j], -ang);
        if (!EXACT) {
          cc = cs[j];
          ss = sn[j];
        } else if (ORDER2) {
          const float hh = 0.5f * lo * lo;
          cc = fmaf(-hh, cs[j], fmaf(lo, sn[j], cs[j]));
          ss = fmaf(-hh, sn[j], fmaf(-lo, cs[j], sn[j]));
        } else {
          cc = fmaf(lo, sn[j], cs[j]);
          ss = fmaf(-lo, cs[j], sn[j]);
        }
        // (no empty-asm pin on cc / ss here: hipcc pads one s_nop behind an asm whose outputs the next VALU reads --
        //  16 of the loop's 20 s_nop per tile -- and the 4-VALU-per-gap interleave holds without it; -0.7 us at C2)
      } else if (t <= NMB) {
        const int mb = t - 1;
        if (SHARED) {
          // one reconstructed key per group: rotate the pair once and hand it to the score MFMA as an fp16 pair
          if (mb == 0) {
            const float k1 = acP[0][4 * j], k2 = acP[0][4 * j + 2];
            kr1 = fmaf(-ss, k2, cc * k1);
            kr2 = fmaf(ss, k1, cc * k2);
            krb[2 * j] = (h16)kr1;
            krb[2 * j + 1] = (h16)kr2;
          }
        } else
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const float k1 = acP[mb][4 * j + h2], k2 = acP[mb][4 * j + 2 + h2];
          const int s = 2 * mb + h2;
          if (FOLD) {
            part[s] = fmaf(cc, k1, fmaf(ss, k2, part[s]));
          } else {
            const float t1 = fmaf(q2[s][j], k2, q1[s][j] * k1);
            const float t2 = fmaf(-q1[s][j], k2, q2[s][j] * k1);
            part[s] = fmaf(cc, t1, fmaf(ss, t2, part[s]));
          }
          asm volatile("" : "+v"(part[s]));
        }
      } else {
        // advance the exact-angle state by 32 positions: (c, s) <- (c*rc - s*rs, s*rc + c*rs)
        const float c2 = fmaf(-sn[j], rs[j], cs[j] * rc[j]);
        sn[j] = fmaf(cs[j], rs[j], sn[j] * rc[j]);
        cs[j] = c2;
        asm volatile("" : "+v"(cs[j]), "+v"(sn[j]));
      }
    };
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int mb = 0; mb < NMM; ++mb) {
        const int gap = ks * NMM + mb;
        if (KIND != 3) {
          acN[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[mb][ks], xf[ks % XD], acN[mb], 0, 0, 0);
          asm volatile("" : "+v"(acN[mb]));    // empty volatile asm = ordering pin (arithmetic floats across sched_barrier)
        }
#pragma unroll
        for (int q = 0; q < CPG; ++q)
          if (gap * CPG + q < NC) chunk(gap * CPG + q);
        if (KIND != 3 && mb == NMM - 1) {
          // refill the ring entry just consumed with the fragment XD k-steps ahead
          const int r = ks + XD;
          if (r < NKS) {
            xf[ks % XD] = read_frag(r, blk);
            if (LAST) {                        // fa[r] has served its last read of this tile: move it to the next slot
              fa[r] += nd;
              asm volatile("" : "+v"(fa[r]));
            }
          } else {
            if (LAST) {
              fa[r - NKS] += nd;
              asm volatile("" : "+v"(fa[r - NKS]));
              xf[ks % XD] = read_frag(r - NKS, 0);
            } else {
              xf[ks % XD] = read_frag(r - NKS, blk + 1);
            }
          }
        }
        if (KIND == 0 && QBITS == 0 && gap % (2 * NMM) == 1 && gap / (2 * NMM) < Geo::SPT)
          dma_piece(stt, sslot, gap / (2 * NMM));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    lf += 32.0f;
    // lanes n and n+32 hold complementary pairs of the same position: one half-swap per head PAIR leaves head
    // 2mb in the low half and head 2mb+1 in the high half, so one add and one store cover two heads
    const unsigned rdst = red_lane + (unsigned)((erslot * RED_STRIDE + eblk * 32) * sizeof(float));
    if (SHARED) {
      // scores of the block's heads over this wave's 16 key components: D[m = head, n = position]; lanes hi = 0 hold
      // rows 0..3 in registers 0..3 (both halves' components are already summed by the contraction)
      f32x16 sd;
#pragma unroll
      for (int e = 0; e < 16; ++e) sd[e] = 0.f;
      sd = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa, krb, sd, 0, 0, 0);
      // drain only (once per workgroup): slack between the last q-dot and its v_permlane32_swap.  Belt and braces for the
      // cold-start flake of profiles/r03_shared_b_cold_start.txt (60 of 60 fresh processes right with it even under the
      // priority mode that provoked it); the steady state keeps hipcc's own padding.
      if (KIND == 3) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(sd));
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sd[2 * mb]), __float_as_uint(sd[2 * mb + 1]), false, false);
        *(__attribute__((address_space(3))) float*)(uintptr_t)(rdst + (unsigned)(mb * 2 * TL * sizeof(float))) =
            __uint_as_float(sw[0]);
      }
    } else {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[2 * mb]), __float_as_uint(part[2 * mb + 1]), false, false);
      *(__attribute__((address_space(3))) float*)(uintptr_t)(rdst + (unsigned)(mb * 2 * TL * sizeof(float))) =
          __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    }
    if (KIND == 0 && QBITS != 0) {
      store_q(sslot);                         // registers hold tile min(stt, ntile-1)
      load_q(min(stt + 1, ntile - 1));
    }
    if (KIND == 1) reduce_store(stt, sslot);
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x16 accA[NMM], accB[NMM];
#pragma unroll
  for (int mb = 0; mb < NMM; ++mb)
#pragma unroll
    for (int e = 0; e < 16; ++e) accB[mb][e] = 0.f;

  stamp();  // 3: fold done

  // the second-dispatched half of the workgroup loses issue arbitration on every segment
  // (priority, then age): give it static priority so both waves of a SIMD finish together
  const bool young = w >= 4;
  if (p.prio_mode == 1 && young) __builtin_amdgcn_s_setprio(1);

  dma_wait();
  stamp();  // 4: first tiles landed
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < XD; ++ks) xf[ks] = read_frag(ks, 0);

  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  using NotLast = std::false_type;
  using Last = std::true_type;
  // ring slots: tile tt sits in slot tt % 3 (LDS ring and red[] alike); kept as three rotating scalars
  int s_cur = 0, s_nxt = 1, s_prv = 2;     // tt % 3, (tt + 1) % 3, (tt + 2) % 3 == (tt - 1) % 3
  auto main_loop = [&](auto exact_c) {
  // the partial tail tile (last tile of the last workgroup of a group, tail_nb < 4 blocks) is handled after the loop
  const int nmain = tail_nb < 4 ? ntile - 1 : ntile;
  for (int tt = 0; tt < nmain; ++tt) {
    stamp();  // 5+2*tt: arrive at barrier
    if (tt > 0) {
      dma_wait();  // this wave's pieces of tile tt+1 have landed -> published by the barrier
      __syncthreads();
    }
    stamp();  // 6+2*tt: leave barrier
    // young half leads the first half tile -- but never on a workgroup's LAST tile: raising the priority there is what
    // provoked the shared-B kernel's cold-start flake (20 of 30 fresh processes right with it, 30 of 30 without the raise
    // on the last tile, 30 of 30 with s_nop in place of both s_setprio; profiles/r03_shared_b_cold_start.txt)
    if (p.prio_mode == 2 && young && tt + 1 < ntile) __builtin_amdgcn_s_setprio(1);
    // one straight-line region per block: MFMAs of block b+1, RoPE epilogue of block b, staging of tile
    // tt+2 into the ring, cross-wave reduction + store of tile tt-2
    if (TIMING && (p.exp_flags & 1) && young) continue;   // experiment: one computing wave per SIMD
    const unsigned nd = (unsigned)((s_nxt - s_cur) * Geo::TILE_BYTES);
    region(K0{}, NotLast{}, exact_c, accA, 0, s_prv, 3, accB, min(tt + 2, ntile - 1), s_prv, 0u);   // tt == 0: epilogue result discarded
    region(K1{}, NotLast{}, exact_c, accB, 1, s_cur, 0, accA, tt - 2, s_nxt, 0u);
    if (p.prio_mode == 2 && young) __builtin_amdgcn_s_setprio(0);  // ... the old half catches up in the second
    region(K2{}, NotLast{}, exact_c, accA, 2, s_cur, 1, accB, 0, 0, 0u);
    region(K2{}, Last{}, exact_c, accB, 3, s_cur, 2, accA, 0, 0, nd);       // prefetches block 0 of the next tile
    const int t3 = s_prv;
    s_prv = s_cur;
    s_cur = s_nxt;
    s_nxt = t3;
  }
  int drain_slot = s_prv, drain_blk = 3;   // red[] slot / block index the epilogue of the last computed block goes to
  if (tail_nb < 4) {
    // Tail tile: only the 32-row blocks that hold rows < L (the others keep stale partials in red[]: their rows are
    // >= L and never stored).  Its X rows were staged by the loop (staging indices clamp to ntile - 1), its block-0
    // fragments were prefetched by the last region of the previous tile.  The last computed block is handed to the drain
    // in accB whatever its parity (one drain instantiation, one live accumulator set).
    const int tt = ntile - 1;
    if (tt > 0) {
      dma_wait();
      __syncthreads();
    }
    region(K2{}, NotLast{}, exact_c, accA, 0, s_prv, 3, accB, 0, 0, 0u);           // block 0 | epilogue of the previous tile's block 3
    if (tail_nb >= 2) region(K1{}, NotLast{}, exact_c, accB, 1, s_cur, 0, accA, tt - 2, s_nxt, 0u);
    else reduce_store(tt - 2, s_nxt);                                                // (what region K1 carries)
    if (tail_nb == 3) region(K2{}, NotLast{}, exact_c, accA, 2, s_cur, 1, accB, 0, 0, 0u);
    if (tail_nb != 2) {
#pragma unroll
      for (int mb = 0; mb < NMM; ++mb) accB[mb] = accA[mb];
    }
    drain_slot = s_cur;
    drain_blk = tail_nb - 1;
    const int t3 = s_prv;
    s_prv = s_cur;
    s_cur = s_nxt;
    s_nxt = t3;
  }
  region(K3{}, NotLast{}, exact_c, accA, 0, drain_slot, drain_blk, accB, 0, 0, 0u);   // epilogue of the very last block
  };
  // ORDER2 launches (positions beyond 2^18) keep the correction on every wave.  The shortcut (EXACT = false) is taken by
  // a wave only when every angle it will see stays below 1024 rad: |fl(l*f) - l*f| <= half an ulp = 2^-15 = 3.1e-5 rad
  // there -- the size of the second-order term the exact path drops anyway.  Decided from the caller's table itself
  // (inv_freq is an argument: any theta / custom table is safe), wave-uniform: at theta = 1e4 waves 4-7 qualify up to
  // 102k positions (config 2), nobody does at 256k.
  float frmax = fmaxf(fmaxf(fr[0], fr[1]), fmaxf(fr[2], fr[3]));
  frmax = wave_max(fabsf(frmax));
  const bool small_angles = frmax * (float)(p.pos0 + p.L) < 1024.0f;
  if (ORDER2 || w < 4 || !small_angles || (p.exp_flags & 4)) main_loop(std::true_type{}); else main_loop(std::false_type{});
  stamp();
  dma_wait();
  __syncthreads();
  reduce_store(ntile - 2, s_nxt);   // (ntile - 2) % 3 == (ntile + 1) % 3
  reduce_store(ntile - 1, s_prv);
  stamp();
}


template <typename K>
int launch_kernel(K kern, int smem, const AbxParams& p, int nwg, hipStream_t stream, int nthreads = NTHREADS) {
  const int rc = palu_func_max_lds(reinterpret_cast<const void*>(kern), smem);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(nthreads), smem, stream, p);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

struct AbxPlan {
  int gs, nmb, hpw, hb, nks_tot, nkc;
  bool chunked;
};

inline bool abx_plan(int H, int G, int R, AbxPlan* pl) {
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

// Wave priorities inside a workgroup (PALU_ABX_PRIO_MODE overrides): 0 none, 1 waves 4-7 raised for the whole kernel,
// 2 waves 4-7 raised for the first half of every tile (both waves of a SIMD then finish a tile together; worth 1-2 % on
// the packed score kernels, nothing at C2).  The shared-B kernel runs WITHOUT priorities: with mode 2 its first launch in
// a fresh process returned, in 20-30 % of the processes, one wave's partial scores of the last 16 rows of a workgroup's
// last tile wrong (lanes 16-31 / 48-63 of the drain's q-dot); 0 of 98 cold processes with modes 0 / 1, and the per-head
// and packed kernels are deterministic under mode 2 (72 of 72 and 48 of 48 fresh processes).  Not understood -- the MFMA -> v_permlane32_swap
// distance in the drain is 12 wait states, and tools/ubench_mfma_hazard.hip shows 6 are enough with and without a
// second wave on the matrix pipe -- so the toggling is simply not used there (profiles/r03_shared_b_cold_start.txt).
inline int abx_prio_mode(bool shared = false) {
  static int m = -2;
  if (m == -2) {
    const char* e = palu_exp_env("PALU_ABX_PRIO_MODE");
    m = e ? atoi(e) : -1;
  }
  return m >= 0 ? m : 0;
}

// fills the launch-independent part of the parameters; returns the number of workgroups
inline int abx_fill_params(AbxParams& p, const AbxPlan& pl, int H, int G, int L, int R, int pos0) {
  p.H = H; p.G = G; p.gs = pl.gs; p.HB = pl.hb; p.L = L; p.R = R; p.pos0 = pos0;
  p.nt_total = (L + TL - 1) / TL;
  p.nkc = pl.nkc;
  p.dbg = nullptr;
  p.prio_mode = abx_prio_mode();
  static int exp_flags = -1;        // PALU_ABX_EXP: experiment flags (4 = angle correction on every wave); read once
  if (exp_flags < 0) {
    const char* e = palu_exp_env("PALU_ABX_EXP");
    exp_flags = e ? atoi(e) : 0;
  }
  p.exp_flags = exp_flags;
  const int ngb = G * pl.hb;
  static int cu_cap = -1;           // PALU_ABX_CUS: experiments that leave part of the GPU to another kernel
  if (cu_cap < 0) {
    const char* e = palu_exp_env("PALU_ABX_CUS");
    cu_cap = e ? atoi(e) : 0;
  }
  const int cus = (cu_cap > 0 && cu_cap < palu_num_cus()) ? cu_cap : palu_num_cus();
  int nch = cus / ngb;              // one 8-wave workgroup per CU
  if (nch < 1) nch = 1;
  if (nch > p.nt_total) nch = p.nt_total;
  p.nch = nch;
  return nch * ngb;
}

}  // namespace
