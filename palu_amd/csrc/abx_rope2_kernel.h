// Two-band score kernel: the fused  K = X.B -> RoPE -> q.K^T  of abx_rope_kernel.h with the RoPE pairs split by
// frequency.  Replaces the Triton `_abx_fwd` (kernel/abx_rope.py:79-111) like abx_rope_kernel does; same numerics
// contract (oracle `torch_abx`, kernel/abx_rope.py:152-171).
//
// With the query folded into the weight (P[r,i] = q_i B[r,i] + q_{i+64} B[r,i+64], Q[r,i] = q_{i+64} B[r,i] -
// q_i B[r,i+64], abx_rope_kernel.h FOLD) the score of head h at position l is
//     s[h,l] = sum_i cos(l f_i) U_i + sin(l f_i) V_i,      U_i = x_l . P_h[:,i],  V_i = x_l . Q_h[:,i].
// HIGH band (pairs 0..31, f_i >= 0.0133 at theta = 1e4): as before -- U, V on the matrix cores, rotation coefficients by
//   recurrence (+ the correction to the oracle's fp32-rounded angle), 2 FMAs per (position, pair, head).
// LOW band (pairs 32..63): inside a 128-position tile centred at l_c the angle is phi_i + tau psi_i with
//   phi_i = l_c f_i, psi_i = 64 f_i <= 0.64 rad, tau in (-1, 1), so
//     cos(phi + tau psi) = sum_k tau^k psi^k / k! cos(phi + k pi/2)      (Taylor in tau psi, degree 7: < 7e-7)
//   and the band's score is a degree-7 polynomial in tau whose coefficients are dot products with per-tile weights:
//     s_low[h, l_c + d] = sum_k (tau psi_max)^k / k!  *  x_l . W_{h,k},
//     W_{h,k}[r] = sum_i a'_{k,i} P_h[r,i] + b'_{k,i} Q_h[r,i],   a' = (psi_i/psi_max)^k cos(phi_i + k pi/2), b' likewise sin.
//   Stage 1 (once per tile): W = [P|Q]_low . [a'; b']   -- 64 x v_mfma_f32_16x16x32_f16 per tile and workgroup;
//   stage 2 (per 32-position block): x . W^T            -- ONE 32-row M-block (4 heads x 8 terms) instead of eight.
//   a', b' depend on positions and frequencies only: a table built once (palu_rope_table_build), read from L2.
// Per 128-position tile and workgroup: 256 + 32 + 32 MFMA-equivalents instead of 512, and 1024 + ~150 VALU wave
// instructions of RoPE work instead of 2048.  The low band's angle is the exact product l f_i (no emulation of the
// oracle's fp32 rounding of the angle, which is <= 2^-14 rad there: the launcher only selects this kernel while
// f_32 (pos0 + L) < 2048 rad).
//
// Work split of the 8 waves (2 per SIMD): wave w owns the high pairs 4w..4w+3 of all 4 heads (one M-block: rows =
// (pair 2) x (P|Q) x (head 4) per lane half) for every 32-position block; stage 1: r-block w (16 latent columns) of all
// heads; stage 2 of block s: waves s and s+4, half of the k-steps each.  Tile staging (LDS-DMA ring), software pipeline
// (MFMAs of block b+1 beside the epilogue of block b) and the cross-wave reduction follow abx_rope_kernel.
#pragma once
#include "abx_rope_kernel.h"

namespace {

// (abx2_dot2x8, the asm block of eight v_dot2_f32_f16 the folds use, lives in abx_fold.h)
constexpr int ABX2_I0 = 32;       // first pair of the low band
constexpr int ABX2_K = 8;         // polynomial terms (degree 7)

// fragment buffer of the two-band kernel (follows the abx_rope_kernel fragments in the same allocation):
//   high [g][w 8][j NKS][lane 64] u32x4 : A operand of v_mfma_f32_32x32x16_f16; lane = m + 32*hiA, row m <-> h0 = m&1,
//        u = (m>>1)&1 (0: d = i, 1: d = i + 64), pair bit = (m>>2)&1, h1 = (m>>3)&1, pp = m>>4; head 2*h1 + h0,
//        pair i = 4w + 2pp + pair bit; k-slots r = 16 ks + 8 hiA + e with ks = (j + (w >= 4 ? NKS/2 : 0)) % NKS: waves 4-7
//        walk the k-steps half a turn ahead, so that "the first NKS/2 local k-steps" are different halves of the rank for
//        the two waves that share a block's stage 2.  After the MFMA lane (n, hi) holds register 8pp + 4h1 + 2u + h0.
//   low  [g][rb NKS][h 4][cs 2][lane 64] u32x4 : A operand of v_mfma_f32_16x16x32_f16; lane = m16 + 16 q: row
//        r = 16 rb + m16, k-slots (e4 = 0..3) pair i = 32 + 16 cs + 4 q + e4 as the fp16 pair (B[h][r][i], B[h][r][i+64]).
inline size_t abx2_frag_u32x4(int G, int nks) { return (size_t)G * 8 * nks * 64 + (size_t)G * nks * 4 * 2 * 64; }

__global__ void abx2_prepare_b_kernel(const h16* __restrict__ b, int64_t sb_h, int64_t sb_r, int64_t sb_d, int G, int R,
                                      int nks, u32x4* __restrict__ out, int64_t n_hi, int64_t total) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  h16x8 v;
  if (idx < n_hi) {
    const int lane = (int)(idx & 63);
    int64_t t = idx >> 6;
    const int j = (int)(t % nks); t /= nks;
    const int w = (int)(t % 8);
    const int g = (int)(t / 8);
    const int ks = (j + (w >= 4 ? nks / 2 : 0)) % nks;
    const int m = lane & 31, hiA = lane >> 5;
    const int h0 = m & 1, u = (m >> 1) & 1, pb = (m >> 2) & 1, h1 = (m >> 3) & 1, pp = m >> 4;
    const int h = g * 4 + 2 * h1 + h0;
    const int d = 4 * w + 2 * pp + pb + 64 * u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 16 * ks + 8 * hiA + e;
      v[e] = r < R ? b[h * sb_h + r * sb_r + d * sb_d] : (h16)0.f;
    }
  } else {
    const int64_t i2 = idx - n_hi;
    const int lane = (int)(i2 & 63);
    int64_t t = i2 >> 6;
    const int cs = (int)(t & 1); t >>= 1;
    const int hh = (int)(t & 3); t >>= 2;
    const int rb = (int)(t % nks);
    const int g = (int)(t / nks);
    const int m16 = lane & 15, q = lane >> 4;
    const int r = 16 * rb + m16;
    const int h = g * 4 + hh;
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const int i = ABX2_I0 + 16 * cs + 4 * q + e4;
      v[2 * e4] = r < R ? b[h * sb_h + r * sb_r + i * sb_d] : (h16)0.f;
      v[2 * e4 + 1] = r < R ? b[h * sb_h + r * sb_r + (i + 64) * sb_d] : (h16)0.f;
    }
  }
  out[idx] = *reinterpret_cast<u32x4*>(&v);
}

// Low-band coefficient table: [tile][cs 2][q 4][k 8] u32x4 (1 KB per tile) = the non-zero half of the B operand of stage 1
// (v_mfma_f32_16x16x32_f16) for the 128-position tile starting at absolute position 128 * (tile_first + tile): MFMA lane
// k + 16 q holds term k < 8 (lanes of the terms 8..15 carry zeros and load nothing), k-slots (e4, u) = pair
// i = 32 + 16 cs + 4 q + e4, u = 0: a'_{k,i}, u = 1: b'_{k,i}.  fp64 arithmetic on the caller's fp32 frequencies (the angle
// is the exact product, phi = (l0 + 63.5) f_i), one rounding to fp16.
__global__ void abx2_rope_table_kernel(const float* __restrict__ inv_freq, int tile_first, int ntiles, u32x4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)ntiles * 64) return;
  const int k = (int)(idx & 7), q = (int)((idx >> 3) & 3), cs = (int)((idx >> 5) & 1);
  const int ti = (int)(idx >> 6);
  const double psimax = 64.0 * (double)inv_freq[ABX2_I0];
  const double lc = (double)(tile_first + ti) * 128.0 + 63.5;
  h16x8 v;
#pragma unroll
  for (int e4 = 0; e4 < 4; ++e4) {
    const int i = ABX2_I0 + 16 * cs + 4 * q + e4;
    const double f = (double)inv_freq[i];
    double sn, cn;
    sincos(lc * f, &sn, &cn);
    double rel = 1.0;
    const double ratio = (64.0 * f) / psimax;
    for (int t = 0; t < k; ++t) rel *= ratio;
    // cos(phi + k pi/2), sin(phi + k pi/2)
    const double ck = (k & 1) ? ((k & 2) ? sn : -sn) : ((k & 2) ? -cn : cn);
    const double sk = (k & 1) ? ((k & 2) ? -cn : cn) : ((k & 2) ? -sn : sn);
    v[2 * e4] = (h16)(float)(rel * ck);
    v[2 * e4 + 1] = (h16)(float)(rel * sk);
  }
  out[idx] = *reinterpret_cast<u32x4*>(&v);
}

// RoPE start tables of the position-split kernel (abx_rope3_kernel.h), behind the coefficient tiles in the same allocation:
//   T1 [tile][hi 2][q 16][2] fp32 = (cos, sin) of the EXACT product (128 (tile_first + tile)) f_i, pair i = 4 (q >> 1) + 2 (q & 1) + hi
//      (the order a lane of that kernel holds its 16 high-band pairs in), 256 B per tile;
//   T2 [n 0..32][hi 2][q 16][2] fp32 = (cos, sin)(n f_i): the lane's offset inside a block (n < 32), the step between
//      blocks (n = 32) and the one-block-early start of the last M-block's pairs (32 - n).
// fp64 sincos of the exact products, one rounding to fp32: a wave's start state is one complex product per pair.
constexpr int ABX2_T1_TILE_FLOATS = 64;
constexpr int ABX2_T2_FLOATS = 33 * 64;
__global__ void abx2_rope_start_kernel(const float* __restrict__ inv_freq, int tile_first, int ntiles, float* __restrict__ t1,
                                       float* __restrict__ t2) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n1 = (int64_t)ntiles * 32;
  if (idx >= n1 + 33 * 32) return;
  const int64_t row = idx < n1 ? idx / 32 : (idx - n1) / 32;          // tile, or in-block offset n
  const int e = (int)(idx % 32);
  const int hi = e >> 4, q = e & 15;
  const double f = (double)inv_freq[4 * (q >> 1) + 2 * (q & 1) + hi];
  const double pos = idx < n1 ? (double)(tile_first + row) * 128.0 : (double)row;
  double sn, cn;
  sincos(pos * f, &sn, &cn);
  float* dst = idx < n1 ? t1 + idx * 2 : t2 + (idx - n1) * 2;
  dst[0] = (float)cn;
  dst[1] = (float)sn;
}

constexpr int ABX2_RED_STRIDE = 8 * 4 * TL;   // floats per partial-sum slot: [8 waves][4 heads][TL]
constexpr int abx2_smem(int nks) { return 3 * TL * 32 * nks + 3 * ABX2_RED_STRIDE * (int)sizeof(float) + 2 * nks * 1024; }

typedef __attribute__((address_space(3))) float lds_f32;

// ACC: 0 = scores rounded to fp16 and stored to `out`; 1 / 2 = this launch is one pass over a column WINDOW of a wider
// rank (x and the fragments of that window): fp32 partial scores stored to (1) / added to (2) p.acc (abx_rope_kernel's ACC);
// 3 = the last window: p.acc + this window's scores, rounded to fp16 and stored to `out`.
// p.ncols != 0: a last window of p.ncols < 16 NKS valid columns run at the full width -- its fragments carry zero rows past
// p.ncols, the fp16 rows over-read into the next row (finite values x 0; the buffer range ends at the last row's true end).
template <int NKS, int QBITS = 0, int ACC = 0>
__global__ __launch_bounds__(NTHREADS, 2) void abx_rope2_kernel(AbxParams p) {
  using Geo = LdsGeom<NKS>;
  constexpr int NRING = 3;
  constexpr int WB = NKS * 1024;                   // bytes of one W image: [ks][hiA 2][m 32][8 fp16]
  constexpr int NS2 = NKS / 2;                     // stage-2 k-steps per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned smem_lds = (unsigned)reinterpret_cast<uintptr_t>(smem);
  const unsigned red_base = smem_lds + (unsigned)(NRING * Geo::TILE_BYTES);
  const unsigned w_base = red_base + (unsigned)(3 * ABX2_RED_STRIDE * sizeof(float));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hi = lane >> 5;
  const int g = blockIdx.x % p.G;
  const int cidx = blockIdx.x / p.G;

  // tile range of this workgroup (abx_rope_kernel: full tiles dealt evenly, the partial tail tile to the last workgroup)
  const int nt_full = p.L / TL;
  const int base = nt_full / p.nch, rem = nt_full % p.nch;
  const int tile0 = cidx * base + min(cidx, rem);
  const bool has_tail = (p.L % TL) != 0 && cidx == p.nch - 1;
  const int ntile = base + (cidx < rem ? 1 : 0) + (has_tail ? 1 : 0);
  if (ntile <= 0) return;
  const int tail_nb = has_tail ? (p.L % TL + 31) / 32 : 4;

  // ---- tile staging: identical to abx_rope_kernel (LDS-DMA with the XOR swizzle in the lane's source offset / packed codes
  //      dequantised in registers)
  const h16* xg = p.x + (int64_t)g * p.sx_g;
  constexpr int RPP = NTHREADS / Geo::CPR;
  u32x4 xrs;
  {
    const unsigned long long xb = reinterpret_cast<unsigned long long>(xg);
    xrs[0] = __builtin_amdgcn_readfirstlane((unsigned)xb);
    xrs[1] = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32));
    xrs[2] = __builtin_amdgcn_readfirstlane((unsigned)(((int64_t)(p.L - 1) * p.sx_l + (p.ncols ? p.ncols : 16 * NKS)) * 2));
    xrs[3] = 0x00020000u;
  }
  const unsigned dma_voff = (unsigned)((tid / Geo::CPR) * p.sx_l * 2 + Geo::swz(tid / Geo::CPR, tid % Geo::CPR) * 16);
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
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // packed rows: LPR lanes share a row of 16 NKS codes, each loading whole dwords -- a quarter row (4 lanes: every thread of
  // the workgroup stages) except for 3-bit rows of 64 / 32 codes, where 12 bytes = 32 codes are the smallest whole-dword
  // piece: 2 lanes (threads 0..255) / 1 lane (threads 0..127) per row
  constexpr int LPR = (QBITS == 3 && NKS < 8) ? NKS / 2 : 4;
  constexpr int CPQ = 16 * NKS / LPR;
  constexpr int NW = QBITS ? (CPQ * QBITS) / 32 : 1;
  static_assert(QBITS == 0 || (CPQ * QBITS) % 32 == 0, "the packed piece of a lane must be whole dwords");
  const bool qactive = tid < TL * LPR;            // (wave-uniform: whole waves)
  unsigned qraw[NW];
  unsigned qmeta = 0;
  const unsigned char* xqg = QBITS ? p.xq + (int64_t)g * p.sq_g : nullptr;
  const h16* xmg = QBITS ? p.xmeta + (int64_t)g * p.sm_g : nullptr;
  if (QBITS != 0 && p.qgroup > 0) {
    // rows quantised in column groups (quantize_tensor with group_size > 0, quant.py:11-13): the (scale, zero) pair of THIS
    // lane's piece -- qgroup is a multiple of 32, a piece (8 .. 32 columns at a multiple of its width) never straddles two
    // groups; a padded last window's pieces past the row's end take the last valid pair (their fragment rows are 0)
    int c = p.qcol0 + (tid % LPR) * CPQ;
    if (p.ncols) c = min(c, p.qcol0 + p.ncols - 1);
    xmg += 2 * (c / p.qgroup);
  }
  auto load_q_into = [&](int tt, unsigned (&qraw)[NW], unsigned& qmeta) {
    if (!qactive) return;
    const int row = tid / LPR, quarter = tid % LPR;
    const int l = min((tile0 + tt) * TL + row, p.L - 1);
    // (a padded last window, p.ncols valid columns: the quarters past the row's end re-read quarter 0 -- their fragment rows are 0)
    const unsigned* src = reinterpret_cast<const unsigned*>(xqg + (int64_t)l * p.sq_l) + (p.ncols && quarter * CPQ >= p.ncols ? 0 : quarter * NW);
#pragma unroll
    for (int k = 0; k < NW; ++k) qraw[k] = __builtin_nontemporal_load(src + k);
    qmeta = *reinterpret_cast<const unsigned*>(xmg + (int64_t)l * p.sm_l);
  };
  auto store_q_from = [&](int slot, const unsigned (&qraw)[NW], const unsigned& qmeta) {
    if (!qactive) return;
    const int row = tid / LPR, quarter = tid % LPR;
    const h16x2 m2 = __builtin_bit_cast(h16x2, qmeta);
    const h16x2 scale2 = h16x2{m2[0], m2[0]};
    const h16 nb = -((h16)1024.f + m2[1]);
    const h16x2 negbias2 = h16x2{nb, nb};
    char* dst = smem + slot * Geo::TILE_BYTES + row * Geo::RB;
#pragma unroll
    for (int gq = 0; gq < CPQ / 8; ++gq) {
      unsigned grp;
      if (QBITS == 4) {
        grp = qraw[gq % NW];
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
        const unsigned pw = 0x64006400u | c0 | (c1 << 16);
        h16x2 v = __builtin_bit_cast(h16x2, pw);
        v = (v + negbias2) * scale2;
        o[e] = __builtin_bit_cast(unsigned, v);
      }
      const int c = quarter * (CPQ / 8) + gq;
      *reinterpret_cast<u32x4*>(dst + Geo::swz(row, c) * 16) = o;
    }
  };
  auto load_q = [&](int tt) { load_q_into(tt, qraw, qmeta); };
  auto store_q = [&](int slot) { store_q_from(slot, qraw, qmeta); };

  // ---- prologue: small loads first (they feed ~300 VALU operations that depend on nothing else), then the bulk
  float fr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) fr[j] = p.inv_freq[4 * w + 2 * j + hi];
  const float psimax = 64.0f * p.inv_freq[ABX2_I0];
  // the query of this group: [4 heads][128] fp16, one element per thread, parked in LDS as (q_i, q_{i+64}) pairs (W image 1
  // is free until the first tile's stage 1)
  const unsigned qbuf = w_base + (unsigned)WB;
  {
    const int hh = tid >> 7, d = tid & 127;
    const h16 v = p.a[(int64_t)(g * 4 + hh) * p.sa_h + (int64_t)d * p.sa_d];
    *(__attribute__((address_space(3))) h16*)(uintptr_t)(qbuf + (unsigned)(((hh * 64 + (d & 63)) * 2 + (d >> 6)) * 2)) = v;
  }

  // packed rows: the first TWO tiles are requested at once (a second register set for the prologue) and stored behind the
  // fragment loads and the RoPE start below -- one HBM round trip in front of the fragment requests instead of two in a row
  unsigned qrawB[NW];
  unsigned qmetaB = 0;
  if (QBITS == 0) {
    dma_tile(0, 0);
    dma_tile(min(1, ntile - 1), 1);
  } else {
    load_q(0);
    load_q_into(min(1, ntile - 1), qrawB, qmetaB);
  }

  // weight fragments
  const u32x4* bh_base = p.bfrag2 + ((int64_t)(g * 8 + w) * NKS) * 64 + lane;
  h16x8 bf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    u32x4 v = bh_base[(int64_t)ks * 64];
    bf[ks] = *reinterpret_cast<h16x8*>(&v);
  }
  const bool s1_wave = w < NKS;                   // stage 1: wave w = r-block w
  const u32x4* bl_base = p.bfrag2 + (int64_t)p.G * 8 * NKS * 64 + ((int64_t)(g * NKS + (s1_wave ? w : 0)) * 8) * 64 + lane;
  h16x8 lowf[4][2];
#pragma unroll
  for (int hh = 0; hh < 4; ++hh)
#pragma unroll
    for (int cs = 0; cs < 2; ++cs) {
      u32x4 v = bl_base[(int64_t)(hh * 2 + cs) * 64];
      lowf[hh][cs] = *reinterpret_cast<h16x8*>(&v);
    }
  // coefficient fragments of this workgroup's first two tiles
  const u32x4* tab = p.rope_tab + ((int64_t)(p.tab_tile0 + tile0) * 2) * 32 + (lane >> 4) * 8 + (lane & 7);
  const bool coef_lane = (lane & 15) < ABX2_K;    // MFMA columns 8..15 of stage 1 are zero: those lanes load nothing
  auto load_coef = [&](int tt, h16x8 (&cf)[2]) {
    const int t = min(tt, ntile - 1);
#pragma unroll
    for (int cs = 0; cs < 2; ++cs) {
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (coef_lane) v = tab[(int64_t)(t * 2 + cs) * 32];
      cf[cs] = *reinterpret_cast<h16x8*>(&v);
    }
  };
  h16x8 cf0[2], cfC[2];
  load_coef(0, cf0);
  load_coef(1, cfC);

  // RoPE state of this lane: position n of a block, pairs i = 4w + 2j + hi; started one block early (pipeline warm-up)
  float rc[2], rs[2], cs_[2], sn[2];
  float lf = (float)(p.pos0 + tile0 * TL + n - 32);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sincos_exact_product(lf, fr[j], &sn[j], &cs_[j]);
    sincos_exact_product(32.0f, fr[j], &rs[j], &rc[j]);
  }
  // stage-2 polynomial weights of this lane: block s = w & 3 of a tile, position d = 32 s + n, terms k = 4 hi + c
  float pw[4];
  {
    const float tau = (float)(2 * (32 * (w & 3) + n) + 1 - TL) * (1.0f / TL);
    const float t = tau * psimax;
    const float t2 = t * t;
    pw[0] = hi ? t2 * t2 * (1.0f / 24.0f) : 1.0f;
    pw[1] = pw[0] * t * (hi ? 0.2f : 1.0f);
    pw[2] = pw[1] * t * (hi ? (1.0f / 6.0f) : 0.5f);
    pw[3] = pw[2] * t * (hi ? (1.0f / 7.0f) : (1.0f / 3.0f));
  }

  if (QBITS != 0) {
    store_q(0);
    store_q_from(1, qrawB, qmetaB);
    load_q(min(2, ntile - 1));
  }
  __syncthreads();                                // query and zeroed partial sums visible

  // ---- fold the query into the fragments (v_dot2_f32_f16: both products exact in fp32, one rounding to fp16)
  {
    const int m = lane & 31;
    const int hh = 2 * ((m >> 3) & 1) + (m & 1);
    const int i = 4 * w + 2 * (m >> 4) + ((m >> 2) & 1);
    const int u = (m >> 1) & 1;
    const unsigned qp = *(const __attribute__((address_space(3))) unsigned*)(uintptr_t)(qbuf + (unsigned)((hh * 64 + i) * 4));
    const h16x2 q2 = __builtin_bit_cast(h16x2, qp);
    h16x2 coef;
    coef[0] = u ? -q2[0] : q2[0];
    coef[1] = q2[1];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4 own = __builtin_bit_cast(u32x4, bf[ks]);
      u32x4 res;
      unsigned da[8], db[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned ow = own[e];
        unsigned par = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ow, 0x4E, 0xF, 0xF, false);   // lane ^ 2: the other of (d, d + 64)
        da[2 * e] = __builtin_amdgcn_perm(par, ow, 0x05040100u);
        da[2 * e + 1] = __builtin_amdgcn_perm(par, ow, 0x07060302u);
        db[2 * e] = db[2 * e + 1] = __builtin_bit_cast(unsigned, coef);
      }
      float dr[8];
      abx2_dot2x8(dr, da, db);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h16x2 r2;
        r2[0] = (h16)dr[2 * e];
        r2[1] = (h16)dr[2 * e + 1];
        res[e] = __builtin_bit_cast(unsigned, r2);
      }
      bf[ks] = __builtin_bit_cast(h16x8, res);
    }
    // low band: a register holds (B[r,i], B[r,i+64]) -> (P[r,i], Q[r,i])
    const int q = lane >> 4;
#pragma unroll
    for (int h4 = 0; h4 < 4; ++h4)
#pragma unroll
      for (int cs = 0; cs < 2; ++cs) {
        const u32x4 qq = *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)(qbuf + (unsigned)((h4 * 64 + ABX2_I0 + 16 * cs + 4 * q) * 4));
        u32x4 own = __builtin_bit_cast(u32x4, lowf[h4][cs]);
        u32x4 res;
        unsigned da[8], db[8];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          // (element -> scalar -> bit_cast: hipcc 7.2 folds __builtin_bit_cast(h16x2, vec[e4]) to element 0 for every e4)
          const unsigned qe = qq[e4], oe = own[e4];
          const h16x2 cp = __builtin_bit_cast(h16x2, qe);                 // (q_i, q_{i+64})
          h16x2 cq;
          cq[0] = cp[1];
          cq[1] = -cp[0];                                                  // (q_{i+64}, -q_i)
          da[2 * e4] = da[2 * e4 + 1] = oe;
          db[2 * e4] = qe;
          db[2 * e4 + 1] = __builtin_bit_cast(unsigned, cq);
        }
        float dr[8];
        abx2_dot2x8(dr, da, db);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          h16x2 r2;
          r2[0] = (h16)dr[2 * e4];
          r2[1] = (h16)dr[2 * e4 + 1];
          res[e4] = __builtin_bit_cast(unsigned, r2);
        }
        lowf[h4][cs] = __builtin_bit_cast(h16x8, res);
      }
  }

  // ---- stage 1: W image of one tile.  D = [P|Q](16 latent columns x 64) . coef(64 x 16 terms): lane (k, q) gets
  //      W_h[k][16 w + 4 q + j]; stored as the A operand of stage 2: [ks = w][hiA = q >> 1][m = 8 h + k][e = 4 (q & 1) + j]
  const unsigned w_st = w_base + (unsigned)(((w * 2 + (lane >> 5)) * 32 + (lane & 15)) * 16 + 8 * ((lane >> 4) & 1));
  auto stage1 = [&](const h16x8 (&cf)[2], int wslot) {
    f32x4 wa[4];
#pragma unroll
    for (int h4 = 0; h4 < 4; ++h4) {
      wa[h4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cs = 0; cs < 2; ++cs) wa[h4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lowf[h4][cs], cf[cs], wa[h4], 0, 0, 0);
    }
    if ((lane & 15) < ABX2_K) {
#pragma unroll
      for (int h4 = 0; h4 < 4; ++h4) {
        h16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (h16)wa[h4][j];
        *(__attribute__((address_space(3))) h16x4*)(uintptr_t)(w_st + (unsigned)(wslot * WB + h4 * 8 * 16)) = o;
      }
    }
  };
  if (s1_wave) stage1(cf0, 0);
  if (p.dbg && blockIdx.x == 0) {
    // debug dump (tools/diag_two_band.py dump): W image of the first tile, this lane's folded low fragments and coefficients
    __syncthreads();
    u32x4* d = reinterpret_cast<u32x4*>(p.dbg);
    d[tid] = *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(w_base + (unsigned)(tid * 16));   // 8 KB (NKS = 8)
    for (int hh = 0; hh < 4; ++hh)
      for (int cs = 0; cs < 2; ++cs) d[512 + (hh * 2 + cs) * 512 + tid] = __builtin_bit_cast(u32x4, lowf[hh][cs]);
    for (int cs = 0; cs < 2; ++cs) d[512 + 8 * 512 + cs * 512 + tid] = __builtin_bit_cast(u32x4, cf0[cs]);
  }

  // scores leave through a buffer store (invalid lanes get an out-of-range offset the hardware drops)
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);

  u32x4 arsrc;      // fp32 score array of a multi-pass launch (same [head][position] indexing; range drops the invalid lanes)
  {
    const unsigned long long ab = reinterpret_cast<unsigned long long>(p.acc);
    arsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)ab);
    arsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(ab >> 32));
    arsrc[2] = __builtin_amdgcn_readfirstlane((unsigned)((((int64_t)(p.H - 1) * p.acc_ld + p.L) * 4)));
    arsrc[3] = 0x00020000u;
  }
  const __amdgpu_buffer_rsrc_t arsrc_ld = __builtin_amdgcn_make_buffer_rsrc(
      p.acc, 0, p.acc ? (int)(((int64_t)(p.H - 1) * p.acc_ld + p.L) * 4) : 0, 0x00020000);
  // cross-wave reduction of tile tt (partial sums of the 8 waves in slot rslot) and the fp16 store
  // (measured: LDS float atomics -- the two waves of a SIMD adding into one word -- cost 50 us per launch at C2)
  auto reduce_store = [&](int tt, int rslot) {
    const int hh = tid >> 7, pos = tid & 127;
    const unsigned r = red_base + (unsigned)((rslot * ABX2_RED_STRIDE + hh * TL + pos) * sizeof(float));
    const int l = (tile0 + tt) * TL + pos;
    const bool ok = tt >= 0 && l < p.L;
    float s = 0.f;
    if (ACC == 3)      // last window: the earlier windows' sum (out-of-range lanes read 0)
      s = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                        arsrc_ld, ok ? (unsigned)(((int64_t)(g * 4 + hh) * p.acc_ld + l) * 4) : 0xFFFFFFF0u, 0, 0));
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) s += *(lds_f32*)(uintptr_t)(r + (unsigned)(ww * 4 * TL * sizeof(float)));
    if (ACC == 0 || ACC == 3) {
      const unsigned off = ok ? (unsigned)(((int64_t)(g * 4 + hh) * p.so_h + l) * 2) : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (h16)s), orsrc, off, 0, 0);
    } else {
      // every (head, position) is written once per pass, the passes are stream-ordered: deterministic
      const unsigned aoff = ok ? (unsigned)(((int64_t)(g * 4 + hh) * p.acc_ld + l) * 4) : 0xFFFFFFF0u;
      if (ACC == 1) asm volatile("buffer_store_dword %0, %1, %2, 0 offen\n\ts_nop 0" ::"v"(s), "v"(aoff), "s"(arsrc) : "memory");
      else asm volatile("buffer_atomic_add_f32 %0, %1, %2, 0 offen\n\ts_nop 0" ::"v"(s), "v"(aoff), "s"(arsrc) : "memory");
    }
  };

  // X fragments: ring of XD registers sets, refilled XD k-steps ahead (abx_rope_kernel); local k-step j of this wave is
  // k-step (j + koff) % NKS of the tile
  constexpr int XD = NKS < 4 ? NKS : 4;
  const int koff = w >= 4 ? NKS / 2 : 0;
  h16x8 xf[XD];
  unsigned fa[NKS];
#pragma unroll
  for (int j = 0; j < NKS; ++j) fa[j] = smem_lds + (unsigned)(n * Geo::RB + Geo::swz(n, 2 * ((j + koff) % NKS) + hi) * 16);
  auto read_frag = [&](int i, int blk) {
    return *(const __attribute__((address_space(3))) h16x8*)(uintptr_t)(fa[i] + (unsigned)(blk * 32 * Geo::RB));
  };
  // partial sums: after the half swap lane (n, hi) holds head 2 mb + hi of position n
  const unsigned red_lane = red_base + (unsigned)(((w * 4 + hi) * TL + n) * sizeof(float));
  // stage-2 A operand of this lane: W image row m = n, k-slot half hi, the wave's first k-step koff
  const unsigned w_rd = w_base + (unsigned)(((koff * 2 + hi) * 32 + n) * 16);

  f32x16 accS2;
#pragma unroll
  for (int e = 0; e < 16; ++e) accS2[e] = 0.f;

  // ---- one straight-line region per 32-position block (abx_rope_kernel): the NKS MFMAs of block blk, the epilogue of
  //      the PREVIOUS block (accumulators acP -> partial sums of block eblk in slot erslot), one chunk of 4 VALU operations
  //      per MFMA gap.  S2M: this wave also runs stage 2 of block blk (its NS2 k-steps); S2E: the previous block was this
  //      wave's stage-2 block, its 16 FMAs join the epilogue.
  auto region = [&](auto kind_c, auto last_c, auto s2m_c, auto s2e_c, auto s1_c, f32x16& acN, int blk, int erslot, int eblk,
                    const f32x16& acP, int stt, int sslot, unsigned nd, unsigned wrd, int wnext) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    constexpr bool S2M = decltype(s2m_c)::value;
    constexpr bool S2E = decltype(s2e_c)::value;
    constexpr bool S1 = decltype(s1_c)::value;      // stage 1 of the NEXT tile rides in this region's MFMA gaps
    constexpr int S1PG = 8 / NKS;                   // stage-1 MFMAs per gap (8 per tile and wave)
    f32x4 wa[4];
    if (S1) {
#pragma unroll
      for (int h4 = 0; h4 < 4; ++h4) wa[h4] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int GAPS = NKS;
    constexpr int NC = 8 + (S2E ? 4 : 0);
    constexpr int CPG = (NC + GAPS - 1) / GAPS;
#pragma unroll
    for (int e = 0; e < 16; ++e) acN[e] = 0.f;
    float part[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) part[s] = 0.f;
    float cc = 0.f, ss = 0.f;
    h16x8 wfr[NS2];
    if (S2M && KIND != 3) {
#pragma unroll
      for (int j = 0; j < NS2; ++j)
        wfr[j] = *(const __attribute__((address_space(3))) h16x8*)(uintptr_t)(wrd + (unsigned)(j * 1024));
#pragma unroll
      for (int e = 0; e < 16; ++e) accS2[e] = 0.f;
    }
    auto chunk = [&](int c) {
      if (c < 8) {
        const int j = c >> 2, t = c & 3;
        if (t == 0) {
          // cos/sin at the oracle's fp32-rounded angle fl(l f): exact angle = ang + lo, first order in lo (positions < 2^18)
          const float ang = lf * fr[j];
          const float lo = fmaf(lf, fr[j], -ang);
          cc = fmaf(lo, sn[j], cs_[j]);
          ss = fmaf(-lo, cs_[j], sn[j]);
        } else if (t <= 2) {
          const int h1 = t - 1;
#pragma unroll
          for (int h0 = 0; h0 < 2; ++h0) {
            const float k1 = acP[8 * j + 4 * h1 + h0], k2 = acP[8 * j + 4 * h1 + 2 + h0];
            const int s = 2 * h1 + h0;
            part[s] = fmaf(cc, k1, fmaf(ss, k2, part[s]));
            asm volatile("" : "+v"(part[s]));
          }
        } else {
          const float c2 = fmaf(-sn[j], rs[j], cs_[j] * rc[j]);
          sn[j] = fmaf(cs_[j], rs[j], sn[j] * rc[j]);
          cs_[j] = c2;
          asm volatile("" : "+v"(cs_[j]), "+v"(sn[j]));
        }
      } else {
        const int hh = c - 8;           // low band of head hh: terms k = 4 hi + c of this lane
        part[hh] = fmaf(pw[0], accS2[4 * hh], fmaf(pw[1], accS2[4 * hh + 1],
                   fmaf(pw[2], accS2[4 * hh + 2], fmaf(pw[3], accS2[4 * hh + 3], part[hh]))));
        asm volatile("" : "+v"(part[hh]));
      }
    };
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (KIND != 3) {
        acN = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ks], xf[ks % XD], acN, 0, 0, 0);
        asm volatile("" : "+v"(acN));
        if (S2M && ks < NS2) {
          accS2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfr[ks], xf[ks % XD], accS2, 0, 0, 0);
          asm volatile("" : "+v"(accS2));
        }
        if (S1) {
#pragma unroll
          for (int i = 0; i < S1PG; ++i) {
            const int m = ks * S1PG + i;            // (head, c-step) = (m >> 1, m & 1)
            wa[m >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lowf[m >> 1][m & 1], cfC[m & 1], wa[m >> 1], 0, 0, 0);
            asm volatile("" : "+v"(wa[m >> 1]));
          }
        }
      }
#pragma unroll
      for (int q = 0; q < CPG; ++q)
        if (ks * CPG + q < NC) chunk(ks * CPG + q);
      if (KIND != 3) {
        const int r = ks + XD;
        if (r < NKS) {
          xf[ks % XD] = read_frag(r, blk);
          if (LAST) {
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
      if (KIND == 0 && QBITS == 0 && ks % 2 == 1 && ks / 2 < Geo::SPT) dma_piece(stt, sslot, ks / 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    lf += 32.0f;
    // lanes n and n + 32 hold complementary pairs (and polynomial terms) of the same position: one half swap per head PAIR
    const unsigned rdst = red_lane + (unsigned)((erslot * ABX2_RED_STRIDE + eblk * 32) * sizeof(float));
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[2 * mb]), __float_as_uint(part[2 * mb + 1]), false, false);
      *(lds_f32*)(uintptr_t)(rdst + (unsigned)(mb * 2 * TL * sizeof(float))) = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    if (KIND == 0 && QBITS != 0) {
      store_q(sslot);
      load_q(min(stt + 1, ntile - 1));
    }
    if (KIND == 1) reduce_store(stt, sslot);
    if (S1) {
      // W image of the next tile: lane (k, q) holds W_h[k][16 w + 4 q + j] (waves beyond the rank's r-blocks computed on
      // r-block 0 and store nothing)
      if (s1_wave && (lane & 15) < ABX2_K) {
#pragma unroll
        for (int h4 = 0; h4 < 4; ++h4) {
          h16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (h16)wa[h4][j];
          *(__attribute__((address_space(3))) h16x4*)(uintptr_t)(w_st + (unsigned)(wnext * WB + h4 * 8 * 16)) = o;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x16 accA, accB;
#pragma unroll
  for (int e = 0; e < 16; ++e) accB[e] = 0.f;

  dma_wait();
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < XD; ++ks) xf[ks] = read_frag(ks, 0);

  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  using NotLast = std::false_type;
  using Last = std::true_type;
  int s_cur = 0, s_nxt = 1, s_prv = 2;

  auto main_loop = [&](auto s_c) {
    constexpr int S = decltype(s_c)::value;       // this wave's stage-2 block
    // S2M = (block == S), S2E = (block whose epilogue runs == S)
    // stage 1 of the next tile rides in region (S + 2) & 3: neither this wave's stage-2 region nor the one after it, so its
    // 16 accumulator registers can be the (then dead) stage-2 accumulators
#define ABX2_REGION(KIND, LASTT, BLK, ACN, ERSLOT, EBLK, ACP, STT, SSLOT, ND, WRD)                                          \
  region(KIND{}, LASTT{}, std::integral_constant<bool, (BLK) == S>{}, std::integral_constant<bool, (EBLK) == S>{},          \
         std::integral_constant<bool, (BLK) == ((S + 2) & 3)>{}, ACN, BLK, ERSLOT, EBLK, ACP, STT, SSLOT, ND, WRD, wnext)
#define ABX2_REGION_T(KIND, LASTT, BLK, ACN, ERSLOT, EBLK, ACP, STT, SSLOT, ND, WRD)                                        \
  region(KIND{}, LASTT{}, std::integral_constant<bool, (BLK) == S>{}, std::integral_constant<bool, (EBLK) == S>{},          \
         std::false_type{}, ACN, BLK, ERSLOT, EBLK, ACP, STT, SSLOT, ND, WRD, 0)
    const int nmain = tail_nb < 4 ? ntile - 1 : ntile;
    for (int tt = 0; tt < nmain; ++tt) {
      if (tt > 0) {
        dma_wait();
        __syncthreads();
      }
      // the coefficient fragments requested during the previous tile have landed (vmcnt(0) above): make hipcc place its
      // own wait for them HERE and not in front of the stage-1 MFMAs, where it would also wait for this tile's DMA pieces
      asm volatile("" : "+v"(cfC[0]), "+v"(cfC[1]));
      const unsigned nd = (unsigned)((s_nxt - s_cur) * Geo::TILE_BYTES);
      const unsigned wrd = w_rd + (unsigned)((tt & 1) * WB);
      const int wnext = (tt + 1) & 1;             // (a stage 1 past the last tile writes an image nobody reads)
      // the coefficients of tile tt + 2 (its stage 1 runs during tile tt + 1) are requested as soon as this tile's stage 1
      // has consumed the current ones: a whole tile and a barrier's vmcnt(0) ahead of their use
      constexpr int R1 = (S + 2) & 3;
      ABX2_REGION(K0, NotLast, 0, accA, s_prv, 3, accB, min(tt + 2, ntile - 1), s_prv, 0u, wrd);
      if (R1 == 0) load_coef(tt + 2, cfC);
      ABX2_REGION(K1, NotLast, 1, accB, s_cur, 0, accA, tt - 2, s_nxt, 0u, wrd);
      if (R1 == 1) load_coef(tt + 2, cfC);
      ABX2_REGION(K2, NotLast, 2, accA, s_cur, 1, accB, 0, 0, 0u, wrd);
      if (R1 == 2) load_coef(tt + 2, cfC);
      ABX2_REGION(K2, Last, 3, accB, s_cur, 2, accA, 0, 0, nd, wrd);
      if (R1 == 3) load_coef(tt + 2, cfC);
      const int t3 = s_prv;
      s_prv = s_cur;
      s_cur = s_nxt;
      s_nxt = t3;
    }
    int drain_slot = s_prv, drain_blk = 3;
    if (tail_nb < 4) {
      // partial tail tile: only its 32-row blocks (abx_rope_kernel); its W image was built during the previous tile
      const int tt = ntile - 1;
      if (tt > 0) {
        dma_wait();
        __syncthreads();
      }
      const unsigned wrd = w_rd + (unsigned)((tt & 1) * WB);
      ABX2_REGION_T(K2, NotLast, 0, accA, s_prv, 3, accB, 0, 0, 0u, wrd);
      if (tail_nb >= 2) ABX2_REGION_T(K1, NotLast, 1, accB, s_cur, 0, accA, tt - 2, s_nxt, 0u, wrd);
      else reduce_store(tt - 2, s_nxt);
      if (tail_nb == 3) ABX2_REGION_T(K2, NotLast, 2, accA, s_cur, 1, accB, 0, 0, 0u, wrd);
      if (tail_nb != 2) accB = accA;
      drain_slot = s_cur;
      drain_blk = tail_nb - 1;
      const int t3 = s_prv;
      s_prv = s_cur;
      s_cur = s_nxt;
      s_nxt = t3;
    }
    // drain: epilogue of the very last block (with this wave's stage-2 terms if that block is S)
    if (drain_blk == S) region(K3{}, NotLast{}, std::false_type{}, std::true_type{}, std::false_type{}, accA, 0, drain_slot, drain_blk, accB, 0, 0, 0u, 0u, 0);
    else region(K3{}, NotLast{}, std::false_type{}, std::false_type{}, std::false_type{}, accA, 0, drain_slot, drain_blk, accB, 0, 0, 0u, 0u, 0);
#undef ABX2_REGION
#undef ABX2_REGION_T
  };
  switch (w & 3) {
    case 0: main_loop(std::integral_constant<int, 0>{}); break;
    case 1: main_loop(std::integral_constant<int, 1>{}); break;
    case 2: main_loop(std::integral_constant<int, 2>{}); break;
    default: main_loop(std::integral_constant<int, 3>{}); break;
  }
  dma_wait();
  __syncthreads();
  reduce_store(ntile - 2, s_nxt);
  reduce_store(ntile - 1, s_prv);
}

}  // namespace
