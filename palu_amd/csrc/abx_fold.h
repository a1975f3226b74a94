// The query fold of the two-band score kernels as a stand-alone step: one wave per (head, RoPE pair) turns the MFMA
// fragments of B (abx2_prepare_b_kernel's layout) into the fragments of
//     P[r,i] = q_i B[r,i] + q_{i+64} B[r,i+64],     Q[r,i] = q_{i+64} B[r,i] - q_i B[r,i+64]
// (the weight the reference's `_abx_fwd` multiplies the rotated key with, kernel/abx_rope.py:79-111, with the query moved
// onto B; abx_rope2_kernel.h's header has the algebra).  Round 5 ran this fold inside the position-split score kernel: each
// of its 32 workgroups per latent group folded the same query into the same B (18.1 k of 73.9 k cycles per
// wave were prologue, 32 MB of fragment reads and 32 x the fold's VALU work for 1 MB of distinct result).  Here the wave
// that owns rows (i, i+64) of head h -- in decode_qkv_kernel that is the wave which has just produced the rotated q_i,
// q_{i+64} -- folds the 2 x 16 NKS values of its pair: 512 B of B in, 512 B out at R = 128; the kernel boundary is the
// hand-off, and abx_rope3_kernel<.., PREFOLD> streams the folded fragments straight into LDS (LDS-DMA).
//
// Arithmetic is the in-kernel fold's, instruction for instruction (v_dot2_f32_f16: exact products, one fp32 rounding, then
// one rounding to fp16; same operand order): the folded fragments, and with them the scores, are bit-identical.
//
// Folded-fragment buffer ("qfold"), per latent group g 16 NKS KB:
//   high [mb 8][ks NKS][lane 64] u32x4 : A operand of v_mfma_f32_32x32x16_f16 of M-block mb, TRUE k-step order (the rotation
//        of the source layout for M-blocks 4-7 is undone here), row m of lane = m + 32 hiA as in abx2_prepare_b_kernel with
//        u = 0 -> P, u = 1 -> Q;
//   low  [rb NKS][h 4][cs 2][lane 64] u32x4 : the source's layout, the fp16 pair (B[r,i], B[r,i+64]) replaced by (P[r,i], Q[r,i]).
#pragma once
#include "palu_common.h"

namespace {

// Eight dot products r[e] = a[e].x * b[e].x + a[e].y * b[e].y in fp32 (exact products, one rounding) as ONE asm block.
// hipcc turns the builtin with a zero accumulator into v_mov + v_dot2c_f32_f16 + hazard padding (2.7 instructions per dot in the
// folds); the VOP3P form takes the inline 0.  On gfx950 a DOT result is not interlocked against the next VALU read (3 wait
// states; an asm v_dot2 followed directly by its consumer returns garbage -- measured, round 5) and hipcc pads only the DOTs
// it can see: the block itself ends 3 wait states after its last DOT, so whatever follows is safe.
static __device__ __forceinline__ void abx2_dot2x8(float (&r)[8], const unsigned (&a)[8], const unsigned (&b)[8]) {
  asm("v_dot2_f32_f16 %0, %8, %16, 0\n\t"
      "v_dot2_f32_f16 %1, %9, %17, 0\n\t"
      "v_dot2_f32_f16 %2, %10, %18, 0\n\t"
      "v_dot2_f32_f16 %3, %11, %19, 0\n\t"
      "v_dot2_f32_f16 %4, %12, %20, 0\n\t"
      "v_dot2_f32_f16 %5, %13, %21, 0\n\t"
      "v_dot2_f32_f16 %6, %14, %22, 0\n\t"
      "v_dot2_f32_f16 %7, %15, %23, 0\n\t"
      "s_nop 2"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
        "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
}

constexpr int ABX_FOLD_I0 = 32;                    // first pair of the low band (ABX2_I0)

inline size_t abx_fold_u32x4_per_group(int nks) { return (size_t)16 * nks * 64; }

// What a wave needs of B to fold pair i of head (g, hh): requested before the wave's GEMV, used behind it.
struct AbxFoldSrc {
  u32x4 v;                                         // high: this lane's 16-byte chunk; low: v[0], v[1] = the (B_i, B_{i+64}) pairs of r = lane, lane + 64
};

// high band: lane t < 4 nks holds the chunk (u = t & 1, hiA = (t >> 1) & 1, source k-step j = t >> 2)
static __device__ __forceinline__ int abx_fold_hi_row(int hh, int i, int u) {
  return (hh & 1) + 2 * u + 4 * (i & 1) + 8 * (hh >> 1) + 16 * ((i >> 1) & 1);
}

static __device__ __forceinline__ AbxFoldSrc abx_fold_load(const u32x4* __restrict__ bfrag2, int G, int nks, int g, int hh, int i,
                                                           int lane) {
  AbxFoldSrc s;
  s.v = u32x4{0u, 0u, 0u, 0u};
  if (i < ABX_FOLD_I0) {
    if (lane < 4 * nks) {
      const int u = lane & 1, hiA = (lane >> 1) & 1, j = lane >> 2;
      const int mb = i >> 2;
      s.v = bfrag2[((int64_t)(g * 8 + mb) * nks + j) * 64 + abx_fold_hi_row(hh, i, u) + 32 * hiA];
    }
  } else {
    const int ii = i - ABX_FOLD_I0;
    const int cs = ii >> 4, q = (ii >> 2) & 3, e4 = ii & 3;
    const unsigned* low = reinterpret_cast<const unsigned*>(bfrag2 + (int64_t)G * 8 * nks * 64);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int r = lane + 64 * k;
      if (r < 16 * nks) {
        const int rb = r >> 4, m16 = r & 15;
        s.v[k] = low[((((int64_t)(g * nks + rb) * 4 + hh) * 2 + cs) * 64 + m16 + 16 * q) * 4 + e4];
      }
    }
  }
  return s;
}

// qa = q_i, qb = q_{i+64} (the rotated query as the score kernel reads it: the fp16 values of q_out)
static __device__ __forceinline__ void abx_fold_store(const AbxFoldSrc& s, u32x4* __restrict__ qfold, int nks, int g, int hh, int i,
                                                      h16 qa, h16 qb, int lane) {
  u32x4* dst = qfold + (int64_t)g * 16 * nks * 64;
  if (i < ABX_FOLD_I0) {
    const int u = lane & 1, hiA = (lane >> 1) & 1, j = lane >> 2;
    const int mb = i >> 2;
    h16x2 coef;
    coef[0] = u ? -qa : qa;
    coef[1] = qb;
    unsigned da[8], db[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned ow = s.v[e];
      const unsigned par = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ow, 0xB1, 0xF, 0xF, false);   // lane ^ 1: the (d, d + 64) partner row
      da[2 * e] = __builtin_amdgcn_perm(par, ow, 0x05040100u);
      da[2 * e + 1] = __builtin_amdgcn_perm(par, ow, 0x07060302u);
      db[2 * e] = db[2 * e + 1] = __builtin_bit_cast(unsigned, coef);
    }
    float dr[8];
    abx2_dot2x8(dr, da, db);
    u32x4 res;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h16x2 r2;
      r2[0] = (h16)dr[2 * e];
      r2[1] = (h16)dr[2 * e + 1];
      res[e] = __builtin_bit_cast(unsigned, r2);
    }
    if (lane < 4 * nks) {
      const int ks = (j + (mb >= 4 ? nks / 2 : 0)) % nks;              // (abx2_prepare_b_kernel: M-blocks 4-7 are stored half a turn ahead)
      dst[(int64_t)(mb * nks + ks) * 64 + abx_fold_hi_row(hh, i, u) + 32 * hiA] = res;
    }
  } else {
    const int ii = i - ABX_FOLD_I0;
    const int cs = ii >> 4, q = (ii >> 2) & 3, e4 = ii & 3;
    h16x2 cp, cq;
    cp[0] = qa;
    cp[1] = qb;                                                          // (q_i, q_{i+64})
    cq[0] = qb;
    cq[1] = -qa;                                                         // (q_{i+64}, -q_i)
    unsigned da[8], db[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      da[2 * k] = da[2 * k + 1] = s.v[k & 1];
      db[2 * k] = __builtin_bit_cast(unsigned, cp);
      db[2 * k + 1] = __builtin_bit_cast(unsigned, cq);
    }
    float dr[8];
    abx2_dot2x8(dr, da, db);
    unsigned* low = reinterpret_cast<unsigned*>(dst + (int64_t)8 * nks * 64);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int r = lane + 64 * k;
      if (r < 16 * nks) {
        const int rb = r >> 4, m16 = r & 15;
        h16x2 r2;
        r2[0] = (h16)dr[2 * k];
        r2[1] = (h16)dr[2 * k + 1];
        low[((int64_t)((rb * 4 + hh) * 2 + cs) * 64 + m16 + 16 * q) * 4 + e4] = __builtin_bit_cast(unsigned, r2);
      }
    }
  }
}

}  // namespace
