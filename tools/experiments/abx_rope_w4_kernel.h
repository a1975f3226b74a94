// abx_rope for the flagship shape (R = 128, 3..4 heads per group, fp16 latents, folded query):
// ONE wave per SIMD with the 512-register budget, instead of two waves sharing 256 each.
//
// Why: the 8-wave kernel (abx_rope_kernel.h) is bound by instruction issue, not by the matrix pipe -- a SIMD
// issues roughly one instruction per 4-5 cycles whatever its type, and two half-size waves pay every
// per-wave overhead twice (X-fragment ds_reads + their waits, the cross-wave reduction, staging, loop
// control).  Here each wave owns FOUR 32-row M-blocks (16 RoPE pairs x 2 x 4 heads = 128 rows of the
// folded B), so one X fragment read feeds 4 MFMAs, the reduction is over 4 waves, and a 32-position block
// carries 32 MFMAs + 128 RoPE VALU + ~30 others (~6 issues per MFMA, the budget one wave can hide).
// The 128 B-fragment registers are MFMA-only operands and live in AGPRs; accumulators and RoPE state in VGPRs.
//
// Everything else (fragment layout produced by abx_prepare_b_kernel, fold, exact-angle RoPE recurrence, LDS
// tile image, buffer-descriptor LDS-DMA staging, red[] protocol) is the 8-wave kernel's; see there.
#pragma once
#include "abx_rope_kernel.h"

namespace {

constexpr int W4_THREADS = 256;

// single VALU ops as volatile asm: program order == issue order
static __device__ __forceinline__ float v_mul(float a, float b) {
  float d;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
static __device__ __forceinline__ float v_fma(float a, float b, float c) {
  float d;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
static __device__ __forceinline__ float v_fnma(float a, float b, float c) {   // -a*b + c
  float d;
  asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
static __device__ __forceinline__ float v_fmsub(float a, float b, float c) {  // a*b - c
  float d;
  asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
static __device__ __forceinline__ void v_fmac(float& acc, float a, float b) {
  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}
constexpr int abx_smem_w4() { return 3 * TL * 256 + 3 * 4 * 4 * TL * (int)sizeof(float); }

template <bool TIMING = false>
__global__ __launch_bounds__(W4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void abx_rope_w4_kernel(AbxParams p) {
  constexpr int NKS = 8;                   // R = 128
  constexpr int NMB4 = 4;                  // M-blocks per wave: mb4 = 2*pp + hp (pair half pp, head pair hp)
  constexpr int NRING = 3;
  constexpr int NWV = 4;
  constexpr int RED_STRIDE = NWV * 4 * TL;
  using Geo = LdsGeom<NKS>;
  constexpr int RPP = W4_THREADS / Geo::CPR;   // 16 rows per staging piece
  constexpr int PIECES = TL / RPP;             // 8 pieces per wave and tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
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
  const int base = p.nt_total / p.nch, rem = p.nt_total % p.nch;
  const int tile0 = cidx * base + min(cidx, rem);
  const int ntile = base + (cidx < rem ? 1 : 0);
  if (ntile <= 0) return;

  // ---- staging (see abx_rope_kernel: buffer descriptor + scalar offsets; lane offset is tile/piece invariant)
  const h16* xg = p.x + (int64_t)g * p.sx_g;
  u32x4 xrs;
  {
    const unsigned long long xb = reinterpret_cast<unsigned long long>(xg);
    xrs[0] = __builtin_amdgcn_readfirstlane((unsigned)xb);
    xrs[1] = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32));
    xrs[2] = __builtin_amdgcn_readfirstlane((unsigned)(((int64_t)(p.L - 1) * p.sx_l + 16 * NKS) * 2));
    xrs[3] = 0x00020000u;
  }
  const unsigned dma_voff = (unsigned)((tid / Geo::CPR) * p.sx_l * 2 + Geo::swz(tid / Geo::CPR, tid % Geo::CPR) * 16);
  const unsigned row_bytes = __builtin_amdgcn_readfirstlane((unsigned)(p.sx_l * 2));
  auto dma_piece = [&](int tt, int slot, int k) {
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((tile0 + tt) * TL + k * RPP) * row_bytes);
    const unsigned dst = (unsigned)(slot * Geo::TILE_BYTES + (W4_THREADS * k + 64 * w) * 16);
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(dst), "v"(dma_voff), "s"(xrs), "s"(soff)
        : "memory");
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
#pragma unroll
  for (int k = 0; k < PIECES; ++k) dma_piece(0, 0, k);
#pragma unroll
  for (int k = 0; k < PIECES; ++k) dma_piece(min(1, ntile - 1), 1, k);

  // ---- B fragments: W4 wave w, block (pp, hp) = the 8-wave layout's wave 2w+pp, M-block hp
  h16x8 bf[NMB4][NKS];
#pragma unroll
  for (int mb = 0; mb < NMB4; ++mb) {
    const u32x4* src = p.bfrag + ((int64_t)((gb * 8 + 2 * w + (mb >> 1)) * 2 + (mb & 1))) * NKS * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4 v = src[ks * 64];
      bf[mb][ks] = *reinterpret_cast<h16x8*>(&v);
    }
  }
  stamp();  // 1

  // ---- RoPE state: pair q = 4*pp + j is i = 16w + 8pp + 2j + hi; started one block early (pipeline warm-up)
  float fr[8], rc[8], rs[8], cs[8], sn[8];
  float lf = (float)(p.pos0 + tile0 * TL + n - 32);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    fr[q] = p.inv_freq[16 * w + 8 * (q >> 2) + 2 * (q & 3) + hi];
    sincos_exact_product(lf, fr[q], &sn[q], &cs[q]);
    sincos_exact_product(32.0f, fr[q], &rs[q], &rc[q]);
  }
  stamp();  // 2

  // ---- fold the query into the fragments (row m of an M-block: t = m&1 head, u = (m>>1)&1, pair = m>>2)
  {
    const int m = lane & 31;
    const int t = m & 1, u = (m >> 1) & 1, pair = m >> 2;
#pragma unroll
    for (int mb = 0; mb < NMB4; ++mb) {
      const int hloc = hb * 4 + 2 * (mb & 1) + t;
      const bool valid = hloc < p.gs;
      const int h = g * p.gs + (valid ? hloc : 0);
      const int i = 16 * w + 8 * (mb >> 1) + pair;
      const h16 qi = valid ? p.a[h * p.sa_h + i * p.sa_d] : (h16)0.f;
      const h16 qj = valid ? p.a[h * p.sa_h + (i + 64) * p.sa_d] : (h16)0.f;
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
          unsigned par = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ow, 0x4E, 0xF, 0xF, false);  // lane^2
          unsigned lo2 = __builtin_amdgcn_perm(par, ow, 0x05040100u);
          unsigned hi2 = __builtin_amdgcn_perm(par, ow, 0x07060302u);
          float r0 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, lo2), coef, 0.f, false);
          float r1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, hi2), coef, 0.f, false);
          h16x2 r2;
          r2[0] = (h16)r0;
          r2[1] = (h16)r1;
          res[e] = __builtin_bit_cast(unsigned, r2);
        }
        bf[mb][ks] = __builtin_bit_cast(h16x8, res);
        asm volatile("" : "+a"(bf[mb][ks]));   // from here on one opaque, aligned AGPR quad per fragment
      }
    }
  }

  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const unsigned red_base = smem_lds + (unsigned)(NRING * Geo::TILE_BYTES);

  // cross-wave reduction of tile tt (partials in red[rslot]): thread = (heads s and s+2, position)
  auto reduce_store = [&](int tt, int rslot) {
    const int s0 = tid >> 7, pos = tid & 127;
    unsigned r = red_base + (unsigned)((rslot * RED_STRIDE + s0 * TL + pos) * sizeof(float));
    asm volatile("" : "+v"(r));
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWV; ++ww) {
      acc0 += *(const __attribute__((address_space(3))) float*)(uintptr_t)(r + (unsigned)((ww * 4) * TL * sizeof(float)));
      acc1 += *(const __attribute__((address_space(3))) float*)(uintptr_t)(r + (unsigned)((ww * 4 + 2) * TL * sizeof(float)));
    }
    const int l = (tile0 + tt) * TL + pos;
    const int okl = (tt >= 0) & (l < p.L);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int hloc = hb * 4 + s0 + 2 * e;
      const int ok = okl & (hloc < p.gs);
      const unsigned off = ok ? (unsigned)(((int64_t)(g * p.gs + hloc) * p.so_h + l) * 2) : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (h16)(e ? acc1 : acc0)), orsrc, off, 0, 0);
    }
  };

  constexpr int XD = 4;
  h16x8 xf[XD];
  unsigned fa[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) fa[ks] = smem_lds + (unsigned)(n * Geo::RB + Geo::swz(n, 2 * ks + hi) * 16);
  auto read_frag = [&](int i, int blk) {
    return *(const __attribute__((address_space(3))) h16x8*)(uintptr_t)(fa[i] + (unsigned)(blk * 32 * Geo::RB));
  };
  const unsigned red_lane = red_base + (unsigned)(((w * 4 + hi) * TL + n) * sizeof(float));

  // residual of pair 0 at the block the first (discarded) epilogue covers
  float lo_next = fmaf(lf, fr[0], -(lf * fr[0]));

  // region: 32 MFMAs of block blk (4 M-blocks x 8 k-steps) interleaved with the epilogue of the previous block,
  // one 4-op chunk per MFMA gap: per pair q: [coefficients] [heads 0,1] [heads 2,3] [advance]
  auto region = [&](auto kind_c, auto last_c, f32x16 (&acN)[NMB4], int blk, int erslot, int eblk,
                    const f32x16 (&acP)[NMB4], int stt, int sslot, unsigned nd) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    // The epilogue is written instruction by instruction (inline asm keeps the order): a lone wave pays the VALU
    // result latency (~8 cycles dependent vs ~5 independent issue, tools/ubench_issue.hip), so every op's
    // producer is at least two VALU slots (or one MFMA) upstream: the angle/residual of pair q+1 is computed
    // between the cos/sin of pair q; heads run as two interleaved chains; the advance as mul,mul,fma,fma.
    float part[4];
    float cc = 0.f, ss = 0.f;
    const float lfn = lf + 32.0f;
    auto chunk = [&](int c) {
      const int q = c >> 2, t = c & 3;
      const int pp = q >> 2, j = q & 3;
      if (t == 0) {
        const int qn = (q + 1) & 7;
        const float lfx = (q == 7) ? lfn : lf;       // pair 0 of the NEXT block
        float ang;
        ss = v_fnma(lo_next, cs[q], sn[q]);          // first order in lo: sin(a + lo) = s - ... see abx_rope_kernel.h
        ang = v_mul(lfx, fr[qn]);
        cc = v_fma(lo_next, sn[q], cs[q]);
        lo_next = v_fmsub(lfx, fr[qn], ang);
      } else if (t <= 2) {
        const int hp = t - 1, mb = 2 * pp + hp;
        const int s0 = 2 * hp, s1 = 2 * hp + 1;
        if (q == 0) {
          part[s0] = v_mul(ss, acP[mb][4 * j + 2]);
          part[s1] = v_mul(ss, acP[mb][4 * j + 3]);
        } else {
          v_fmac(part[s0], ss, acP[mb][4 * j + 2]);
          v_fmac(part[s1], ss, acP[mb][4 * j + 3]);
        }
        v_fmac(part[s0], cc, acP[mb][4 * j + 0]);
        v_fmac(part[s1], cc, acP[mb][4 * j + 1]);
      } else {
        // advance the exact-angle state by 32 positions: (c, s) <- (c*rc - s*rs, s*rc + c*rs)
        const float m1 = v_mul(cs[q], rc[q]);
        const float m2 = v_mul(sn[q], rc[q]);
        const float c2 = v_fnma(sn[q], rs[q], m1);
        sn[q] = v_fma(cs[q], rs[q], m2);
        cs[q] = c2;
      }
    };
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int mb = 0; mb < NMB4; ++mb) {
        const int gap = ks * NMB4 + mb;
        if (KIND != 3) {
          // MFMA by inline asm: pins the operand banks (A fragments in AGPRs, accumulators in VGPRs -- left to
          // itself hipcc parks the accumulators in AGPRs and copies every element back for the VALU epilogue)
          // and the issue order.  Hazards are met by construction: the epilogue first reads an accumulator
          // >= 4 MFMA gaps (> 19 wait states) after its last MFMA; dependent MFMAs are 4 gaps apart.
          if (ks == 0)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acN[mb]) : "a"(bf[mb][ks]), "v"(xf[ks % XD]));
          else
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acN[mb]) : "a"(bf[mb][ks]), "v"(xf[ks % XD]));
        }
        chunk(gap);
        if (KIND != 3 && mb == NMB4 - 1) {
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
        if (KIND == 0 && mb == 1) dma_piece(stt, sslot, ks);   // 8 pieces, one per k-step
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    lf = lfn;
    const unsigned rdst = red_lane + (unsigned)((erslot * RED_STRIDE + eblk * 32) * sizeof(float));
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[2 * hp]), __float_as_uint(part[2 * hp + 1]), false, false);
      *(__attribute__((address_space(3))) float*)(uintptr_t)(rdst + (unsigned)(hp * 2 * TL * sizeof(float))) =
          __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    if (KIND == 1) reduce_store(stt, sslot);
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x16 accA[NMB4], accB[NMB4];
#pragma unroll
  for (int mb = 0; mb < NMB4; ++mb)
#pragma unroll
    for (int e = 0; e < 16; ++e) accB[mb][e] = 0.f;

  stamp();  // 3
  dma_wait();
  stamp();  // 4
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
  for (int tt = 0; tt < ntile; ++tt) {
    stamp();
    if (tt > 0) {
      dma_wait();
      __syncthreads();
    }
    stamp();
    const unsigned nd = (unsigned)((s_nxt - s_cur) * Geo::TILE_BYTES);
    region(K0{}, NotLast{}, accA, 0, s_prv, 3, accB, min(tt + 2, ntile - 1), s_prv, 0u);
    region(K1{}, NotLast{}, accB, 1, s_cur, 0, accA, tt - 2, s_nxt, 0u);
    region(K2{}, NotLast{}, accA, 2, s_cur, 1, accB, 0, 0, 0u);
    region(K2{}, Last{}, accB, 3, s_cur, 2, accA, 0, 0, nd);
    const int t3 = s_prv;
    s_prv = s_cur;
    s_cur = s_nxt;
    s_nxt = t3;
  }
  region(K3{}, NotLast{}, accA, 0, s_prv, 3, accB, 0, 0, 0u);
  stamp();
  dma_wait();
  __syncthreads();
  reduce_store(ntile - 2, s_nxt);
  reduce_store(ntile - 1, s_prv);
  stamp();
}

}  // namespace
