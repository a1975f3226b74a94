// One kernel for the attention core of a decode step: reconstruct-K -> RoPE -> q.K^T (the abx math,
// kernel/abx_rope.py:44-111 / kernel/palu_attention.py:219), /sqrt(D) + softmax (:219, :238) and the latent
// P.V (:246-251), split over L with flash-decoding partials (merged by pv_combine_kernel, decode_pv.hip).
//
// Why one kernel: the score part is matrix-core/issue bound and leaves HBM idle (2 TB/s), the P.V part is a
// pure HBM stream (403 MB at C2) with almost no math; run back to back they add up (62 + 75 us), and two
// kernels cannot share a CU because the score kernel owns the whole register file and LDS.  Here the V rows
// stream UNDER the score MFMAs of later rows:
//   * score pipeline = abx_rope_kernel (abx_rope_kernel.h) on 64-row steps: K-latent tiles by LDS-DMA into a
//     3-slot ring, B^T fragments register-resident with the query folded in, hand-interleaved MFMA / RoPE
//     epilogue regions, cross-wave partial sums through LDS;
//   * softmax: wave h (h < 4) owns head h: it sums the 8 waves' partials of a 64-row tile (one row per lane),
//     applies the reference's fp16 rounding points, keeps a running maximum (online softmax) and publishes
//     fp16 probabilities + the rescale factor through LDS -- no extra barrier, everything rides on the
//     one barrier per step the score pipeline already has (3-step software pipeline: scores(t), softmax(t-2),
//     P.V(t-3));
//   * P.V on the matrix cores (pv_mfma.h): every wave owns a column slice of V (waves 4-7: NTL col-tiles =
//     128 B of each row at Rv=384, waves 0-3: NTS col-tiles = 64 B) for all rows, streams it by LDS-DMA into
//     a wave-private 3-unit ring (no cross-wave synchronisation), reads it back with the hardware transpose
//     read and accumulates D^T = V^T.P^T with v_mfma_f32_16x16x32_f16: no VALU work per V element at all.
// The DMA bookkeeping is a static schedule: every step issues SPT K pieces + 2*NT V pieces per wave, so every
// wait is an s_waitcnt vmcnt(constant) (loads return in order); the kernel performs no stores inside the loop.
#pragma once
#include "abx_rope_kernel.h"
#include "pv_mfma.h"

namespace {

constexpr int FTL = 64;   // cache rows per step

struct FusedParams {
  const h16* a;
  int64_t sa_h, sa_d;
  const u32x4* bfrag;
  const h16* x;
  int64_t sx_g, sx_l;
  const h16* v;
  int64_t sv_g, sv_l;
  const h16* mask;   // [L] additive (kernel/palu_attention.py:229-234) or null
  const float* inv_freq;
  float* part;   // [G][nch][gs][Rv]
  float* ml;     // [G][nch][gs][2]
  int H, G, gs, L, R, Rv, pos0;
  int nch;       // workgroups (L-ranges) per group
  int nt_total;  // 64-row tiles covering L
  float sqrt_d;
  int prio_mode;
  int exp_flags;   // experiments (PALU_FUSED_EXP; results are wrong when set): 1 = no V DMA and every wait is vmcnt(0),
                   // 2 = no P.V reads / MFMAs, 4 = no softmax, 8 = default instead of nt cache policy on the DMAs, 16 = no score blocks
  unsigned long long* dbg;
};

template <int NKS>
struct FGeo {
  static constexpr int CPR = 2 * NKS;
  static constexpr int RB = 32 * NKS;
  static constexpr int TILE_BYTES = FTL * RB;
  static constexpr int SPT = FTL * CPR / NTHREADS;   // K pieces per tile and wave
  static_assert(FTL * CPR % NTHREADS == 0 && SPT >= 1, "fused kernel: R must be 64 or 128");
  static constexpr int RPB = 256 / RB;
  static constexpr int SH = (RPB == 4) ? 2 : (RPB == 2 ? 1 : 0);
  static constexpr int MASK = CPR - 1;
  static __device__ __forceinline__ int swz(int row, int c) { return c ^ ((row >> SH) & MASK); }
};

template <int NKS, int NTS, int NTL>
struct FusedLds {
  using Geo = FGeo<NKS>;
  static constexpr int RED_STRIDE = 8 * 4 * FTL;   // floats per red slot: [8 waves][4 heads][FTL]
  static constexpr unsigned RED_OFF = 3 * Geo::TILE_BYTES;
  static constexpr unsigned P_OFF = RED_OFF + 3 * RED_STRIDE * 4;   // [2][4][FTL] fp16
  static constexpr unsigned AL_OFF = P_OFF + 2 * 4 * FTL * 2;       // [2][4] fp32
  static constexpr unsigned V_OFF = (AL_OFF + 32 + 255) & ~255u;    // wave-private V rings
  static constexpr unsigned TOTAL = V_OFF + 4 * 3 * 1024 * (NTS + NTL);
};

// wave-wide maximum through DPP (row butterflies + row broadcasts), returned wave-uniform
static __device__ __forceinline__ float wave_max_dpp(float v) {
#define PALU_DPP_MAX(CTRL, ROWMASK)                                                                              \
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

template <int N>
static __device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NKS, int NTS, int NTL, bool TIMING = false>
__global__ __launch_bounds__(NTHREADS, 2) void decode_fused_kernel(FusedParams p) {
  using Geo = FGeo<NKS>;
  using Lds = FusedLds<NKS, NTS, NTL>;
  constexpr int NMB = 2;
  constexpr int RED_STRIDE = Lds::RED_STRIDE;
  constexpr int SPT = Geo::SPT;
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

  const int g = blockIdx.x % p.G;
  const int cidx = blockIdx.x / p.G;
  const int base = p.nt_total / p.nch, rem = p.nt_total % p.nch;
  const int tile0 = cidx * base + min(cidx, rem);
  const int T = base + (cidx < rem ? 1 : 0);   // >= 1: the host never launches more ranges than tiles

  // ---- K-latent staging (LDS-DMA, see abx_rope_kernel.h): wave w, piece k -> LDS slots [512k + 64w, +64) of a tile
  constexpr int RPP = NTHREADS / Geo::CPR;
  u32x4 xrs;
  {
    const unsigned long long xb = reinterpret_cast<unsigned long long>(p.x + (int64_t)g * p.sx_g);
    xrs[0] = __builtin_amdgcn_readfirstlane((unsigned)xb);
    xrs[1] = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32));
    xrs[2] = __builtin_amdgcn_readfirstlane((unsigned)(((int64_t)(p.L - 1) * p.sx_l + 16 * NKS) * 2));
    xrs[3] = 0x00020000u;
  }
  const unsigned k_voff = (unsigned)((tid / Geo::CPR) * p.sx_l * 2 + Geo::swz(tid / Geo::CPR, tid % Geo::CPR) * 16);
  const unsigned k_row_bytes = __builtin_amdgcn_readfirstlane((unsigned)(p.sx_l * 2));
  auto k_piece = [&](int tt, int slot, int k) {
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((tile0 + tt) * FTL + k * RPP) * k_row_bytes);
    const unsigned dst = smem_lds + (unsigned)(slot * Geo::TILE_BYTES + (NTHREADS * k + 64 * w) * 16);
    if (p.exp_flags & 8) {
      pvm::dma_piece<false>(dst, k_voff, xrs, soff);
    } else {
      pvm::dma_piece<true>(dst, k_voff, xrs, soff);   // latents are read exactly once per step: non-temporal
    }
  };
#pragma unroll
  for (int k = 0; k < SPT; ++k) k_piece(0, 0, k);

  // ---- V-latent staging: wave-private column slice (pv_mfma.h); the first units are issued during the warm-up
  //      steps, behind the B fragments the first MFMA waits for
  u32x4 vrs;
  {
    const unsigned long long vb = reinterpret_cast<unsigned long long>(p.v + (int64_t)g * p.sv_g);
    vrs[0] = __builtin_amdgcn_readfirstlane((unsigned)vb);
    vrs[1] = __builtin_amdgcn_readfirstlane((unsigned)(vb >> 32));
    vrs[2] = __builtin_amdgcn_readfirstlane((unsigned)(((int64_t)(p.L - 1) * p.sv_l + p.Rv) * 2));
    vrs[3] = 0x00020000u;
  }
  const unsigned v_row_bytes = __builtin_amdgcn_readfirstlane((unsigned)(p.sv_l * 2));
  const bool big = w >= 4;
  const int col0 = big ? (w - 4) * 16 * NTL : 4 * 16 * NTL + w * 16 * NTS;
  const unsigned vring = smem_lds + Lds::V_OFF + (big ? (unsigned)(4 * 3 * 1024 * NTS + (w - 4) * 3 * 1024 * NTL) : (unsigned)(w * 3 * 1024 * NTS));
  const int row_base = tile0 * FTL;          // first cache row of this workgroup's range
  const int last_unit = 2 * T - 1;

  // ---- B fragments
  const u32x4* bf_base = p.bfrag + ((int64_t)(g * 8 + w) * NMB) * NKS * 64 + lane;
  h16x8 bf[NMB][NKS];
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4 v = bf_base[(int64_t)(mb * NKS + ks) * 64];
      bf[mb][ks] = *reinterpret_cast<h16x8*>(&v);
    }
#pragma unroll
  for (int k = 0; k < SPT; ++k) k_piece(min(1, T - 1), 1, k);
  stamp();  // 1

  // ---- RoPE state (one block early: the pipeline runs one discarded epilogue first)
  float fr[4], rc[4], rs[4], cs[4], sn[4];
  float lf = (float)(p.pos0 + tile0 * FTL + n - 32);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    fr[j] = p.inv_freq[8 * w + 2 * j + hi];
    sincos_exact_product(lf, fr[j], &sn[j], &cs[j]);
    sincos_exact_product(32.0f, fr[j], &rs[j], &rc[j]);
    // pinned in front of the first barrier together with the fold below (abx_rope_kernel.h, PALU_ABX_PIN_PROLOGUE):
    // otherwise hipcc sinks this arithmetic behind the barrier, onto every workgroup's critical path
    asm volatile("" : "+v"(sn[j]), "+v"(cs[j]), "+v"(rs[j]), "+v"(rc[j]));
  }
  stamp();  // 2

  // ---- fold the query into the fragments (abx_rope_kernel.h, FOLD)
  {
    const int m = lane & 31;
    const int t = m & 1, u = (m >> 1) & 1, pair = m >> 2;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const int hloc = 2 * mb + t;
      const bool valid = hloc < p.gs;
      const int h = g * p.gs + (valid ? hloc : 0);
      const int i = 8 * w + pair;
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
          const unsigned ow = own[e];
          const unsigned par = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ow, 0x4E, 0xF, 0xF, false);
          const unsigned lo2 = __builtin_amdgcn_perm(par, ow, 0x05040100u);
          const unsigned hi2 = __builtin_amdgcn_perm(par, ow, 0x07060302u);
          const float r0 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, lo2), coef, 0.f, false);
          const float r1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, hi2), coef, 0.f, false);
          h16x2 r2;
          r2[0] = (h16)r0;
          r2[1] = (h16)r1;
          res[e] = __builtin_bit_cast(unsigned, r2);
        }
        bf[mb][ks] = __builtin_bit_cast(h16x8, res);
        asm volatile("" : "+v"(bf[mb][ks]));
      }
    }
  }

  // ---- X fragment prefetch ring (abx_rope_kernel.h)
  constexpr int XD = NKS < 4 ? NKS : 4;
  h16x8 xf[XD];
  unsigned fa[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) fa[ks] = smem_lds + (unsigned)(n * Geo::RB + Geo::swz(n, 2 * ks + hi) * 16);
  auto read_frag = [&](int i, int blk) {
    return *(const __attribute__((address_space(3))) h16x8*)(uintptr_t)(fa[i] + (unsigned)(blk * 32 * Geo::RB));
  };
  const unsigned red_lane = smem_lds + Lds::RED_OFF + (unsigned)(((w * 4 + hi) * FTL + n) * sizeof(float));

  // ---- one 32-row block: MFMAs of block blk (KIND != 3), RoPE epilogue of the PREVIOUS block (acP -> red[erslot],
  //      rows eblk*32..), KIND 0 additionally issues the K pieces of tile stt into ring slot sslot.
  auto region = [&](auto kind_c, auto last_c, f32x16 (&acN)[NMB], int blk, int erslot, int eblk, const f32x16 (&acP)[NMB],
                    int stt, int sslot, unsigned nd) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    constexpr int GAPS = NKS * NMB;
    constexpr int CPP = 2 + NMB;
    constexpr int NC = 4 * CPP;
    constexpr int CPG = (NC + GAPS - 1) / GAPS;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acN[mb][e] = 0.f;
    float part[2 * NMB];
#pragma unroll
    for (int s = 0; s < 2 * NMB; ++s) part[s] = 0.f;
    float cc = 0.f, ss = 0.f;
    auto chunk = [&](int c) {
      const int j = c / CPP, t = c % CPP;
      if (t == 0) {
        // cos/sin at the oracle's fp32-rounded angle fl(l*f): exact angle = ang + lo, second order in lo
        const float ang = lf * fr[j];
        const float lo = fmaf(lf, fr[j], -ang);
        const float hh = 0.5f * lo * lo;
        cc = fmaf(-hh, cs[j], fmaf(lo, sn[j], cs[j]));
        ss = fmaf(-hh, sn[j], fmaf(-lo, cs[j], sn[j]));
        asm volatile("" : "+v"(cc), "+v"(ss));
      } else if (t <= NMB) {
        const int mb = t - 1;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const float k1 = acP[mb][4 * j + h2], k2 = acP[mb][4 * j + 2 + h2];
          const int s = 2 * mb + h2;
          part[s] = fmaf(cc, k1, fmaf(ss, k2, part[s]));
          asm volatile("" : "+v"(part[s]));
        }
      } else {
        const float c2 = fmaf(-sn[j], rs[j], cs[j] * rc[j]);
        sn[j] = fmaf(cs[j], rs[j], sn[j] * rc[j]);
        cs[j] = c2;
        asm volatile("" : "+v"(cs[j]), "+v"(sn[j]));
      }
    };
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) {
        const int gap = ks * NMB + mb;
        if (KIND != 3) {
          acN[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[mb][ks], xf[ks % XD], acN[mb], 0, 0, 0);
          asm volatile("" : "+v"(acN[mb]));
        }
#pragma unroll
        for (int q = 0; q < CPG; ++q)
          if (gap * CPG + q < NC) chunk(gap * CPG + q);
        if (KIND != 3 && mb == NMB - 1) {
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
        if (KIND == 0 && gap % (2 * NMB) == 1 && gap / (2 * NMB) < SPT) k_piece(stt, sslot, gap / (2 * NMB));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    lf += 32.0f;
    const unsigned rdst = red_lane + (unsigned)((erslot * RED_STRIDE + eblk * 32) * sizeof(float));
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[2 * mb]), __float_as_uint(part[2 * mb + 1]), false, false);
      *(__attribute__((address_space(3))) float*)(uintptr_t)(rdst + (unsigned)(mb * 2 * FTL * sizeof(float))) =
          __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x16 accA[NMB], accB[NMB];
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
    for (int e = 0; e < 16; ++e) accB[mb][e] = 0.f;

  // ---- softmax of one 64-row tile by wave h = w < 4 (lane = row): running max / sum, fp16 probabilities to LDS
  float m_run = -INFINITY, l_run = 0.f;
  const float rsd = 1.0f / p.sqrt_d;
  auto softmax_tile = [&](int t, int rslot, int pslot) {
    int lane_o = lane;                 // opaque copy: no loop-invariant address registers (see pv_step)
    asm volatile("" : "+v"(lane_o));
    unsigned r = smem_lds + Lds::RED_OFF + (unsigned)(((rslot * 8) * 4 + w) * FTL * sizeof(float)) + (unsigned)(lane_o * sizeof(float));
    float sc = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww)
      sc += *(const __attribute__((address_space(3))) float*)(uintptr_t)(r + (unsigned)(ww * 4 * FTL * sizeof(float)));
    // abx output is an fp16 tensor; fp16 tensor / python float -> fp32 divide rounded to fp16 (palu_attention.py:219).
    // The divide is a reciprocal multiply with one FMA correction step (the correctly rounded quotient).
    const float sf = (float)(h16)sc;
    float qv = sf * rsd;
    qv = fmaf(fmaf(-qv, p.sqrt_d, sf), rsd, qv);
    h16 x16 = (h16)qv;
    const int l = (tile0 + t) * FTL + lane_o;
    if (p.mask) {
      // additive mask in fp16 like the reference (:229-234).  A compiler-visible load in the middle of the statically
      // counted DMA schedule: it is completed on the spot (vmcnt(0) also drains the older DMA pieces, so every later
      // vm_wait<N> of the schedule still holds, merely over-waits) -- the masked step pays for it, the plain one does not.
      const h16 mv = p.mask[min(l, p.L - 1)];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      x16 = (h16)((float)x16 + (float)mv);
    }
    const float x = l < p.L ? (float)x16 : -INFINITY;
    const float tmax = wave_max_dpp(x);
    const float m_new = fmaxf(m_run, tmax);
    const bool dead = m_new == -INFINITY;
    const float nm2 = -m_new * 1.4426950408889634f;
    const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(fmaf(m_run, 1.4426950408889634f, nm2));
    const float pe = dead ? 0.f : __builtin_amdgcn_exp2f(fmaf(x, 1.4426950408889634f, nm2));
    const h16 p16 = (h16)pe;
    l_run = fmaf(l_run, alpha, (float)p16);
    m_run = m_new;
    *(__attribute__((address_space(3))) h16*)(uintptr_t)(smem_lds + Lds::P_OFF + (unsigned)(((pslot * 4 + w) * FTL + lane_o) * 2)) = p16;
    if (lane_o == 0)
      *(__attribute__((address_space(3))) float*)(uintptr_t)(smem_lds + Lds::AL_OFF + (unsigned)((pslot * 4 + w) * 4)) = alpha;
  };

  stamp();  // 3
  const bool young = w >= 4;
  if (p.prio_mode == 1 && young) __builtin_amdgcn_s_setprio(1);
  vm_wait<0>();
  stamp();  // 4
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < XD; ++ks) xf[ks] = read_frag(ks, 0);

  using K0 = std::integral_constant<int, 0>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  using NotLast = std::false_type;
  using Last = std::true_type;

  // Roles: the two waves of a SIMD (w, w + 4) run their non-score work (softmax, P.V, V staging) at OPPOSITE ends of a
  // step -- waves 0-3 first thing after the barrier, waves 4-7 after their score blocks -- so that while one of them sits
  // in LDS / DMA latencies the other owns the matrix pipe; everything within a step only depends on data published by
  // the barrier that opened it.  The DMA issue order is K, V, V for both roles (static vmcnt schedule).
  auto run = [&](auto nt_c, auto big_c) {
    constexpr int NT = decltype(nt_c)::value;
    constexpr bool BIG = decltype(big_c)::value;
    using VC = pvm::Cfg<NT>;
    f32x4 acc[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    int s_cur = 0, s_nxt = 1, s_prv = 2;   // s % 3, (s + 1) % 3, (s + 2) % 3 == (s - 1) % 3
    int vs = 0;                            // ring slot of the next V unit to consume (u % 3)
    auto pv_step = [&](auto wait_c, int u, int uu, int pslot) {
      constexpr int WAIT = decltype(wait_c)::value;
      // the lane constants of the V path are re-derived here from an opaque copy of the lane id (a dozen VALU ops per
      // unit): kept as loop invariants they push the kernel over its 256 registers, and a scratch reload would cost
      // an s_waitcnt vmcnt(0) in the middle of the DMA pipeline
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const pvm::Lane<NT> ln = pvm::make_lane<NT>(lane_o, col0, v_row_bytes, FTL * 2);
      // (the range's last units: the younger requests the count relies on are re-reads of the last unit, clamped; in the
      //  position-split score kernel such re-reads were seen retiring ahead of an older request on cold launches -- every
      //  request is waited for there: abx_rope3_kernel.h, tools/stress_tail_cold.py)
      if ((p.exp_flags & 1) || u + 3 > last_unit) {
        vm_wait<0>();
      } else {
        vm_wait<WAIT>();                   // this unit's DMA pieces have landed (static issue schedule)
      }
      const unsigned ua = vring + (unsigned)(vs * VC::UB);
      if (!(p.exp_flags & 2)) {
        if (uu == 0) {
          const float al = *(const __attribute__((address_space(3))) float*)(uintptr_t)(smem_lds + Lds::AL_OFF + (unsigned)((pslot * 4 + (lane_o & 3)) * 4));
          if (__builtin_amdgcn_ballot_w64(al != 1.0f)) {   // the running maximum moved (rare after the first tiles)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) acc[ct] *= al;
          }
        }
        pvm::pv_unit<NT>(acc, ln, ua, smem_lds + Lds::P_OFF + (unsigned)((pslot * 4 * FTL + 32 * uu) * 2));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's reads are done before it is refilled
      }
      if (!(p.exp_flags & 1)) {
        if (p.exp_flags & 8) {
          pvm::dma_unit<NT, false>(ln, vrs, ua, row_base + 32 * min(u + 3, last_unit), p.L, v_row_bytes, lane_o);
        } else {
          pvm::dma_unit<NT, true>(ln, vrs, ua, row_base + 32 * min(u + 3, last_unit), p.L, v_row_bytes, lane_o);
        }
      }
      vs = vs == 2 ? 0 : vs + 1;
    };
    // everything of a step that is not the score blocks
    auto side_work = [&](int s) {
      if (s >= 8 && s < 20) stamp();
      if (p.prio_mode == 3) __builtin_amdgcn_s_setprio(1);
      if (!BIG && s >= 2 && s <= T + 1 && !(p.exp_flags & 4)) softmax_tile(s - 2, s_nxt, s & 1);
      if (s >= 3) {
        pv_step(std::integral_constant<int, 2 * NT + 2 * SPT>{}, 2 * (s - 3), 0, (s - 3) & 1);
        pv_step(std::integral_constant<int, 2 * NT + SPT>{}, 2 * (s - 3) + 1, 1, (s - 3) & 1);
      } else if (!(p.exp_flags & 1)) {
        // warm-up: the first three units go out one per step (the B fragments and K tiles come first)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const pvm::Lane<NT> ln = pvm::make_lane<NT>(lane_o, col0, v_row_bytes, FTL * 2);
        pvm::dma_unit<NT, true>(ln, vrs, vring + (unsigned)(s * VC::UB), row_base + 32 * min(s, last_unit), p.L, v_row_bytes, lane_o);
      }
      if (p.prio_mode == 3) __builtin_amdgcn_s_setprio(0);
      if (s >= 8 && s < 20) stamp();
    };
    for (int s = 0; s <= T + 2; ++s) {
      if (s >= 8 && s < 20) stamp();
      if (s > 0) {
        if (s <= 3 || s + 3 >= T || (p.exp_flags & 1)) {   // (the range's last steps: see pv_step)
          vm_wait<0>();
        } else {
          vm_wait<2 * NT>();               // K tile s+1 (issued first thing in step s-1) has landed
        }
        __syncthreads();
      }
      if (s >= 8 && s < 20) stamp();
      const int ktile = s < T ? min(s + 2, T - 1) : T - 1;   // beyond the range: a dead slot, keeps the schedule static
      if (!BIG) {
#pragma unroll
        for (int k = 0; k < SPT; ++k) k_piece(ktile, s_prv, k);
        side_work(s);
      }
      if (p.prio_mode == 2 && young) __builtin_amdgcn_s_setprio(1);
      // ---- first half: score block 0 of tile s | epilogue of block 1 of tile s-1 (| K pieces of tile s+2)
      if (s < T && (p.exp_flags & 16)) {
        if (BIG) {
#pragma unroll
          for (int k = 0; k < SPT; ++k) k_piece(ktile, s_prv, k);
        }
      } else if (s < T) {
        if (BIG) {
          region(K0{}, NotLast{}, accA, 0, s_prv, 1, accB, ktile, s_prv, 0u);
        } else {
          region(K2{}, NotLast{}, accA, 0, s_prv, 1, accB, 0, 0, 0u);
        }
      } else {
        if (s == T) region(K3{}, NotLast{}, accA, 0, s_prv, 1, accB, 0, 0, 0u);
        if (BIG) {
#pragma unroll
          for (int k = 0; k < SPT; ++k) k_piece(ktile, s_prv, k);
        }
      }
      if (p.prio_mode == 2 && young) __builtin_amdgcn_s_setprio(0);
      // ---- second half: score block 1 of tile s | epilogue of block 0 of tile s
      if (s < T && !(p.exp_flags & 16)) {
        const unsigned nd = (unsigned)((s_nxt - s_cur) * Geo::TILE_BYTES);
        region(K2{}, Last{}, accB, 1, s_cur, 0, accA, 0, 0, nd);
      }
      if (BIG) side_work(s);
      const int t3 = s_prv;
      s_prv = s_cur;
      s_cur = s_nxt;
      s_nxt = t3;
    }
    stamp();
    // ---- partial context of this range: D layout lane l, reg j -> latent column 16*ct + 4*(l/16) + j, head l%16
    const int hn = lane & 15, qd = lane >> 4;
    if (hn < p.gs) {
      float* dst = p.part + ((size_t)(g * p.nch + cidx) * p.gs + hn) * p.Rv + col0 + 4 * qd;
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) *reinterpret_cast<f32x4*>(dst + 16 * ct) = acc[ct];
    }
  };
  if (big) {
    run(std::integral_constant<int, NTL>{}, std::true_type{});
  } else {
    run(std::integral_constant<int, NTS>{}, std::false_type{});
  }
  if (w < 4 && w < p.gs) {
    const float S = wave_sum(l_run);
    if (lane == 0) {
      float* ml = p.ml + ((size_t)(g * p.nch + cidx) * p.gs + w) * 2;
      ml[0] = m_run;
      ml[1] = S;
    }
  }
  stamp();
}

}  // namespace
