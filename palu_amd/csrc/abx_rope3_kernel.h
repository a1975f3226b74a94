// Position-split form of the two-band score kernel: the fused  K = X.B -> RoPE -> q.K^T  of abx_rope2_kernel.h (same
// algebra, same fragments, same coefficient table, same numerics contract: oracle `torch_abx`, kernel/abx_rope.py:152-171;
// replaces the Triton `_abx_fwd`, kernel/abx_rope.py:79-111) with the work split by POSITION instead of by RoPE pair.
//
// abx_rope2_kernel gives each of its 8 waves 4 of the 32 high-band pairs of every position: every wave reads every X
// fragment from LDS (one ds_read_b128 + wait per MFMA), the per-position sum over the pairs is a cross-wave LDS
// reduction behind a per-tile workgroup barrier, and two half-size waves per SIMD pay every per-wave overhead twice
// (417 issued instructions per wave and tile for 44 MFMAs, 62 % of the wave time waiting).
//
// Here a workgroup is 4 waves, ONE per SIMD, each with the 512-register budget:
//   * a wave owns whole 32-position blocks (a contiguous range of 128-position tiles) for ALL 32 high-band pairs, all 4
//     heads and the low band.  The 8 x NKS folded high-band A fragments (256 registers at R = 128) are MFMA-only operands
//     and live in AGPRs for the whole kernel; the block's NKS X fragments are read from LDS ONCE and feed 9 M-blocks
//     (8 high + the low band's stage 2): one ds_read_b128 per 9 MFMAs instead of one per MFMA.
//   * a lane holds one position; the sum over the pairs is a chain of FMAs in that lane (two accumulators per head), one
//     v_permlane32_swap per head pair finishes it and the fp16 scores leave through a buffer store: no cross-wave
//     reduction, no partial sums in LDS.
//   * nothing in the main loop is shared between waves: every wave stages its own X blocks (LDS-DMA into a private
//     two-slot ring, the slot is free as soon as its fragments are in registers, so a block has two block times to land),
//     builds its own per-tile low-band weights W (stage 1, from the folded [P|Q]_low fragments all waves share read-only in
//     LDS) and keeps them in a private 8 KB image.  No barrier after the prologue.
//   * the prologue is cooperative: every wave folds the query into a quarter of the fragments, the folded fragments go
//     through LDS once (the high ones through the space the rings use later).  Its loads are ordered for in-order return (what
//     comes from HBM is requested last), both folds run in one loop behind their loads, dot products as asm blocks.
//   * the wave's last tile is a peeled copy of the loop body: no stage 1 for a tile that does not exist.  Every block waits
//     for ALL its outstanding requests (round 6: the counted waits of round 5 relied on an ordering that clamped re-reads broke).
// Per 32-position block and wave: 8 NKS + NKS big MFMAs (+ 8 NKS small ones per tile in the tile's last block), 256 + 16
// VALU of rotation work, 2 NKS LDS reads; at R = 128 about 6.3 issued instructions per big MFMA -- what one wave can
// issue in a matrix-pipe slot (tools/ubench_issue.hip).
//
// Inline-asm MFMAs pin the operand banks (A in AGPRs, accumulators in VGPRs) and the issue order; hipcc cannot see the
// hazards of an asm MFMA, they are met by construction: an accumulator is first read by the VALU at least one full MFMA
// (32 cycles) plus four VALU operations after its last MFMA was issued (tools/ubench_mfma_hazard.hip: 6 wait states are
// enough), no MFMA source is VALU-written, dependent MFMAs use the SrcC = vDst forwarding.
#pragma once
#include "abx_rope2_kernel.h"

namespace {

constexpr int ABX3_THREADS = 256;

template <int NKS>
struct Abx3Lds {
  static constexpr int BLK = 32 * 32 * NKS;        // one 32-position X block image: 32 rows of RB = 32 NKS bytes
  static constexpr int LOWF = 8 * NKS * 1024;      // folded [P|Q]_low fragments [rb][h][cs][lane] (shared, read-only after the prologue)
  static constexpr int WIMG = NKS * 1024;          // one W image [ks][hiA 2][m 32][8 fp16]
  static constexpr int HIF = 8 * NKS * 1024;       // folded high fragments [mb][ks][lane] (prologue only)
  static constexpr int OFF_LOWF = 0;
  static constexpr int OFF_X0 = LOWF;              // ring slot 0 of the 4 waves
  static constexpr int OFF_X1 = OFF_X0 + 4 * BLK;  // ring slot 1
  static constexpr int OFF_W = OFF_X1 + 4 * BLK;   // W images of the 4 waves
  static constexpr int OFF_HIF = OFF_X1;           // prologue: aliases [slot 1 | W images] = 8 NKS KB exactly
  static constexpr int OFF_Q = OFF_X0;             // prologue, until the folds are done: the query as (q_i, q_{i+64}) pairs, 1 KB (ring slot 0 is
                                                   // filled behind them)
  static constexpr int TOTAL = OFF_W + 4 * WIMG;
  static_assert(4 * BLK + 4 * WIMG == HIF, "the folded high fragments alias ring slot 1 + the W images");
};
constexpr int abx3_smem(int nks) { return 8 * nks * 1024 + 8 * 32 * 32 * nks + 4 * nks * 1024; }

template <int I, int N, class F>
static __device__ __forceinline__ void abx3_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    abx3_for<I + 1, N>(f);
  }
}

// dbg (TIMING): [workgroup][wave 4][64] s_memtime stamps.  In-kernel fold: 0 start, 1 low fragments requested, 2 query in LDS,
// 3 RoPE state, 4 both folds done, 5 fragments in AGPRs, 6 first W image, 7 first block landed, 8.. start of every block, then
// drain start, end.  PREFOLD: 0 start, 1 low fragments requested, 2 they have landed and the table values are in registers (barriers
// A, A2), 3 high fragments and first block requested + RoPE state, 4 = 3, 5 first W image, 6 high fragments landed (barrier B), 7 fragments in AGPRs (barrier
// C) + second block requested, 8.. blocks.  The stamps use scalar registers only: the TIMING build does not spill.
// PREFOLD (round 6): the folded fragments come from p.qfold (abx_fold.h: written once per launch by the projection kernel's q
// waves or by abx_fold_kernel) instead of being folded from (p.a, p.bfrag2) by every workgroup.  The prologue is then: the
// RoPE start tables (3 loads per wave, shared through LDS) + the folded LOW fragments by LDS-DMA -> wait, barrier -> table
// values into registers, barrier -> the folded HIGH fragments by LDS-DMA (into the ring's space) and the first block's latents
// straight into registers -> the first W image (needs the low fragments only: it runs while the high ones land) -> wait,
// barrier -> LDS -> AGPRs -> barrier -> main loop.  Every wait is a vmcnt(0) (no counted wait on mixed request types), and
// no VALU work, L2 read or LDS pass of the fold is left in the kernel.  The prologue is bound by what a CU takes in (~28 B per
// clock): 64 + 64 KB of folded fragments, 32 KB of latents, 12 KB of tables.
template <int NKS, bool TIMING = false, bool PREFOLD = false>
__global__ __launch_bounds__(ABX3_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void abx_rope3_kernel(AbxParams p) {
  using Geo = LdsGeom<NKS>;
  using M = Abx3Lds<NKS>;
  constexpr int CPS = 8 / NKS;                     // epilogue chunks per MFMA slot (an M-block has NKS slots, its epilogue 8 chunks)
  constexpr int FPW = 2 * NKS;                     // fragments a wave folds (of the 8 NKS high and the 8 NKS low ones)
  constexpr int WD = NKS < 4 ? NKS : 4;            // depth of the stage-2 A-fragment ring
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(smem);
  typedef __attribute__((address_space(3))) h16x8 lds_h16x8;
  typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
  typedef __attribute__((address_space(3))) unsigned lds_u32;
  typedef __attribute__((address_space(3))) h16x4 lds_h16x4;
  typedef __attribute__((address_space(3))) h16 lds_h16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int n = lane & 31;                               // (re-derived behind the main loop, see there)
  const int hi = lane >> 5;
  const int g = blockIdx.x % p.G;
  const int cidx = blockIdx.x / p.G;
  // (TIMING: scalar registers only -- s_memtime into an SGPR pair, s_store_dwordx2 through the scalar cache, written back at
  //  the kernel's end: the R = 128 kernel has no VGPR to spare, round 5's vector-store stamps made the TIMING build spill)
  unsigned stamp_off = 0;
  unsigned long long* const dbg_w = TIMING ? p.dbg + ((size_t)blockIdx.x * 4 + w) * 64 : nullptr;
  auto stamp = [&]() {
    if constexpr (TIMING) {
      unsigned long long t;                         // (no branch: stamps past the 64th wrap around inside the wave's own row)
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
      asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(t), "s"(dbg_w), "s"(stamp_off) : "memory");
      stamp_off = (stamp_off + 8) & (64 * 8 - 1);
    }
  };
  stamp();  // 0

  // ---- tiles of this wave: the full tiles dealt evenly over the 4 nch waves of the group, the partial tail tile to the last
  const int nwv = 4 * p.nch;
  const int wv = cidx * 4 + w;
  const int nt_full = p.L / TL;
  const int base = nt_full / nwv, rem = nt_full % nwv;
  const int tile0 = wv * base + min(wv, rem);
  const bool has_tail = (p.L % TL) != 0 && wv == nwv - 1;
  const int nfull = base + (wv < rem ? 1 : 0);
  const int ntile = nfull + (has_tail ? 1 : 0);
  const int tail_nb = has_tail ? (p.L % TL + 31) / 32 : 0;
  const int nblk = 4 * nfull + tail_nb;            // 32-position blocks of this wave: rows tile0 * 128 + 32 b

  // ---- X staging: LDS-DMA, one 1 KB piece = RPP rows per instruction, the XOR swizzle in the lane's source offset
  const h16* xg = p.x + (int64_t)g * p.sx_g;
  u32x4 xrs;
  {
    const unsigned long long xb = reinterpret_cast<unsigned long long>(xg);
    xrs[0] = __builtin_amdgcn_readfirstlane((unsigned)xb);
    xrs[1] = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32));
    xrs[2] = __builtin_amdgcn_readfirstlane((unsigned)(((int64_t)(p.L - 1) * p.sx_l + 16 * NKS) * 2));
    xrs[3] = 0x00020000u;
  }
  constexpr int RPP = 64 / Geo::CPR;               // rows per piece
  constexpr int NV = NKS >= 4 ? NKS / 2 : 1;       // distinct swizzle phases of a piece (piece k: phase k % NV)
  const unsigned row_bytes = __builtin_amdgcn_readfirstlane((unsigned)(p.sx_l * 2));
  unsigned dvoff[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v)
    dvoff[v] = (unsigned)((lane / Geo::CPR) * p.sx_l * 2 + Geo::swz(v * RPP + lane / Geo::CPR, lane % Geo::CPR) * 16);
  const int row00 = tile0 * TL;
  const int blk_last = max(nblk - 1, 0);
  // piece k of block b into ring slot `slot` (blocks past the wave's range re-read its last block: harmless, keeps the
  // vmcnt bookkeeping uniform)
  auto dma_piece = [&](int b, int slot, int k) {
    const int bb = min(b, blk_last);
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(row00 + 32 * bb + k * RPP) * row_bytes);
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)((slot ? M::OFF_X1 : M::OFF_X0) + w * M::BLK + k * 1024));   // (an SGPR whatever the pressure)
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(dst), "v"(dvoff[k % NV]), "s"(xrs), "s"(soff)
        : "memory");   // (m0 cannot be listed: hipcc treats it as reserved and warns that the clobber is ignored; it never keeps a value in m0 on gfx950)
  };
  // ---- small loads FIRST (loads return in issue order: the query gates the first barrier, behind 32 KB of fragments it
  //      would arrive last): the query, frequencies, the first tile's coefficients
  h16 qv[2] = {(h16)0.f, (h16)0.f};
  if constexpr (!PREFOLD) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + ABX3_THREADS * e;
      qv[e] = p.a[(int64_t)(g * 4 + (idx >> 7)) * p.sa_h + (int64_t)(idx & 127) * p.sa_d];
    }
  }
  // lane (n, hi) holds the 16 high-band pairs i = 4 mb + 2 j + hi, q = 2 mb + j, of one position per block
  float fr[16];
  if constexpr (!PREFOLD) {                        // (PREFOLD: from the LDS table image below -- 16 vector loads less in front of the fragments)
#pragma unroll
    for (int q = 0; q < 16; ++q) fr[q] = p.inv_freq[4 * (q >> 1) + 2 * (q & 1) + hi];
  }
  const float psimax = 64.0f * p.inv_freq[ABX2_I0];
  // stage 1 uses 8 of the 16 MFMA columns: lanes k + 8 load the coefficients of lane k, compute the same W values and store
  // them to the same LDS words (no exec masking, no zero fill)
  const u32x4* tab0 = p.rope_tab + (int64_t)(p.tab_tile0 + tile0) * 64;      // (uniform) this wave's first tile: 2 x 32 u32x4 per tile
  const unsigned tab_lane = (unsigned)(((lane >> 4) * 8 + (lane & 7)) * 16);  // this lane's 16 bytes of a coefficient fragment
  // COOP (PREFOLD, NKS >= 4): the first W images of the workgroup's four waves are built together -- wave w takes NKS / 4 of the
  // r-blocks for ALL four first tiles, so every low fragment is read from LDS once per workgroup instead of once per wave (the
  // build is LDS-bound: 64 KB per wave = 256 KB per CU at 128 B per clock were 2.7 k of the prologue's cycles) -- and needs
  // the coefficients of all four tiles
  constexpr bool COOP = PREFOLD && NKS >= 4;
  constexpr int NCF = COOP ? 4 : 1;
  h16x8 cf0[NCF][2];                               // (COOP: [v4] = the first tile of wave v4, exchanged through LDS below; else [0] = this wave's)
  h16x8 cfown[2];
  {
    // (a wave without tiles takes the coefficients of the launch's first tile: its image is never read)
    const u32x4* tabv = p.rope_tab + (int64_t)(p.tab_tile0 + (ntile > 0 ? tile0 : 0)) * 64;
#pragma unroll
    for (int cs = 0; cs < 2; ++cs) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(tabv + cs * 32) + tab_lane);
      cfown[cs] = __builtin_bit_cast(h16x8, v);
      if constexpr (!COOP) cf0[0][cs] = cfown[cs];
    }
  }
  // exact-angle (cos, sin) of: the wave's first tile start (T1), this lane's offset n, the one-block-early start of M-block 7
  // (its epilogue runs during the NEXT block) and the 32-position step (T2, abx2_rope_start_kernel): 25 loads of 16 bytes per
  // lane (in-kernel fold: requested a few per low-fold step behind the high fragments -- their issue costs the CU's address unit
  // 16 cycles each: 1.6 k cycles per CU that would otherwise sit between the two folds)
  const f32x4* t1p = reinterpret_cast<const f32x4*>(p.rope_t1 + ((int64_t)(p.tab_tile0 + (ntile > 0 ? tile0 : 0)) * 2 + hi) * 32);
  const f32x4* t2n = reinterpret_cast<const f32x4*>(p.rope_t2 + (n * 2 + hi) * 32);
  const f32x4* t2m = reinterpret_cast<const f32x4*>(p.rope_t2 + ((32 - n) * 2 + hi) * 32);
  const f32x4* t2s = reinterpret_cast<const f32x4*>(p.rope_t2 + (32 * 2 + hi) * 32);
  f32x4 vt1[8], vt2[8], vts[8], vtm;
  auto tload = [&](auto i_c) {
    constexpr int i = decltype(i_c)::value;
    if constexpr (i < 8) vt1[i] = t1p[i];
    else if constexpr (i < 16) vt2[i - 8] = t2n[i - 8];
    else if constexpr (i < 24) vts[i - 16] = t2s[i - 16];
    else if constexpr (i == 24) vtm = t2m[7];      // pairs q = 14, 15
  };
  h16x8 bf[8][NKS];                                // the folded high fragments: AGPRs, MFMA-only operands from the prologue's end on
  h16x8 xf[NKS];                                   // the X fragments of the current block

  if constexpr (!PREFOLD) {
    // ---- this wave's share of the fragments first (the bulk: nothing waits for them for a while): high f = mb NKS + j
    //      (memory order [mb][j]), low f = rb 8 + h 2 + cs
    const u32x4* bh_base = p.bfrag2 + ((int64_t)g * 8 * NKS + w * FPW) * 64 + lane;
    const u32x4* bl_base = p.bfrag2 + (int64_t)p.G * 8 * NKS * 64 + ((int64_t)g * NKS * 8 + w * FPW) * 64 + lane;
    // (a CU takes ~25 B per clock of these: a wave sits ~200 cycles on every load it issues once the queue is full.  The low
    //  fragments are requested here; the high ones one per low-fold step below, so that the fold runs while the requests drain
    //  instead of behind all 32 of them)
    u32x4 hraw[FPW], lraw[FPW];
#pragma unroll
    for (int t = 0; t < FPW; ++t) lraw[t] = bl_base[(int64_t)t * 64];

    stamp();  // 1

    // the query to LDS as (q_i, q_{i+64}) pairs
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + ABX3_THREADS * e;
      const int hh = idx >> 7, d = idx & 127;
      *(lds_h16*)(uintptr_t)(lds0 + (unsigned)(M::OFF_Q + ((hh * 64 + (d & 63)) * 2 + (d >> 6)) * 2)) = qv[e];
    }

    stamp();  // 2

    __syncthreads();                                 // #1: the query is in LDS
    asm volatile("" : "+v"(cf0[0][0]), "+v"(cf0[0][1]));   // (hipcc's wait for these loads goes here, where nothing else is in flight)

    // ---- (q_i, q_{i+64}) of this lane's rows in the two high M-blocks it folds
    unsigned qp0, qp1;                               // (two scalars, not an array: the fold below picks one by a run-time bit)
    {
      const int m = lane & 31;
      const int hh = 2 * ((m >> 3) & 1) + (m & 1);
      const int i0 = 4 * (2 * w) + 2 * (m >> 4) + ((m >> 2) & 1);
      qp0 = *(const lds_u32*)(uintptr_t)(lds0 + (unsigned)(M::OFF_Q + (hh * 64 + i0) * 4));
      qp1 = *(const lds_u32*)(uintptr_t)(lds0 + (unsigned)(M::OFF_Q + (hh * 64 + i0 + 4) * 4));
    }
    constexpr int TPS = (25 + FPW - 1) / FPW;        // table loads per fold step
    // ---- the two folds in ONE loop: step t requests high fragment t and its share of the table loads, folds low fragment t
    //      (a register holds (B[r,i], B[r,i+64]) -> (P[r,i], Q[r,i]); v_dot2_f32_f16: exact products, one rounding) and high
    //      fragment t - FD, which was requested FD steps earlier -- the loop is bound by what the CU takes in, and the high fold's
    //      VALU work fills the waits of the low one.  High fold (abx_rope_kernel FOLD): row (pair, u, head) of an M-block:
    //      P = q_i B_i + q_{i+64} B_{i+64} (u = 0), Q = q_{i+64} B_i - q_i B_{i+64} (u = 1); the (d, d + 64) partner row sits in
    //      lane ^ 2.  The query stays readable throughout (ring slot 0; the folded high fragments go to slot 1 + the W region).
    //      (abx_fold.h does the same arithmetic once per launch: the PREFOLD form of this kernel.)
    {
      const int qd = lane >> 4;
      const int u = (lane >> 1) & 1;
      auto fold_low = [&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        const int f = w * FPW + t;
        const int h4 = (f >> 1) & 3, cs2 = t & 1;
        const u32x4 qq = *(const lds_u32x4*)(uintptr_t)(lds0 + (unsigned)(M::OFF_Q + (h4 * 64 + ABX2_I0 + 16 * cs2 + 4 * qd) * 4));
        u32x4 own = lraw[t];
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
        u32x4 res;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          h16x2 r2;
          r2[0] = (h16)dr[2 * e4];
          r2[1] = (h16)dr[2 * e4 + 1];
          res[e4] = __builtin_bit_cast(unsigned, r2);
        }
        *(lds_u32x4*)(uintptr_t)(lds0 + (unsigned)(M::OFF_LOWF + (f * 64 + lane) * 16)) = res;
      };
      auto fold_high = [&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        constexpr int s = t / NKS, j = t % NKS;
        const int mb = 2 * w + s;
        const int ks = (j + (mb >= 4 ? NKS / 2 : 0)) % NKS;              // (abx2_prepare_b_kernel: waves 4-7 store half a turn ahead)
        const h16x2 q2 = __builtin_bit_cast(h16x2, s ? qp1 : qp0);
        h16x2 coef;
        coef[0] = u ? -q2[0] : q2[0];
        coef[1] = q2[1];
        u32x4 own = hraw[t];
        unsigned da[8], db[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned ow = own[e];
          const unsigned par = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ow, 0x4E, 0xF, 0xF, false);   // lane ^ 2
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
        *(lds_u32x4*)(uintptr_t)(lds0 + (unsigned)(M::OFF_HIF + ((mb * NKS + ks) * 64 + lane) * 16)) = res;
      };
      constexpr int FD = FPW / 2;                    // the high fold runs this many steps behind its loads
      abx3_for<0, FPW + FD>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        if constexpr (t < FPW) {
          hraw[t] = bh_base[(int64_t)t * 64];
          abx3_for<0, TPS>([&](auto j_c) { tload(std::integral_constant<int, t * TPS + decltype(j_c)::value>{}); });
          __builtin_amdgcn_sched_barrier(0);
          fold_low(t_c);
        }
        if constexpr (t >= FD) fold_high(std::integral_constant<int, t - FD>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  } else {
    // ---- PREFOLD: tables, then this wave's quarter of the folded LOW fragments (LDS-DMA, 1 KB per instruction: fragment f of
    //      the group's buffer lands at OFF_LOWF + f KB as it lies in memory)
    // (the RoPE start tables go through LDS: a lane needs 25 x 16 bytes of them -- 25 KB of loads per wave, 100 KB per CU,
    //  as much as the fragments, on a prologue that is bound by what a CU takes in (~28 B per clock: profiles/
    //  r06_abx_prefold_timeline.txt) -- but the workgroup only 8.25 KB of T2 + 256 B of T1 per wave: 3 loads per wave)
    // table image: [T2 rows 0..32: 33 x 256 B][T1 row of wave 0..3: 4 x 256 B][inv_freq 0..31: 128 B] in the space of the ring
    // slots, free until the high fragments are requested
    constexpr int TAB = M::OFF_X0;
    constexpr int TAB_T1 = 33 * 256, TAB_F = TAB_T1 + 4 * 256;
    constexpr int TAB_CF = TAB_F + 128;              // COOP: [wave 4][cs 2][lane 64] x 16 B coefficient fragments of the waves' first tiles
    static_assert(TAB_CF + (COOP ? 8 * 1024 : 0) <= 8 * NKS * 1024, "the table image fits the two ring slots");
    u32x4 tq[3];
    {
      const char* t2b = reinterpret_cast<const char*>(p.rope_t2);
      const char* t1b = reinterpret_cast<const char*>(p.rope_t1 + (int64_t)(p.tab_tile0 + (ntile > 0 ? tile0 : 0)) * 64);
      tq[0] = *reinterpret_cast<const u32x4*>(t2b + w * 2048 + lane * 16);
      tq[1] = *reinterpret_cast<const u32x4*>(t2b + w * 2048 + 1024 + lane * 16);
      // lanes 0..15: T2 row 32, 16..31: this wave's T1 row, 32..39: the 32 high-band frequencies, 40..63 mirror 0..15
      const int l4 = lane & 15;
      tq[2] = *reinterpret_cast<const u32x4*>(lane < 16 ? t2b + 8192 + lane * 16 : lane < 32 ? t1b + (lane - 16) * 16
                                              : lane < 40 ? reinterpret_cast<const char*>(p.inv_freq) + (lane - 32) * 16
                                                          : t2b + 8192 + l4 * 16);
    }
    u32x4 qrs;
    {
      const unsigned long long qb = reinterpret_cast<unsigned long long>(p.qfold + (int64_t)g * 16 * NKS * 64);
      qrs[0] = __builtin_amdgcn_readfirstlane((unsigned)qb);
      qrs[1] = __builtin_amdgcn_readfirstlane((unsigned)(qb >> 32));
      qrs[2] = 16 * NKS * 1024;
      qrs[3] = 0x00020000u;
    }
    const unsigned lane16 = (unsigned)(lane * 16);
    auto dma_frag = [&](int src_kb, int dst_off) {
      const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(src_kb * 1024));
      const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)dst_off);
      asm volatile(
          "s_mov_b32 m0, %0\n\t"
          "s_nop 0\n\t"
          "buffer_load_dwordx4 %1, %2, %3 offen lds"
          :
          : "s"(dst), "v"(lane16), "s"(qrs), "s"(soff)
          : "memory");   // (m0 cannot be listed: hipcc treats it as reserved and warns that the clobber is ignored; it never keeps a value in m0 on gfx950)
    };
    // the folded LOW fragments first.  (Requesting everything up front -- low, high, first block -- was measured and is slower:
    // what a CU takes in (~20-28 B per clock) is the bound, all of it landed at 9.8 k cycles instead of the low fragments at 6.4 k
    // with the high ones arriving under the W build, and the main loop started 1.4 k cycles later: profiles/r06_abx_prologue_variants.txt)
#pragma unroll
    for (int t = 0; t < FPW; ++t) dma_frag(8 * NKS + w * FPW + t, M::OFF_LOWF + (w * FPW + t) * 1024);
    stamp();  // 1
    // everything requested so far has landed (tables, coefficients, this wave's low fragments) ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(cfown[0]), "+v"(cfown[1]), "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]));
    if constexpr (COOP) {
#pragma unroll
      for (int cs = 0; cs < 2; ++cs)
        *(lds_h16x8*)(uintptr_t)(lds0 + (unsigned)(TAB + TAB_CF + ((w * 2 + cs) * 64 + lane) * 16)) = cfown[cs];
    } else {
      cf0[0][0] = cfown[0];
      cf0[0][1] = cfown[1];
    }
    {
      const int l4 = lane & 15;
      *(lds_u32x4*)(uintptr_t)(lds0 + (unsigned)(TAB + w * 2048 + lane * 16)) = tq[0];
      *(lds_u32x4*)(uintptr_t)(lds0 + (unsigned)(TAB + w * 2048 + 1024 + lane * 16)) = tq[1];
      *(lds_u32x4*)(uintptr_t)(lds0 + (unsigned)(TAB + (lane < 16 ? 8192 + lane * 16 : lane < 32 ? TAB_T1 + w * 256 + (lane - 16) * 16
                                                            : lane < 40 ? TAB_F + (lane - 32) * 16 : 8192 + l4 * 16))) = tq[2];
    }
    __syncthreads();                                 // A: every wave's low fragments and the tables are in LDS
    {
      typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
      const unsigned tb = lds0 + (unsigned)(TAB + hi * 128);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        vt1[i] = *(const lds_f32x4*)(uintptr_t)(tb + (unsigned)(TAB_T1 + w * 256 + i * 16));
        vt2[i] = *(const lds_f32x4*)(uintptr_t)(tb + (unsigned)(n * 256 + i * 16));
        vts[i] = *(const lds_f32x4*)(uintptr_t)(tb + (unsigned)(32 * 256 + i * 16));
      }
      vtm = *(const lds_f32x4*)(uintptr_t)(tb + (unsigned)((32 - n) * 256 + 7 * 16));
      typedef __attribute__((address_space(3))) float lds_f32s;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        fr[q] = *(const lds_f32s*)(uintptr_t)(lds0 + (unsigned)(TAB + TAB_F + (4 * (q >> 1) + 2 * (q & 1) + hi) * 4));
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(fr[q]));
      if constexpr (COOP) {
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4)
#pragma unroll
          for (int cs = 0; cs < 2; ++cs) {
            cf0[v4][cs] = *(const lds_h16x8*)(uintptr_t)(lds0 + (unsigned)(TAB + TAB_CF + ((v4 * 2 + cs) * 64 + lane) * 16));
            asm volatile("" : "+v"(cf0[v4][cs]));
          }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(vt1[i]), "+v"(vt2[i]), "+v"(vts[i]));
      asm volatile("" : "+v"(vtm));                  // (the values are in registers ...)
    }
    __syncthreads();                                 // A2: ... in every wave: the table image may be overwritten
    stamp();  // 2
    // ... and the folded HIGH fragments follow, into the space of the two ring slots ([mb][ks][lane] as in memory), with the
    // first block's latents behind them -- straight into registers (this lane's B-operand chunks: row n of the block, columns
    // 16 ks + 8 hi; rows past the cache read its last row: their scores are never stored), the ring is not free yet
#pragma unroll
    for (int t = 0; t < FPW; ++t) dma_frag(w * FPW + t, M::OFF_X0 + (w * FPW + t) * 1024);
    {
      const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(xg), 0, (int)xrs[2], 0x00020000);
      const unsigned vo = (unsigned)min(row00 + n, p.L - 1) * row_bytes + (unsigned)(hi * 16);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(vo + (unsigned)(ks * 32)), 0, 0);
        xf[ks] = __builtin_bit_cast(h16x8, v);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- RoPE state of this lane: (cs, sn)[q] = cos, sin of the exact angle of position n of the wave's first block -- one
  //      complex product per pair; M-block 7's pairs start one block early; (rc, rs) = the step of 32 positions
  float rc[16], rs[16], cs[16], sn[16];
  const float lf0 = (float)(p.pos0 + tile0 * TL + n);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float ct = vt1[q >> 1][2 * (q & 1)], st = vt1[q >> 1][2 * (q & 1) + 1];
    rc[q] = vts[q >> 1][2 * (q & 1)];
    rs[q] = vts[q >> 1][2 * (q & 1) + 1];
    if (q < 14) {
      const float cn = vt2[q >> 1][2 * (q & 1)], sq = vt2[q >> 1][2 * (q & 1) + 1];
      cs[q] = fmaf(ct, cn, -(st * sq));
      sn[q] = fmaf(st, cn, ct * sq);
    } else {                                       // angle of (tile start) - (32 - n) f
      const float cm = vtm[2 * (q & 1)], sm = vtm[2 * (q & 1) + 1];
      cs[q] = fmaf(ct, cm, st * sm);
      sn[q] = fmaf(st, cm, -(ct * sm));
    }
  }
  stamp();  // 3

  // ---- LDS addresses of this lane
  // X fragment of k-step ks: row n, 16-byte chunk swz(n, 2 ks + hi) (slot 0; slot 1 = + 4 BLK).  The chunk index is
  // (2 ks + hi) ^ f(n): k-step ks flips bits 5.. of the k-step-0 address, which the row base (a multiple of RB) leaves clear
  const unsigned fa0 = lds0 + (unsigned)(M::OFF_X0 + w * M::BLK + n * Geo::RB + Geo::swz(n, hi) * 16);
  const unsigned lowa = lds0 + (unsigned)(M::OFF_LOWF + lane * 16);                              // + f KB
  const unsigned w_rd = lds0 + (unsigned)(M::OFF_W + w * M::WIMG + (hi * 32 + n) * 16);          // + ks KB
  // stage-1 result of r-block rb, head h: lane (k, qd) holds W_h[k][16 rb + 4 qd + j] -> image [ks = rb][hiA = qd >> 1][m = 8 h + k][e = 4 (qd & 1) + j]
  const unsigned w_st = lds0 + (unsigned)(M::OFF_W + w * M::WIMG + ((lane >> 5) * 32 + (lane & 7)) * 16 + 8 * ((lane >> 4) & 1));
  auto read_x = [&](int ks, int slot) {
    return *(const lds_h16x8*)(uintptr_t)((fa0 ^ (unsigned)(ks << 5)) + (unsigned)(slot * 4 * M::BLK));
  };
  auto read_w = [&](int ks) { return *(const lds_h16x8*)(uintptr_t)(w_rd + (unsigned)(ks * 1024)); };
  auto read_lowf = [&](int f) { return *(const lds_h16x8*)(uintptr_t)(lowa + (unsigned)(f * 1024)); };
  auto write_w = [&](int rb, int h4, const h16x4& v) { *(lds_h16x4*)(uintptr_t)(w_st + (unsigned)(rb * 1024 + h4 * 128)) = v; };

  // ---- W image of the first tile (stage 1, plain form): D = [P|Q](16 latent columns x 64) . coef(64 x 16 terms).  Asm MFMAs
  //      with VGPR accumulators (hipcc puts the builtin's into AGPRs -- all 256 hold fragments here: it spilled two of them to
  //      scratch around this block -- and reads every element back); two fragment sets and two accumulator sets alternate:
  //      r-block rb's fragments are requested one r-block ahead, its results are converted and stored behind the MFMAs of
  //      r-block rb + 1 (a full burst after their own: the asm-MFMA rule of this file), the four first halves of an r-block
  //      before the four second halves (no dependent pair back to back)
  auto build_first_w = [&]() {
    h16x8 lfa[8];
    f32x4 waA[4], waB[4];
    auto load8 = [&](h16x8 (&lf)[8], int rb) {
#pragma unroll
      for (int i = 0; i < 8; ++i) lf[i] = read_lowf(rb * 8 + i);
    };
    auto mfma8 = [&](const h16x8 (&lf)[8], const h16x8 (&cf)[2], f32x4 (&wa)[4]) {
#pragma unroll
      for (int h4 = 0; h4 < 4; ++h4) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(wa[h4]) : "v"(lf[2 * h4]), "v"(cf[0]));
#pragma unroll
      for (int h4 = 0; h4 < 3; ++h4) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(wa[h4]) : "v"(lf[2 * h4 + 1]), "v"(cf[1]));
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 7" : "+v"(wa[3]) : "v"(lf[7]), "v"(cf[1]));
    };
    // (image of wave v4: OFF_W + v4 WIMG; w_st carries this wave's own)
    auto store4 = [&](const f32x4 (&wa)[4], int rb, int v4) {
      const unsigned base_v = w_st + (unsigned)((v4 - w) * M::WIMG);
#pragma unroll
      for (int h4 = 0; h4 < 4; ++h4) {
        h16x4 wpk;
#pragma unroll
        for (int j = 0; j < 4; ++j) wpk[j] = (h16)wa[h4][j];
        *(lds_h16x4*)(uintptr_t)(base_v + (unsigned)(rb * 1024 + h4 * 128)) = wpk;
      }
    };
    if constexpr (COOP) {
      // step s = (r-block rr of this wave's share, target wave v4): 8 MFMAs with that tile's coefficients; the results of step
      // s - 1 are converted and stored behind them (the asm-MFMA rule: a full burst after their own)
      constexpr int RPW = NKS / 4;
      const int rb0 = w * RPW;
      load8(lfa, rb0);
      abx3_for<0, 4 * RPW>([&](auto s_c) {
        constexpr int st = decltype(s_c)::value;
        constexpr int rr = st / 4, v4 = st % 4;
        mfma8(lfa, cf0[v4], (st & 1) ? waB : waA);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (v4 == 3 && rr + 1 < RPW) load8(lfa, rb0 + rr + 1);   // (an MFMA has read its sources long before an LDS load returns)
        if constexpr (st > 0) store4((st & 1) ? waA : waB, rb0 + (st - 1) / 4, (st - 1) % 4);
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the last burst's results (no MFMAs follow to cover them)
      store4(((4 * RPW - 1) & 1) ? waB : waA, rb0 + RPW - 1, 3);
    } else {
      load8(lfa, 0);
      abx3_for<0, NKS>([&](auto rb_c) {
        constexpr int rb = decltype(rb_c)::value;
        mfma8(lfa, cf0[0], (rb & 1) ? waB : waA);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (rb + 1 < NKS) load8(lfa, rb + 1);   // (an MFMA has read its sources long before an LDS load returns)
        if constexpr (rb > 0) store4((rb & 1) ? waA : waB, rb - 1, w);
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the last burst's results (no MFMAs follow to cover them)
      store4(((NKS - 1) & 1) ? waB : waA, NKS - 1, w);
    }
  };
  // every wave takes ALL high fragments from LDS: 8 NKS AGPR quads, MFMA-only operands from here on
  auto take_high = [&](int off) {
    const unsigned hsrc = lds0 + (unsigned)(off + lane * 16);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)             // (LDS -> AGPR directly; the asm order keeps the wait below behind the loads)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(bf[mb][ks]) : "v"(hsrc), "n"((mb * NKS + ks) * 1024));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  if constexpr (!PREFOLD) {
    stamp();  // 4
    __syncthreads();                                 // #2: all folded fragments are in LDS, nobody reads the query any more

    // the first block's latents LAST: a wave's loads return in issue order, and these come from HBM while 256 workgroups ask
    // for theirs at once -- in front of the fragment loads they held every fold back by their latency (and ring slot 0 held the
    // query until here); nothing reads the block before the main loop (>= 5 k cycles from here)
    if (nblk > 0) {
#pragma unroll
      for (int k = 0; k < NKS; ++k) dma_piece(0, 0, k);
    }
    take_high(M::OFF_HIF);
    stamp();  // 5
    __syncthreads();                                 // #3: ring slot 1 and the W images are free

    if (nblk <= 0) return;                           // (no barrier below this line)

#pragma unroll
    for (int k = 0; k < NKS; ++k) dma_piece(1, 1, k);
    build_first_w();
  } else {
    stamp();  // 4
    build_first_w();                                 // (the high fragments land meanwhile)
    stamp();  // 5
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(xf[ks]));
    __syncthreads();                                 // B: every wave's high fragments are in LDS, the W images are complete
    stamp();  // 6
    take_high(M::OFF_X0);
    __syncthreads();                                 // C: the ring is free

    if (nblk <= 0) return;                           // (no barrier below this line)

#pragma unroll
    for (int k = 0; k < NKS; ++k) dma_piece(1, 1, k);
  }
  if constexpr (!PREFOLD) stamp();  // 6

  // scores leave through a buffer store (invalid lanes get an out-of-range offset the hardware drops)
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const unsigned obase = (unsigned)(((int64_t)(g * 4 + hi) * p.so_h + row00 + n) * 2);          // head hi (+ 2 mb), position of block 0
  const unsigned ohead2 = (unsigned)(2 * p.so_h * 2);

  // ---- main-loop state
  h16x8 wfr[WD];
  // two accumulator sets: phase k of block B (k = 0: the low band's stage 2, k = 1..8: high M-blocks 0..7) accumulates into
  // set (9 B + k) & 1 while the epilogue of the phase before reads the other one (36 phases per tile: the roles repeat)
  f32x16 accA, accB;
#pragma unroll
  for (int e = 0; e < 16; ++e) accB[e] = 0.f;      // the first block runs the (empty) epilogue of "block -1"
  float pc[4], ps[4];                              // per head: cos-side and sin-side partial sums of the current block
#pragma unroll
  for (int s = 0; s < 4; ++s) pc[s] = ps[s] = 0.f;
  float lfe = lf0 - 32.0f;                         // position (of lane n) of the block whose epilogues are running
  float ang_n = lfe * fr[14];
  float lo_cur = fmaf(lfe, fr[14], -ang_n);        // residual of the oracle's fp32 angle of the next pair to be rotated
  h16x8 cfr[2];                                    // coefficients of the next tile (stage-1 B operand)
  cfr[0] = cfr[1] = cf0[0][0];

  // chunk c (0..7) of the RoPE epilogue of high M-block mbp held in `ac`; its pairs q = 2 mbp + j, j = 0, 1:
  //   c = 0, 2 (j = 0, 1): cos/sin at the oracle's fp32-rounded angle fl(l f) (exact angle = ang + lo, first order in lo), start of
  //                        the next pair's residual
  //   c = 1, 3: advance the exact-angle state by 32 positions, finish the next residual
  //   c = 4, 5 / 6, 7: the two head pairs of j = 0 / 1 -- the only chunks that read the accumulators: they sit in the second
  //                    half of the M-block's slots, at least one full MFMA after the accumulators' last MFMA whatever hipcc
  //                    hoists inside a slot (with NKS < 8 a slot carries several chunks)
  float cc[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f}, m1[2] = {0.f, 0.f};
  auto epi_chunk = [&](auto mbp_c, auto c_c, const f32x16& ac) {
    constexpr int mbp = decltype(mbp_c)::value, c = decltype(c_c)::value;
    if constexpr (mbp < 0 || mbp > 7) {
      // (instantiated from slots whose run-time guard never lets them through)
    } else if constexpr (c < 4) {
      constexpr int j = c >> 1;
      constexpr int q = 2 * mbp + j, qn = (q + 1) & 15;
      if constexpr ((c & 1) == 0) {
        if (q == 15) lfe += 32.0f;                 // pair 0 comes next: it belongs to the following block
        cc[j] = fmaf(lo_cur, sn[q], cs[q]);
        ss[j] = fmaf(-lo_cur, cs[q], sn[q]);
        ang_n = lfe * fr[qn];
        m1[j] = fmaf(-sn[q], rs[q], cs[q] * rc[q]);   // (the new cosine rides here, the sine is advanced in place in the next
                                                      //  chunk: 5 + 4 VALU operations for the two slots instead of 3 + 6 -- a
      } else {                                        //  slot hides five beside its MFMA -- and one register copy less per pair)
        lo_cur = fmaf(lfe, fr[qn], -ang_n);
        sn[q] *= rc[q];
        sn[q] = fmaf(cs[q], rs[q], sn[q]);
        cs[q] = m1[j];
      }
    } else {
      constexpr int j = (c - 4) >> 1, h1 = (c - 4) & 1;
      pc[2 * h1] = fmaf(cc[j], ac[8 * j + 4 * h1], pc[2 * h1]);
      ps[2 * h1] = fmaf(ss[j], ac[8 * j + 4 * h1 + 2], ps[2 * h1]);
      pc[2 * h1 + 1] = fmaf(cc[j], ac[8 * j + 4 * h1 + 1], pc[2 * h1 + 1]);
      ps[2 * h1 + 1] = fmaf(ss[j], ac[8 * j + 4 * h1 + 3], ps[2 * h1 + 1]);
    }
  };

  // finish block bp (its sums are in pc / ps): lanes n and n + 32 hold complementary pairs and polynomial terms of the same
  // position: one half swap per head pair, then lane (n, hi) holds head 2 mb + hi
  auto finalize = [&](int bp) {
    const bool ok = bp >= 0 && row00 + 32 * bp + n < p.L;
    const unsigned off0 = ok ? obase + (unsigned)(bp * 64) : 0xFFFFFFF0u;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const float p0 = pc[2 * mb] + ps[2 * mb], p1 = pc[2 * mb + 1] + ps[2 * mb + 1];
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
      const float sc = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      const unsigned off = ok ? off0 + (unsigned)mb * ohead2 : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (h16)sc), orsrc, off, 0, 0);
    }
  };

#define ABX3_MFMA_A0(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "a"(A), "v"(B))
#define ABX3_MFMA_A(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "a"(A), "v"(B))
#define ABX3_MFMA_V0(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "v"(A), "v"(B))
#define ABX3_MFMA_V(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define ABX3_MFMA_S0(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(ACC) : "v"(A), "v"(B))
#define ABX3_MFMA_S(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define ABX3_IC(X) std::integral_constant<int, (X)> {}

  // ---- one 32-position block: 9 phases of NKS MFMA slots.  B = block of the tile (ring slot B & 1, polynomial argument,
  //      accumulator roles), b = block of the wave, tnext = tile (of the wave) whose coefficients block 1 requests.
  //      S1: stage 1 of the NEXT tile rides in this block, one r-block (8 small MFMAs) as a burst in front of an M-block's first
  //      MFMA: the big MFMAs of an M-block are a dependent chain on one accumulator, and anything that enters the matrix pipe
  //      between two of them costs ~35 cycles on top of its own (measured: 64 small MFMAs spread one per slot made the block
  //      3300 cycles longer instead of 1024); an M-block's first MFMA starts a new chain, so the burst costs its 128 cycles.
  //      The 8 low fragments of a burst are requested in the last slots of the phase before; its results are rounded and stored
  //      two slots later -- a result of an asm MFMA is never read before the slot AFTER the next one (whatever hipcc hoists
  //      inside a slot, a whole slot with its 32-cycle MFMA lies in between).
  auto block = [&](auto B_c, auto S1_c, auto LAST_c, int b, int tnext) {
    constexpr int B = decltype(B_c)::value;
    constexpr bool S1 = decltype(S1_c)::value;
    constexpr bool LAST = decltype(LAST_c)::value;   // a block of the wave's last tile
    constexpr int SL = B & 1;
    constexpr int PH0 = 9 * B;                       // phase number of this block's stage 2
    constexpr int PSTEP = 8 / NKS;                   // r-block r of stage 1 sits in front of phase 1 + r PSTEP
    constexpr int LSLOTS = NKS < 4 ? NKS : 4;        // its fragments are requested in the last LSLOTS slots of the phase before,
    constexpr int LPS = 8 / LSLOTS;                  // LPS per slot
    stamp();
    // stage-2 polynomial weights of this block: position d = 32 B + n of the tile, terms k = 4 hi + c
    float pw[4];
    {
      int nn = n;
      if constexpr (PREFOLD) {                       // (the lane's row, derived again from the lane id: see behind the main loop)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_and_b32 %0, 31, %0" : "=v"(nn));
      }
      const float tau = (float)(2 * (32 * B + nn) + 1 - TL) * (1.0f / TL);
      const float t = tau * psimax;
      const float t2 = t * t;
      pw[0] = hi ? t2 * t2 * (1.0f / 24.0f) : 1.0f;
      pw[1] = pw[0] * t * (hi ? 0.2f : 1.0f);
      pw[2] = pw[1] * t * (hi ? (1.0f / 6.0f) : 0.5f);
      pw[3] = pw[2] * t * (hi ? (1.0f / 7.0f) : (1.0f / 3.0f));
    }
    h16x8 lf8[8];                                    // stage 1 (S1): the low fragments of the next burst
    f32x4 wa[4];                                     //               its four heads
    // slot `ks` of phase `ph` (0 = stage 2, 1..8 = M-blocks 0..7), after the slot's big MFMA: the stage-1 work of that slot
    auto s1_slot = [&](auto ph_c, auto ks_c) {
      constexpr int ph = decltype(ph_c)::value, ks = decltype(ks_c)::value;
      if (!S1) return;
      // fragments of the burst in front of phase ph + 1
      if (ph + 1 >= 1 && ph + 1 <= 8 && (ph + 1 - 1) % PSTEP == 0 && ks >= NKS - LSLOTS) {
        constexpr int r = (ph + 1 - 1) / PSTEP;
        constexpr int i0 = (ks - (NKS - LSLOTS)) * LPS;
#pragma unroll
        for (int i = 0; i < LPS; ++i) lf8[i0 + i] = read_lowf(r * 8 + i0 + i);
      }
      // results of the burst two slots ago (slot 2 of its phase, or slot 0 of the next phase at NKS = 2)
      constexpr int gs = ph * NKS + ks - 2;          // global slot of the burst
      if (gs >= NKS && gs % NKS == 0 && (gs / NKS - 1) % PSTEP == 0) {
        constexpr int r = (gs / NKS - 1) / PSTEP;
#pragma unroll
        for (int h4 = 0; h4 < 4; ++h4) {
          h16x4 wpk;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) wpk[jj] = (h16)wa[h4][jj];
          write_w(r, h4, wpk);
        }
      }
    };
    // ---- phase 0, the low band's stage 2: acc = W . x; in its gaps the epilogue of the PREVIOUS block's M-block 7
    {
      f32x16& acN = (PH0 & 1) ? accB : accA;
      const f32x16& acP = (PH0 & 1) ? accA : accB;
      abx3_for<0, NKS>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if (ks == 0) ABX3_MFMA_V0(acN, wfr[0], xf[0]);
        else ABX3_MFMA_V(acN, wfr[ks % WD], xf[ks]);
        abx3_for<0, CPS>([&](auto i_c) { epi_chunk(ABX3_IC(7), ABX3_IC(ks * CPS + decltype(i_c)::value), acP); });
        if (ks + WD < NKS) wfr[ks % WD] = read_w(ks + WD);
        s1_slot(ABX3_IC(0), ks_c);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    // ---- phases 1..8, M-blocks 0..7: in the gaps of M-block 0 the previous block is finished and stored, then the polynomial
    //      of this block's low band, and the DMA pieces of block b + 2 go into this block's ring slot (its fragments are all in
    //      registers); in the gaps of M-block mb >= 1 the epilogue of M-block mb - 1
    abx3_for<0, 8>([&](auto mb_c) {
      constexpr int mb = decltype(mb_c)::value;
      f32x16& acN = ((PH0 + 1 + mb) & 1) ? accB : accA;
      const f32x16& acP = ((PH0 + 1 + mb) & 1) ? accA : accB;
      abx3_for<0, NKS>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if (mb == 7 && ks == 0) {
          // block b + 1 must have landed before its fragments are read below.  Round 5 waited with a COUNT here (vmcnt(NKS):
          // "everything but the NKS pieces of block b + 2 requested above"), i.e. relied on LDS-DMA requests retiring in issue
          // order -- and saw that fail on cold launches where the younger requests were clamped re-reads of a partly out-of-range
          // tail block (stale rows in 18 of 240 launches, tools/stress_tail_cold.py; ADVICE r5 found one more such pattern in a
          // non-last tile).  Round 6 measured the alternative the VERDICT asked for: waiting for EVERY request costs 0.0-0.6 %
          // (C2 40.89 -> 41.14 us, R = 64 at 128k 43.80 -> 44.14, R = 32 17.50 -> 17.45, 16k positions 13.46 -> 13.40;
          // profiles/r06_abx_wait0.txt) -- block b + 2's pieces were requested 7 phases (~2 k cycles) earlier and have
          // mostly landed -- so the ordering assumption is gone: vmcnt(0) in every block.  (-DABX3_COUNTED_WAIT restores the
          // counted wait in the tiles before a wave's last one, for A/B runs only.)
#ifdef ABX3_COUNTED_WAIT
          if (LAST || B == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NKS) : "memory");
#else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
          if (B == 2) asm volatile("" : "+v"(cfr[0]), "+v"(cfr[1]));
        }
        constexpr bool BURST = S1 && mb % PSTEP == 0;  // this phase opens with the stage-1 burst of r-block mb / PSTEP
        if (BURST && ks == 0) {
          // 4 heads x (cs 0, cs 1): the four first halves with the phase's four accumulator-free epilogue chunks between them (a
          // lone wave issues nothing while an MFMA waits for the pipe: the 16-cycle shadows would stay empty), then the second
          // halves -- four MFMAs behind their first halves, no dependent pair back to back
          abx3_for<0, 4>([&](auto h_c) {
            constexpr int h4 = decltype(h_c)::value;
            ABX3_MFMA_S0(wa[h4], lf8[2 * h4], cfr[0]);
            if (mb == 0) {
              if (h4 == 0) finalize(b - 1);
            } else {
              epi_chunk(ABX3_IC(mb - 1), h_c, acP);
            }
            __builtin_amdgcn_sched_barrier(0);
          });
#pragma unroll
          for (int h4 = 0; h4 < 3; ++h4) ABX3_MFMA_S(wa[h4], lf8[2 * h4 + 1], cfr[1]);
          // (the burst's last MFMA: 8 wait states, so that a copy hipcc might place behind the burst reads finished results)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 7" : "+v"(wa[3]) : "v"(lf8[7]), "v"(cfr[1]));
        }
        if (mb == 7 && ks == NKS - 1) {
          // the block's last MFMA: hipcc may copy live accumulators at the block's end (loop edges); 11 wait states make
          // the result readable by then
          if (ks == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\ts_nop 7\n\ts_nop 2" : "=&v"(acN) : "a"(bf[mb][ks]), "v"(xf[ks]));
          else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 2" : "+v"(acN) : "a"(bf[mb][ks]), "v"(xf[ks]));
        } else if (ks == 0) {
          ABX3_MFMA_A0(acN, bf[mb][0], xf[0]);
        } else {
          ABX3_MFMA_A(acN, bf[mb][ks], xf[ks]);
        }
        if (mb == 0) {
          abx3_for<0, CPS>([&](auto i_c) {
            constexpr int item = ks * CPS + decltype(i_c)::value;
            if (item == 0 && !BURST) finalize(b - 1);
            if (item >= 4) {                         // (second half of the slots: the stage-2 accumulators are complete)
              constexpr int hh = item - 4;           // low band of head hh: terms k = 4 hi + c of this lane
              pc[hh] = fmaf(pw[0], acP[4 * hh], pw[1] * acP[4 * hh + 1]);
              ps[hh] = fmaf(pw[2], acP[4 * hh + 2], pw[3] * acP[4 * hh + 3]);
            }
          });
          dma_piece(b + 2, SL, ks);
          if (B == 1 && ks == NKS - 1) {
            // coefficients of the next tile (its stage 1 runs in this tile's last block): requested behind this block's DMA
            // pieces; this block's wait covers them (7 phases for a read that may miss every cache)
            const u32x4* src = tab0 + (int64_t)tnext * 64;             // (uniform: scalar base + the lane's offset)
            // (the lane's offset ((lane >> 4) * 8 + (lane & 7)) * 16 is derived again from the lane id right here: kept in a register
            //  across the loop it was one of the two values the R = 128 PREFOLD kernel sent to scratch)
            unsigned tl;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tl));
            tl = ((tl >> 4) * 8 + (tl & 7)) * 16;
            asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:512"
                         : "=&v"(cfr[0]), "=&v"(cfr[1])
                         : "v"(tl), "s"(src)
                         : "memory");
          }
        } else {
          abx3_for<0, CPS>([&](auto i_c) {
            constexpr int c = ks * CPS + decltype(i_c)::value;
            if (!(BURST && c < 4)) epi_chunk(ABX3_IC(mb - 1), ABX3_IC(c), acP);   // (chunks 0..3 of a burst phase ran inside the burst)
          });
        }
        s1_slot(ABX3_IC(mb + 1), ks_c);
        if (mb == 7) {
          xf[ks] = read_x(ks, SL ^ 1);                               // the next block's fragments
          constexpr int wi = ks >= NKS - WD ? ks - (NKS - WD) : 0;
          if (ks >= NKS - WD) wfr[wi] = read_w(wi);                  // and the first stage-2 A fragments
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  // ---- first block: its fragments (PREFOLD: they came straight from memory in the prologue)
  if constexpr (!PREFOLD) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (both: block 1's request may be a re-read of block 0, see the blocks' own waits)
    stamp();  // 7
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) xf[ks] = read_x(ks, 0);
  } else {
    stamp();  // 7
  }
#pragma unroll
  for (int i = 0; i < WD; ++i) wfr[i] = read_w(i);

  // ---- tiles of 4 blocks.  Every tile but the wave's last is full and carries stage 1 of its successor in its last block; the
  //      last one (possibly a partial tail: side exits) builds no image -- a peeled copy, so that the loop body has one shape
  //      (a stage-1 choice INSIDE the loop made hipcc spill; an image nobody reads cost 3.7 % of a 4-tile wave's launch)
  int b = 0, tt = 0;
  for (; tt + 1 < ntile; ++tt) {
    block(ABX3_IC(0), std::false_type{}, std::false_type{}, b, tt + 1);
    block(ABX3_IC(1), std::false_type{}, std::false_type{}, b + 1, tt + 1);
    block(ABX3_IC(2), std::false_type{}, std::false_type{}, b + 2, tt + 1);
    block(ABX3_IC(3), std::true_type{}, std::false_type{}, b + 3, tt + 1);
    b += 4;
  }
  if constexpr (PREFOLD) {
    // the peeled tile reads the lane's row index n directly (the loop's uses are strength-reduced away); kept live across the loop
    // it was the one value of the R = 128 kernel hipcc sent to scratch (256 VGPRs + 256 AGPRs): derive it again from the lane id
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_and_b32 %0, 31, %0" : "=v"(n));
  }
  do {
    block(ABX3_IC(0), std::false_type{}, std::true_type{}, b, tt);
    if (++b == nblk) break;
    block(ABX3_IC(1), std::false_type{}, std::true_type{}, b, tt);
    if (++b == nblk) break;
    block(ABX3_IC(2), std::false_type{}, std::true_type{}, b, tt);
    if (++b == nblk) break;
    block(ABX3_IC(3), std::false_type{}, std::true_type{}, b, tt);
    ++b;
  } while (false);
  // ---- drain: the epilogue of the last block's M-block 7, then its scores
  stamp();
  // (the last block's M-block 7 sits in set (9 B + 8) & 1 = B & 1: accB after an odd B, accA after an even one)
  if ((b - 1) & 1) abx3_for<0, 8>([&](auto c_c) { epi_chunk(ABX3_IC(7), c_c, accB); });
  else abx3_for<0, 8>([&](auto c_c) { epi_chunk(ABX3_IC(7), c_c, accA); });
  finalize(b - 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the re-read pieces of the last blocks: nothing may land in LDS after the wave ends)
  stamp();
  if constexpr (TIMING) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
#undef ABX3_MFMA_A0
#undef ABX3_MFMA_A
#undef ABX3_MFMA_V0
#undef ABX3_MFMA_V
#undef ABX3_MFMA_S0
#undef ABX3_MFMA_S
#undef ABX3_IC
}

}  // namespace
