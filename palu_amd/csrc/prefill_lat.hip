// Prompt attention straight from the LATENT caches (SURVEY 8(f) N1 as the survey wrote it): the prompt branch of
// LlamaPaluAttention.forward (kernel/palu_attention.py:196-257) with
//   * K~ = RoPE(X_k . B_h) rebuilt per 64-position kv tile INSIDE the flash kernel (the reference reconstructs the full keys,
//     :67-77, :199-205, and prefill_attn.hip read them from a [H, kv, D] workspace: 512 MiB at 64k positions), and
//   * the latent values read in the cache's own row-major layout [G, L, Rv] (prefill_attn.hip needs a transposed, zero-padded
//     copy: 384 MiB) through the hardware transpose read,
// so that no transient of a prompt pass grows with the number of cached positions.
//
// Same flash formulation as prefill_attn.hip (S^T = K~ . Q~^T, O^T = V^T . P^T, fp32 online softmax, lane = query), but the
// eight waves of a workgroup (128 queries x one head) are SPLIT BY ROLE instead of by kv half / column half -- the pair kernel
// sits at 251 of its 256 registers at Rv = 384 and has no room for the rebuild's operands:
//   S-waves (0-3, one per SIMD, 32 queries each): the scores of the whole 64-position tile, the online softmax, and the rebuild
//       of a quarter of the NEXT K~ tile -- wave w: kv half w & 1 x RoPE pairs 32 (w >> 1) .. + 31, i.e. 64 rows of B_h^T resident as
//       64 registers of MFMA A fragments (these waves have the registers), the X tile's B fragments from LDS (each read feeds two
//       MFMAs), rotary values prefetched from the rotary cache one phase ahead; fp32 -> fp16 -> rotation with the reference's fp16
//       roundings -> the other K~ tile image in LDS (double-buffered).  Probabilities (fp16, already in B-operand order) and the
//       rescale factor go to LDS;
//   O-waves (4-7, the partner on the same SIMD): O^T += V^T . P^T for all Rv columns (192 accumulator registers), V rows staged
//       by LDS-DMA in half tiles of 32 positions, A operand by ds_read_b64_tr_b16 (32-byte granules XOR-swizzled by row & 3:
//       the four rows of a transpose read fall on different bank groups).  They run half a tile behind the S-waves, so ONE
//       probability buffer serves: each k-step pair is written in one phase and read in the next.
// Both forms of the kernel are bound by LDS bandwidth (~2.8 k cycles of LDS traffic per tile in the pair kernel, of which 1.5 k are
// the P.V A operands); what the rebuild adds is counted in LDS bytes, not MFMAs: X tile in (16 KB) + 32 KB of B-fragment reads +
// 16 KB of K~ writes.  Measured steps (profiles/r06_prefill_lat_variants.txt): rebuild on the S-waves by d quarters with its rotary
// values straight from L2 105 ms at 64k tokens (workspace form: 72-75 ms); rebuild on the O-waves with B^T fragments and rotary rows
// through LDS 105-114 ms (4.2 k cycles of LDS traffic per tile); this form: see there.
// Two workgroup barriers per tile; every DMA is waited for with vmcnt(0) at the barrier that publishes it.
// MFMAs per tile and workgroup: 64 (rebuild) + 64 (scores) + 192 (P.V): the rebuild is the +25 % DESIGN 4.6 estimated.
#include <type_traits>

#include "palu_common.h"
#include "pv_mfma.h"

#ifndef PL_EXP
#define PL_EXP 0   // timing experiments only (results are wrong when set): 1 = no K~ rebuild, 2 = no exponentials, 4 = no P.V MFMAs, 8 = no LDS-DMA staging
#endif

namespace {
unsigned long long* g_pl_timeline = nullptr;

constexpr int PL_THREADS = 512;
constexpr int PL_BM = 128;   // queries per workgroup
constexpr int PL_BN = 64;    // kv positions per tile

struct PfLatParams {
  const h16* q;            // rotated queries [H][Tq][128]
  int64_t sq_h, sq_t;
  const h16* xk;           // latent keys   [G][>= Tk][128]
  int64_t sxk_g, sxk_l;
  const h16* xv;           // latent values [G][>= Tk][Rv]
  int64_t sxv_g, sxv_l;
  const h16* bt;           // B^T [H][128 d][128 r]: row d of head h = the weights that rebuild K[., d]
  const h16* cs;           // [pos][2][64] fp16: cos row, sin row of key position pos (0 .. Tk - 1)
  h16* out;
  int64_t so_t;
  int H, G, gs, Tq, Tk, past, causal;
  float scale_log2;
  int nqt, head_major;
  // packed 4-bit caches (QB = 4; quant.hip's layout): codes [G][>= Tk][R / 2 bytes], meta [G][>= Tk][2] fp16 (scale, zero) per row;
  // bt then holds B^T with the columns of every group of 8 in the order 0 4 1 5 2 6 3 7 (the order the nibbles come out of a dword)
  const unsigned char* kc;
  int64_t skc_g, skc_l;    // bytes
  const h16* km;
  int64_t skm_g, skm_l;    // elements
  const unsigned char* vc;
  int64_t svc_g, svc_l;
  const h16* vm;
  int64_t svm_g, svm_l;
  unsigned long long* dbg;   // -DPL_TIMELINE builds: [wave 8][64] s_memtime stamps of workgroup 0, tiles 16 .. (tools/time_prefill_lat.py)
};

template <int I, int N, class F>
static __device__ __forceinline__ void pl_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pl_for<I + 1, N>(f);
  }
}

typedef __attribute__((address_space(3))) h16x8 lds_h16x8_t;
typedef __attribute__((address_space(3))) h16x4 lds_h16x4_t;
typedef __attribute__((address_space(3))) float lds_f32_t;

template <int NCB, int KSR, int QB = 0>
__global__ __launch_bounds__(PL_THREADS, 1) void prefill_lat_kernel(PfLatParams p) {
  static_assert(QB == 0 || QB == 3 || QB == 4, "fp16 or packed 3- / 4-bit latents");
  static_assert(NCB >= 2 && NCB <= 12 && (NCB % 2 == 0 || QB == 0), "rank_v / G = 32 NCB (packed caches: a multiple of 64)");
  static_assert(QB != 3 || (KSR == 8 && NCB % 4 == 0), "3-bit rows are whole 16-byte chunks at multiples of 128 codes");
  static_assert(KSR == 2 || KSR == 4 || KSR == 8, "rank_k / G = 16 KSR in {32, 64, 128}");
  constexpr int RK = 16 * KSR;
  constexpr int RKB = 2 * RK;                       // bytes of an fp16 X row
  constexpr int CPRX = 2 * KSR;                     // its 16-byte chunks (XOR-swizzled by row & (CPRX - 1))
  constexpr int NXP = PL_BN * RKB / 1024;           // DMA pieces of an fp16 X tile (8 or 16)
  constexpr int QBE = QB ? QB : 4;
  constexpr int RQB = RK * QBE / 8;                 // bytes of a packed X row
  constexpr int CQ = RQB / 16;                      // its 16-byte chunks (2 or 4; 3-bit: 3) = the DMA pieces of a tile's codes
  constexpr int RV = 32 * NCB;
  constexpr int RVB = RV * 2;                       // bytes of a V row
  constexpr int XS_BYTES = PL_BN * 256;             // X tile: 64 rows x 128 fp16 (16 chunks per row, XOR-swizzled by row & 15)
  constexpr int KS_BYTES = PL_BN * 256;             // K~ tile, same geometry
  constexpr int VH_BYTES = 32 * RVB;                // half a V tile: 32 rows, row-major, 32-byte granules XOR-swizzled by row & SWZ
  constexpr int SWZ = (NCB % 2 == 0) ? 3 : 1;       // (a row is whole groups of four granules, or -- odd NCB -- of two: the transpose reads
                                                    //  then meet two-way bank conflicts; such ranks are the small configurations)
  constexpr int NVP = VH_BYTES / 1024;              // DMA pieces of half a V tile
  constexpr int OFF_XS = 0;
  constexpr int OFF_VS = OFF_XS + XS_BYTES;         // (the DMA targets first: LDS offsets below 128 KB)
  constexpr int OFF_KS = OFF_VS + 2 * VH_BYTES;     // two K~ tile images: tile jt in image jt & 1
  constexpr int OFF_PS = OFF_KS + 2 * KS_BYTES;     // [qblk 4][ks 4][lane 64] x 16 B probabilities (B-operand order); k-steps 0, 1 are written in
                                                    // phase alpha and read in phase beta, k-steps 2, 3 written in beta and read in the next alpha
  constexpr int OFF_AL = OFF_PS + 4 * 4096;         // [qblk 4][32] floats: rescale factor of the tile (at the end: the row sums)
  constexpr int OFF_QF = OFF_AL + 4 * 32 * 4;       // [qblk 4][ks 8][lane 64] x 16 B: the Q~ fragments (B operand of the scores) -- the S-wave's
                                                    // registers hold the rebuild's 64 B^T fragments instead
  constexpr int VRB = RV * QBE / 8;                 // bytes of a packed V row
  constexpr int CPV = VRB / 16;                     // its 16-byte chunks
  constexpr int NVCP = (32 * VRB + 1023) / 1024;    // DMA pieces of the codes of half a V tile (32 rows, linear as in memory)
  constexpr int VC_BYTES = NVCP * 1024;             // staging for them, two of them
  constexpr int NVC = NCB / 2;                      // waves that de-quantise a half tile: unit 64 w + lane = 32 codes of one row
  constexpr int OFF_VC = OFF_QF + 32 * 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PL_TIMELINE
  unsigned stamp_off = 0;
  unsigned long long* const dbg_w = p.dbg + w * 64;
  const bool stamp_on = p.dbg != nullptr && blockIdx.x == 0 && blockIdx.y == 0;
  auto stamp = [&](int jt) {                          // scalar registers only (abx_rope3_kernel.h)
    if (stamp_on && jt >= 16 && jt < 24) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
      asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(t), "s"(dbg_w), "s"(stamp_off) : "memory");
      stamp_off = (stamp_off + 8) & (64 * 8 - 1);
    }
  };
#else
  auto stamp = [](int) {};
#endif
  const bool swave = w < 4;
  const int qblk = w & 3;
  const int n = lane & 31, hi = lane >> 5;
  int qt_rev, h;
  if (p.head_major == 2) {   // 8 heads at a time, one per XCD; within them heavy (late) query tiles first (prefill_attn.hip)
    const int id = blockIdx.x;
    qt_rev = (id >> 3) % p.nqt;
    h = ((id >> 3) / p.nqt) * 8 + (id & 7);
  } else {
    qt_rev = p.head_major ? blockIdx.y : blockIdx.x;
    h = p.head_major ? blockIdx.x : blockIdx.y;
  }
  const int qt = p.nqt - 1 - qt_rev;
  const int g = h / p.gs;
  const int qrow = qt * PL_BM + qblk * 32 + n;
  const bool qvalid = qrow < p.Tq;
  const int qpos = p.past + qrow;

  int kv_end = p.Tk;
  if (p.causal) {
    const int last_q = min(p.Tq, (qt + 1) * PL_BM) - 1;
    kv_end = min(p.Tk, p.past + last_q + 1);
  }
  const int njt = (kv_end + PL_BN - 1) / PL_BN;

  // ---- staging by LDS-DMA (buffer_load_dwordx4 ... lds): a wave-instruction fills 64 consecutive 16-byte LDS slots, the
  //      swizzle sits in the per-lane SOURCE offset, tile / piece selection is scalar
  auto make_rsrc = [](const void* base, int64_t bytes) {
    u32x4 r;
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
    r[3] = 0x00020000u;
    return r;
  };
  const h16* xkg = p.xk + (int64_t)g * p.sxk_g;
  const h16* xvg = p.xv + (int64_t)g * p.sxv_g;
  const u32x4 xrs = make_rsrc(xkg, ((int64_t)(p.Tk - 1) * p.sxk_l + RK) * 2);
  const u32x4 vrs = make_rsrc(xvg, ((int64_t)(p.Tk - 1) * p.sxv_l + RV) * 2);
  auto dma = [&](unsigned dst, unsigned voff, const u32x4& rs, unsigned soff) {
#if PL_EXP & 8
    return;                                           // (timing experiment: no staging traffic at all)
#endif
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(dst), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
  };
  // ALL staging is requested by the four S-waves (piece w + 4 i belongs to S-wave w): a CU takes in ~28 B per clock, the wave that
  // issues a request sits in it, and the S-waves have the slack (timeline: the O-wave spent 1.3 k of a 7.3 k-tick tile issuing its
  // five pieces on the workgroup's critical path; profiles/r06_prefill_lat_timeline.txt)
  // X tile: piece p = rows (64 / CPRX) p .. , CPRX chunks each (rows past Tk are outside the descriptor: their scores are masked);
  // 4 pieces are a multiple of CPRX rows, so the swizzle key of a lane's row does not depend on i
  const int xrow_l = (64 / CPRX) * (w & 3) + lane / CPRX;
  const unsigned xvo = (unsigned)((xrow_l * p.sxk_l + (((lane % CPRX) ^ (xrow_l & (CPRX - 1))) << 3)) * 2);
  const unsigned xtile_bytes = __builtin_amdgcn_readfirstlane((unsigned)(PL_BN * p.sxk_l * 2));
  const unsigned xstep_bytes = __builtin_amdgcn_readfirstlane((unsigned)(4 * (64 / CPRX) * p.sxk_l * 2));
  static_assert((4 * (64 / CPRX)) % CPRX == 0, "four pieces of the X tile are whole swizzle periods");
  auto dma_x = [&](int jt) {                          // (xvo depends on w & 3 only: either role can issue it)
#pragma unroll
    for (int i = 0; i < NXP / 4; ++i)
      dma(__builtin_amdgcn_readfirstlane(lds0 + OFF_XS + ((w & 3) + 4 * i) * 1024), xvo, xrs,
          __builtin_amdgcn_readfirstlane((unsigned)jt * xtile_bytes + i * xstep_bytes));
  };
  // packed keys: the tile's codes (64 rows x RK / 2 bytes) as CQ pieces of 64 / CQ rows, waves 0 .. CQ - 1 one each; 16-byte chunk c of
  // row r lands at chunk position c ^ ((r >> 1) & (CQ - 1)) (the rebuild reads 4 bytes per lane and k-step: rows two apart would
  // share a bank)
  const u32x4 xqrs = QB ? make_rsrc(p.kc + (int64_t)g * p.skc_g, (int64_t)(p.Tk - 1) * p.skc_l + RQB) : xrs;
  // (3-bit rows, 3 chunks each: plain row-major -- the rebuild's two dwords per lane and k-step then meet two-way conflicts)
  const int xqc = 64 * (w & 3) + lane;
  const unsigned xqvo = QB == 4 ? (unsigned)((lane / CQ) * (int)p.skc_l + (((lane % CQ) ^ (((lane / CQ) >> 1) & (CQ - 1))) << 4))
                                : (QB == 3 ? (unsigned)((xqc / CQ) * (int)p.skc_l + ((xqc % CQ) << 4)) : 0u);
  auto dma_xq = [&](int jt) {
    if ((w & 3) < CQ)
      dma(__builtin_amdgcn_readfirstlane(lds0 + OFF_XS + (w & 3) * 1024), xqvo, xqrs,
          __builtin_amdgcn_readfirstlane((unsigned)(jt * PL_BN + (QB == 4 ? (64 / CQ) * (w & 3) : 0)) * (unsigned)p.skc_l));
  };
  // packed values: the codes of half a tile, linear ([row][RV / 2 bytes] as in memory): piece w = 16-byte chunks 64 w .. 64 w + 63; the
  // lane that de-quantises chunk q = 64 w + lane (row q / NCB, columns 32 (q % NCB) ..) also fetches that row's (scale, zero)
  constexpr int VQI = 1;                            // piece w belongs to wave w (NVC <= 6: the S-waves and, at rank_v / G = 384, O-waves 4 and 5)
  const u32x4 vqrs = QB ? make_rsrc(p.vc + (int64_t)g * p.svc_g, (int64_t)(p.Tk - 1) * p.svc_l + VRB) : vrs;
  unsigned vmeta_next[VQI], vmeta_cur[VQI];
#pragma unroll
  for (int i = 0; i < VQI; ++i) vmeta_next[i] = vmeta_cur[i] = 0;
  auto vc_issue = [&](int jt, int half) {           // codes of half tile (jt, half) into staging `half` + the rows' metas
    if (QB == 0) return;
#pragma unroll
    for (int i = 0; i < VQI; ++i) {
      const int q = 64 * w + lane;
      if (w < NVCP) {                                 // 16-byte chunk q of the half tile's codes (the last piece may reach past them: its
        const int qc = min(q, 32 * CPV - 1);          //  spare lanes re-read the last chunk into the staging's padding)
        const int crow = min(jt * PL_BN + 32 * half + qc / CPV, p.Tk - 1);   // (rows past Tk re-read row Tk - 1: finite values x probability 0)
        dma(__builtin_amdgcn_readfirstlane(lds0 + OFF_VC + half * VC_BYTES + w * 1024),
            (unsigned)(crow * (int)p.svc_l + (qc % CPV) * 16), vqrs, 0u);
      }
      if (w < NVC) {                                  // (scale, zero) of the row whose unit q this lane de-quantises
        const int row = min(jt * PL_BN + 32 * half + q / NCB, p.Tk - 1);
        vmeta_next[i] = *reinterpret_cast<const unsigned*>(p.vm + (int64_t)g * p.svm_g + (int64_t)row * p.svm_l);
      }
    }
  };
  // (a + nb) * s on pairs: a = 0x6400 | code = 1024 + code exactly, nb = -(1024 + zero): the difference is exact, one rounding in the
  // product -- unpack_dequant's arithmetic (quant.hip; quantize_tensor's `(q - zero) * scale` in fp16, quant.py:37-39)
  auto deq2 = [](unsigned w2, h16x2 nb2, h16x2 sc2) {
    h16x2 v = __builtin_bit_cast(h16x2, w2 | 0x64006400u);
    v = (v + nb2) * sc2;
    return __builtin_bit_cast(unsigned, v);
  };
  // 3-bit codes: the pair (code at bit lo, code at bit hi) of a 24-bit group as two fp16 1024 + code, de-quantised like deq2
  auto deq3 = [](unsigned v, int lo, int hi, h16x2 nb2, h16x2 sc2) {
    const unsigned a = ((v >> lo) & 7u) | 0x64006400u;
    const unsigned b = (v >> hi) & 7u;
    h16x2 x = __builtin_bit_cast(h16x2, a | (b << 16));
    x = (x + nb2) * sc2;
    return __builtin_bit_cast(unsigned, x);
  };
  auto vc_dequant = [&](int half) {                 // staging `half` -> fp16 row-major image slot `half` (granule-swizzled like the DMA form)
    if (QB == 0 || w >= NVC) return;
#pragma unroll
    for (int i = 0; i < VQI; ++i) {
      const int q = 64 * w + lane;
      const int vq_row = q / NCB, vq_c = q % NCB;
      const h16x2 m2 = __builtin_bit_cast(h16x2, vmeta_cur[i]);
      const h16x2 sc2 = h16x2{m2[0], m2[0]};
      const h16 nb = -((h16)1024.f + m2[1]);
      const h16x2 nb2 = h16x2{nb, nb};
      const unsigned rowb = lds0 + OFF_VS + (unsigned)(half * VH_BYTES + vq_row * RVB);
      if constexpr (QB == 3) {
        // 32 codes = 3 dwords (quant.hip: code j of a row at bits [3 j, 3 j + 3) of its little-endian stream); 8-code group j = 24 bits
        const unsigned src = lds0 + OFF_VC + (unsigned)(half * VC_BYTES + vq_row * VRB + vq_c * 12);
        const unsigned d0 = *(const __attribute__((address_space(3))) unsigned*)(uintptr_t)src;
        const unsigned d1 = *(const __attribute__((address_space(3))) unsigned*)(uintptr_t)(src + 4);
        const unsigned d2 = *(const __attribute__((address_space(3))) unsigned*)(uintptr_t)(src + 8);
        const unsigned v24[4] = {d0, __builtin_amdgcn_alignbit(d1, d0, 24), __builtin_amdgcn_alignbit(d2, d1, 16), d2 >> 8};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u32x4 o;
#pragma unroll
          for (int m = 0; m < 4; ++m) o[m] = deq3(v24[j], 6 * m, 6 * m + 3, nb2, sc2);
          const int gran = 2 * vq_c + (j >> 1);
          *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(rowb + (unsigned)(((gran ^ (vq_row & SWZ)) << 5) + 16 * (j & 1))) = o;
        }
        continue;
      }
      const u32x4 cd = *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)(lds0 + OFF_VC + half * VC_BYTES + q * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) {                 // dword j = columns 32 c + 8 j .. + 7 in natural order
        const unsigned d = cd[j];
        const unsigned t0 = d & 0x0F0F0F0Fu, t1 = (d >> 4) & 0x0F0F0F0Fu;  // codes 0 2 4 6 / 1 3 5 7 in bytes
        u32x4 o;
        o[0] = deq2(__builtin_amdgcn_perm(t1, t0, 0x0c040c00u), nb2, sc2); // (c0, c1): byte 0 of t0, byte 0 of t1
        o[1] = deq2(__builtin_amdgcn_perm(t1, t0, 0x0c050c01u), nb2, sc2);
        o[2] = deq2(__builtin_amdgcn_perm(t1, t0, 0x0c060c02u), nb2, sc2);
        o[3] = deq2(__builtin_amdgcn_perm(t1, t0, 0x0c070c03u), nb2, sc2);
        const int gran = 2 * vq_c + (j >> 1);
        *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(rowb + (unsigned)(((gran ^ (vq_row & SWZ)) << 5) + 16 * (j & 1))) = o;
      }
    }
  };
  // V half tile (fp16 caches): piece p covers LDS slots [64 p, 64 p + 64); slot s: row s / SPR, 16-byte slot s % SPR of the row; LDS
  // granule (32 B) gl of row r holds source granule gl ^ (r & SWZ).  The lane pattern of a piece repeats every PER pieces (= whole
  // groups of SWZ + 1 rows): a wave takes blocks of PER consecutive pieces (block 4 j + (w & 3)), so PER lane constants serve and the
  // block is a scalar row offset
  constexpr int SPR = RVB / 16;                     // 16-byte slots per row
  constexpr int PER = (NCB % 3 == 0) ? 3 : (NCB == 8 ? 2 : (NCB == 10 ? 5 : 1));
  static_assert((PER * 64) % ((SWZ + 1) * SPR) == 0 && NVP % PER == 0, "blocks of PER pieces are whole swizzle periods of rows");
  constexpr int RPB = PER * 64 / SPR;               // rows per block
  constexpr int NBLK = NVP / PER;                   // blocks of a half tile
  constexpr int NBW = (NBLK + 3) / 4;               // blocks per wave (the last round may not reach every wave)
  unsigned vvo[PER];
#pragma unroll
  for (int c = 0; c < PER; ++c) {
    const int s = c * 64 + lane;
    const int r = s / SPR, sr = s % SPR;
    vvo[c] = (unsigned)(r * (int)(p.sxv_l * 2) + ((((sr >> 1) ^ (r & SWZ)) << 1) + (sr & 1)) * 16);
  }
  const unsigned vrow_bytes = __builtin_amdgcn_readfirstlane((unsigned)(p.sxv_l * 2));
  // `all8`: the blocks are dealt to all eight waves (block 8 j + w) instead of the four waves of one role (block 4 j + (w & 3))
  auto dma_v = [&](int jt, int half, bool all8) {   // half tile (jt, half) into slot `half`
    const int row0 = jt * PL_BN + 32 * half;
    const int nw = all8 ? 8 : 4, wi = all8 ? w : (w & 3);
    if (row0 + 32 <= p.Tk) {
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int blk = nw * j + wi;
        if (blk >= NBLK) continue;
#pragma unroll
        for (int c = 0; c < PER; ++c)
          dma(__builtin_amdgcn_readfirstlane(lds0 + OFF_VS + half * VH_BYTES + (PER * blk + c) * 1024), vvo[c], vrs,
              __builtin_amdgcn_readfirstlane((unsigned)(row0 + RPB * blk) * vrow_bytes));
      }
    } else {
      // the cache's last rows: rows past Tk re-read row Tk - 1 (finite data; their probabilities are exactly zero)
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int blk = nw * j + wi;
        if (blk >= NBLK) continue;
#pragma unroll
        for (int c = 0; c < PER; ++c) {
          const int s = (PER * blk + c) * 64 + lane;
          const int r = s / SPR, sr = s % SPR;
          const int row = min(row0 + r, p.Tk - 1);
          dma(__builtin_amdgcn_readfirstlane(lds0 + OFF_VS + half * VH_BYTES + (PER * blk + c) * 1024),
              (unsigned)(row * (int)(p.sxv_l * 2) + ((((sr >> 1) ^ (r & SWZ)) << 1) + (sr & 1)) * 16), vrs, 0u);
        }
      }
    }
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // ================================================================================================ S-wave state
  // Q~ fragments (B operand of S^T): lane (t, hi) holds Q~[t][16 ks + 8 hi .. + 7]; parked in LDS (this wave's own 8 KB)
  const unsigned qfa = lds0 + OFF_QF + (unsigned)((qblk * 8 * 64 + lane) * 16);
  // B_h^T fragments of this wave's 64 rows (A operand of the rebuild), two M-blocks mbk: MFMA row m = 8 a + 4 hb + b  <->
  // d = 32 dh + 16 mbk + 8 (a & 1) + 4 hb + b + 64 (a >> 1): after the MFMA lane (kv, hi) holds register 4 a + b = K[kv][d(a, hi, b)] --
  // both halves of 8 RoPE pairs per M-block
  h16x8 af[2][KSR];
  h16x4 cs_c[2][2], cs_s[2][2];                       // rotary values of this lane's kv row of the tile rebuilt next: [mbk][a1] x 4 pairs
  float m_run = -INFINITY, l_run = 0.f;
  const int kvh = w & 1, dh = (w >> 1) & 1;
  if (swave) {
    const h16* qp = p.q + (int64_t)h * p.sq_h + (int64_t)(qvalid ? qrow : 0) * p.sq_t + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + 16 * ks);
      if (!qvalid) v = u32x4{0, 0, 0, 0};
      *(lds_h16x8_t*)(uintptr_t)(qfa + (unsigned)(ks * 1024)) = __builtin_bit_cast(h16x8, v);
    }
    const int a = n >> 3, hb = (n >> 2) & 1, b = n & 3;
#pragma unroll
    for (int mbk = 0; mbk < 2; ++mbk) {
      const int d = 32 * dh + 16 * mbk + 8 * (a & 1) + 4 * hb + b + 64 * (a >> 1);
      const h16* bp = p.bt + ((int64_t)h * 128 + d) * RK + 8 * hi;
#pragma unroll
      for (int ks = 0; ks < KSR; ++ks) af[mbk][ks] = __builtin_bit_cast(h16x8, *reinterpret_cast<const u32x4*>(bp + 16 * ks));
    }
  }
  // K~ A-fragment row of this lane for the scores: bits 2 and 3 of the MFMA row swapped, so that the 8 registers of a k-step
  // hold 8 CONSECUTIVE kv positions (prefill_attn.hip)
  const int krow = (n & 0x13) | ((n & 4) << 1) | ((n & 8) >> 1);

  // rotary values for the rebuild of tile jt: requested a phase ahead (plain loads; the phase's vmcnt(0) covers them)
  unsigned kmeta = 0;                                 // packed keys: (scale, zero) of this lane's kv row of the tile rebuilt next
  auto load_cs = [&](int jt) {
    const int pos = min(jt * PL_BN + 32 * kvh + n, p.Tk - 1);
    if (QB) kmeta = *reinterpret_cast<const unsigned*>(p.km + (int64_t)g * p.skm_g + (int64_t)pos * p.skm_l);
    const h16* cr = p.cs + (int64_t)pos * 128 + 32 * dh + 4 * hi;
#pragma unroll
    for (int mbk = 0; mbk < 2; ++mbk)
#pragma unroll
      for (int a1 = 0; a1 < 2; ++a1) {
        cs_c[mbk][a1] = *reinterpret_cast<const h16x4*>(cr + 16 * mbk + 8 * a1);
        cs_s[mbk][a1] = *reinterpret_cast<const h16x4*>(cr + 64 + 16 * mbk + 8 * a1);
      }
  };
  // rebuild of this wave's quarter of K~ tile jt from the X tile in LDS: 8 B-fragment reads, 16 MFMAs, fp16 rounding (the
  // reference's reconstruct GEMM rounds K to fp16, :67-77), rotation with fp16 cos / sin and fp16 products / sum (HF
  // apply_rotary_pos_emb on fp16 tensors, :204-205; rope.hip's arithmetic), 8 x 8 bytes per lane into tile image jt & 1
  auto build = [&](int jt) {
    const int row = 32 * kvh + n;                                       // this lane's kv row of the tile (B-operand column)
    const unsigned xbase = lds0 + OFF_XS + (unsigned)(row * RKB + ((hi ^ (row & (CPRX - 1))) << 4));
    const unsigned kbase = lds0 + OFF_KS + (unsigned)((jt & 1) * KS_BYTES + row * 256 + ((row & 15) << 4) + 8 * hi);
    f32x16 kacc[2];
#pragma unroll
    for (int mbk = 0; mbk < 2; ++mbk)
#pragma unroll
      for (int e = 0; e < 16; ++e) kacc[mbk][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks) {
      h16x8 xf;
      if constexpr (QB == 0) {
        xf = *(const lds_h16x8_t*)(uintptr_t)(xbase ^ (unsigned)(ks << 5));               // chunk (2 ks + hi) ^ (row & (CPRX - 1))
      } else if constexpr (QB == 3) {
        // 8 codes = 24 bits at byte 6 ks + 3 hi of the row's 48: two aligned dwords and a funnel shift -- even ks: byte 6 ks, shift 24 hi;
        // odd ks: byte 6 ks - 2 + 4 hi, shift 16 - 8 hi.  Pairs (e, e + 4) like the 4-bit form (the same permuted B^T).
        const unsigned xa = lds0 + OFF_XS + (unsigned)(row * RQB + ((ks & 1) ? 6 * ks - 2 + 4 * hi : 6 * ks));
        const unsigned d0 = *(const __attribute__((address_space(3))) unsigned*)(uintptr_t)xa;
        const unsigned d1 = *(const __attribute__((address_space(3))) unsigned*)(uintptr_t)(xa + 4);
        const unsigned v24 = __builtin_amdgcn_alignbit(d1, d0, (unsigned)((ks & 1) ? 16 - 8 * hi : 24 * hi));
        const h16x2 m2 = __builtin_bit_cast(h16x2, kmeta);
        const h16x2 sc2 = h16x2{m2[0], m2[0]};
        const h16 nb = -((h16)1024.f + m2[1]);
        const h16x2 nb2 = h16x2{nb, nb};
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = deq3(v24, 3 * e, 3 * e + 12, nb2, sc2);
        xf = __builtin_bit_cast(h16x8, o);
      } else {
        // 8 codes = 4 bytes at byte 8 ks + 4 hi of the row's 64: chunk ks >> 1 (at position ^ ((row >> 1) & 3)), nibble e of the dword =
        // column 16 ks + 8 hi + e; the pairs come out as (0, 4) (1, 5) (2, 6) (3, 7): bt carries its columns in that order
        const unsigned d = *(const __attribute__((address_space(3))) unsigned*)(uintptr_t)(
            lds0 + OFF_XS + (unsigned)(row * RQB + ((((ks >> 1) ^ ((row >> 1) & (CQ - 1)))) << 4) + 8 * (ks & 1) + 4 * hi));
        const h16x2 m2 = __builtin_bit_cast(h16x2, kmeta);
        const h16x2 sc2 = h16x2{m2[0], m2[0]};
        const h16 nb = -((h16)1024.f + m2[1]);
        const h16x2 nb2 = h16x2{nb, nb};
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = deq2((d >> (4 * e)) & 0x000F000Fu, nb2, sc2);
        xf = __builtin_bit_cast(h16x8, o);
      }
      kacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][ks], xf, kacc[0], 0, 0, 0);
      kacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][ks], xf, kacc[1], 0, 0, 0);
    }
#pragma unroll
    for (int mbk = 0; mbk < 2; ++mbk) {
      h16x4 k4[4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) k4[a][b] = (h16)kacc[mbk][4 * a + b];
#pragma unroll
      for (int a1 = 0; a1 < 2; ++a1) {
#pragma clang fp contract(off)   // two fp16 products and one fp16 sum, never an fma (rope.hip)
        const h16x4 lo = k4[a1], hh = k4[a1 + 2];
        const h16x4 olo = lo * cs_c[mbk][a1] + (-hh) * cs_s[mbk][a1];
        const h16x4 ohi = hh * cs_c[mbk][a1] + lo * cs_s[mbk][a1];
        // d = 32 dh + 16 mbk + 8 a1 + 4 hi + b (+ 64): 16-byte chunk 4 dh + 2 mbk + a1 (+ 8), bytes 8 hi .. 8 hi + 7
        *(lds_h16x4_t*)(uintptr_t)(kbase ^ (unsigned)((4 * dh + 2 * mbk + a1) << 4)) = olo;
        *(lds_h16x4_t*)(uintptr_t)(kbase ^ (unsigned)((4 * dh + 2 * mbk + a1 + 8) << 4)) = ohi;
      }
    }
  };

  // scores + online softmax of tile jt in two parts (the workgroup's mid-tile barrier lies between them): part 1 = the score MFMAs,
  // mask, running maximum, rescale factor and the probabilities of the tile's first 32 positions; part 2 = the other 32 and the
  // running sum.  Probabilities (fp16, B-operand order) and the rescale factor go to LDS buffer jt & 1.
  f32x16 sacc[2];
  float sm_moff = 0.f, sm_alpha = 1.f, sm_lsum = 0.f;
  auto softmax_rows = [&](int jt, auto rb_c) {
    constexpr int rb = decltype(rb_c)::value;
    const unsigned pdst = lds0 + OFF_PS + (unsigned)((qblk * 4) * 1024 + lane * 16);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      h16x8 pk;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#if PL_EXP & 2
        const float pv = fmaf(sacc[rb][8 * s2 + e], p.scale_log2, sm_moff);
#else
        const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[rb][8 * s2 + e], p.scale_log2, sm_moff));
#endif
        sm_lsum += pv;
        pk[e] = (h16)pv;
      }
      *(lds_h16x8_t*)(uintptr_t)(pdst + (unsigned)((2 * rb + s2) * 1024)) = pk;
    }
  };
  auto scores_part1 = [&](int jt) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[rb][e] = 0.f;
    // (k-step outer, the two row blocks inner: two independent accumulator chains, one Q~ fragment read for both)
    const unsigned kt = lds0 + OFF_KS + (unsigned)((jt & 1) * KS_BYTES + krow * 256 + ((hi ^ (krow & 15)) << 4));
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const h16x8 qk = *(const lds_h16x8_t*)(uintptr_t)(qfa + (unsigned)(kk * 1024));
      const h16x8 kf0 = *(const lds_h16x8_t*)(uintptr_t)(kt ^ (unsigned)(kk << 5));                    // row krow: chunk (2 kk + hi) ^ (row & 15)
      const h16x8 kf1 = *(const lds_h16x8_t*)(uintptr_t)((kt ^ (unsigned)(kk << 5)) + 32 * 256);     // row 32 + krow (same row & 15)
      sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qk, sacc[0], 0, 0, 0);
      sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qk, sacc[1], 0, 0, 0);
    }
    // register r of block rb in lane (t, hi) is kv = jt*64 + 32 rb + 16 (r >> 3) + 8 hi + (r & 7)
    const int j0 = jt * PL_BN + 8 * hi;
    const bool need_mask = (jt * PL_BN + PL_BN > p.Tk) || (p.causal && jt * PL_BN + PL_BN - 1 > p.past + qt * PL_BM + qblk * 32);
    if (need_mask) {
      const int lim = p.causal ? min(p.Tk - 1, qpos) : p.Tk - 1;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + 32 * rb + 16 * (r >> 3) + (r & 7);
          if (j > lim) sacc[rb][r] = -INFINITY;
        }
    }
    float mloc = sacc[0][0];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[rb][r]);
    {
      const unsigned mb = __float_as_uint(mloc);
      auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);   // both halves of query t see both maxima
      mloc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const float m_new = fmaxf(m_run, mloc);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;                // fully masked so far (padding rows)
    sm_alpha = __builtin_amdgcn_exp2f((m_run - m_use) * p.scale_log2);     // m_run = -inf -> 0
    sm_moff = -m_use * p.scale_log2;
    sm_lsum = 0.f;
    m_run = m_new;
    if (hi == 0) *(lds_f32_t*)(uintptr_t)(lds0 + OFF_AL + (unsigned)((qblk * 32 + n) * 4)) = sm_alpha;
    softmax_rows(jt, std::integral_constant<int, 0>{});
  };
  auto scores_part2 = [&](int jt) {
    softmax_rows(jt, std::integral_constant<int, 1>{});
    l_run = fmaf(l_run, sm_alpha, sm_lsum);
  };

  // ================================================================================================ O-wave state
  f32x16 acc_o[NCB];                                  // (zeroed at the top of the O-wave's branch)
  // transpose-read address of this lane inside a half tile (k-step 0, first 4 rows): 16-lane group j = lane >> 4 covers columns
  // 16 (j & 1) .. + 15 of a 32-column block for the k-half kg = j >> 1; lane i of the group reads 8 bytes of row 8 kg + (i >> 2)
  // at columns 4 (i & 3) .. + 3 and receives column i's four rows.  Granule of (cb, j & 1): 2 cb + (j & 1), XOR (row & 3) = i >> 2
  // on its low two bits (even / odd cb differ in bit 1: two lane constants), the rest rides in the instruction offsets.
  const int tj = lane >> 4, ti = lane & 15;
  const unsigned tbase = lds0 + OFF_VS + (unsigned)((8 * (tj >> 1) + (ti >> 2)) * RVB + 8 * (ti & 3));
  // (odd NCB, key = row & 1: the key moves bit 0 only, every cb is a constant offset on tr_e)
  const unsigned tr_e = tbase + (unsigned)((((tj & 1)) ^ ((ti >> 2) & SWZ)) * 32);
  const unsigned tr_o = tbase + (unsigned)((((2 + (tj & 1))) ^ ((ti >> 2) & SWZ)) * 32);
  auto pv_half = [&](auto half_c) {         // O^T += V^T(jt, half) . P^T(jt, half): k-steps 2 half, 2 half + 1
    constexpr int half = decltype(half_c)::value;
    const unsigned psrc = lds0 + OFF_PS + (unsigned)((qblk * 4) * 1024 + lane * 16);
    if (half == 0) {
      const float al = *(const lds_f32_t*)(uintptr_t)(lds0 + OFF_AL + (unsigned)((qblk * 32 + n) * 4));
      if (__builtin_amdgcn_ballot_w64(al != 1.0f) != 0) {                // wave-uniform: rescale only when a maximum moved
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc_o[cb][e] *= al;
      }
    }
    // software pipeline over the half's 2 NCB (k-step, column block) steps in groups of GS: the transpose reads of group g + 1 are in
    // flight while the MFMAs of group g run (timeline of the first version: a P.V half took 1.9 - 3.2 k ticks for 768 cycles of matrix
    // pipe -- read, wait, three MFMAs, read ...; the O-wave is alone on its SIMD's LDS latency)
    constexpr int GS = (NCB % 3 == 0) ? 3 : 2;
    constexpr int NG = 2 * NCB / GS;
    const h16x8 pf0 = *(const lds_h16x8_t*)(uintptr_t)(psrc + (unsigned)((2 * half) * 1024));
    const h16x8 pf1 = *(const lds_h16x8_t*)(uintptr_t)(psrc + (unsigned)((2 * half + 1) * 1024));
    typedef __attribute__((address_space(3))) char lds_char;
    auto vload = [&](h16x8 (&vf)[GS], auto g_c) {
      constexpr int gq = decltype(g_c)::value;
#pragma unroll
      for (int e = 0; e < GS; ++e) {
        constexpr int dummy = 0;
        (void)dummy;
        const int t = gq * GS + e, st = t / NCB, cb = t % NCB;
        // (constant offsets on an LDS pointer: they ride in the instructions' offset fields instead of address registers)
        lds_char* ap = SWZ == 3 ? (lds_char*)(uintptr_t)((cb & 1) ? tr_o : tr_e) + (half * VH_BYTES + st * 16 * RVB + (cb >> 1) * 128)
                                : (lds_char*)(uintptr_t)tr_e + (half * VH_BYTES + st * 16 * RVB + cb * 64);
        const h16x4 lo = __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((pvm::lds_s16x4*)ap));
        const h16x4 hh = __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((pvm::lds_s16x4*)(ap + 4 * RVB)));
        vf[e] = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    h16x8 vfA[GS], vfB[GS];
    vload(vfA, std::integral_constant<int, 0>{});
    auto pv_group = [&](auto g_c) {
      constexpr int gq = decltype(g_c)::value;
      if constexpr (gq + 1 < NG) vload((gq & 1) ? vfA : vfB, std::integral_constant<int, gq + 1>{});
#pragma unroll
      for (int e = 0; e < GS; ++e) {
        const int t = gq * GS + e, st = t / NCB, cb = t % NCB;
#if !(PL_EXP & 4)
        acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(((gq & 1) ? vfB : vfA)[e], st ? pf1 : pf0, acc_o[cb], 0, 0, 0);
#else
        acc_o[cb][0] += (float)((gq & 1) ? vfB : vfA)[e][0] + (float)pf0[0];
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    pl_for<0, NG>(pv_group);
  };

  // ================================================================================================ pipeline
  // (the two roles run their own copy of the loop -- same barriers, same DMA schedule -- so that the S-wave's operand fragments
  //  and the O-wave's 192 accumulators never share a live range: the kernel needs max(S, O) registers, not their sum)
  auto pin_prefetched = [&]() {
    // (hipcc's own waits for the plain loads of a phase go HERE, behind the phase's vmcnt(0), where nothing is in flight: in front of
    //  their first use they would be a vmcnt(0) behind the next phase's DMA requests)
    if (swave) {
#pragma unroll
      for (int mbk = 0; mbk < 2; ++mbk)
        asm volatile("" : "+v"(cs_c[mbk][0]), "+v"(cs_c[mbk][1]), "+v"(cs_s[mbk][0]), "+v"(cs_s[mbk][1]));
      if (QB) asm volatile("" : "+v"(kmeta));
    }
    if (QB) {
#pragma unroll
      for (int i = 0; i < VQI; ++i) {
        asm volatile("" : "+v"(vmeta_next[i]));
        vmeta_cur[i] = vmeta_next[i];
      }
    }
  };
  if (njt > 0) {
    if (swave) { if (QB) dma_xq(0); else dma_x(0); }
    vc_issue(0, 0);
    if (swave) load_cs(0);
    dma_wait();
    pin_prefetched();
  }
  __syncthreads();
  if (swave && njt > 0) build(0);
  __syncthreads();                                    // K~(0) visible, the X buffer free
  // The O-wave runs half a tile behind the S-wave: alpha(jt) = second half of P.V of tile jt - 1, beta(jt) = first half of tile jt
  // (whose probabilities the S-wave wrote in alpha(jt)) -- so one probability buffer serves, each k-step pair written in one phase
  // and read in the next.
  // who requests what (timelines, profiles/r06_prefill_lat_timeline.txt): a CU takes in ~28 B per clock and a wave sits 100 - 250 ticks
  // in every 1 KB piece it requests -- 64 pieces per tile.  Three assignments were measured within 2 % of each other (all staging by the
  // S-waves 91.7 ms at 64k tokens; phase alpha by the O-waves, beta by the S-waves 92.6; X by S, V(.., 0) by O, V(.., 1) by all eight
  // 93.5): the S-waves request everything -- the O-waves' P.V (incl. the accumulator rescale) is the longer half of both phases.
  auto stage_alpha = [&](int jt) {
    if (QB == 0) {
      if (swave) {
        if (jt + 1 < njt) dma_x(jt + 1);              // consumed by the rebuild in phase beta
        if (jt < njt) dma_v(jt, 0, false);            // consumed in phase beta (slot 0 was read in the last phase beta)
      }
    } else {
      if (!swave && jt + 1 < njt) dma_xq(jt + 1);
      if (jt < njt) {
        vc_dequant(0);                                // codes of (jt, 0): staged in the last phase beta (or the prologue)
        vc_issue(jt, 1);
      }
    }
  };
  auto stage_beta = [&](int jt) {
    if (QB == 0) {
      if (swave && jt < njt) dma_v(jt, 1, false);     // consumed in the next phase alpha (slot 1 was read in this one)
    } else {
      if (jt < njt) vc_dequant(1);
      if (jt + 1 < njt) vc_issue(jt + 1, 0);
    }
  };
  if (swave) {
    for (int jt = 0; jt <= njt; ++jt) {               // iteration njt only drains the last tile's P.V
      stamp(jt);                                      // 0: phase alpha starts
      stage_alpha(jt);                                // ---- phase alpha: scores of tile jt, first half of its softmax
      if (jt + 1 < njt) load_cs(jt + 1);
      stamp(jt);                                      // 1: staging requested
      if (jt < njt) scores_part1(jt);
      stamp(jt);                                      // 2: work done
      dma_wait();
      pin_prefetched();
      stamp(jt);                                      // 3: staging landed
      __syncthreads();
      stamp(jt);                                      // 4: phase beta starts
      stage_beta(jt);                                 // ---- phase beta: second half, then the rebuild of K~ tile jt + 1 (other image)
      if (jt < njt) scores_part2(jt);
      stamp(jt);                                      // 5: softmax done
#if !(PL_EXP & 1)
      if (jt + 1 < njt) build(jt + 1);
#endif
      stamp(jt);                                      // 6: rebuild done
      dma_wait();
      pin_prefetched();
      stamp(jt);                                      // 7: staging landed
      __syncthreads();
    }
    // the row sums (both hi halves) for the O-wave
    const unsigned lb = __float_as_uint(l_run);
    auto sw = __builtin_amdgcn_permlane32_swap(lb, lb, false, false);
    const float l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    if (hi == 0) *(lds_f32_t*)(uintptr_t)(lds0 + OFF_AL + (unsigned)((qblk * 32 + n) * 4)) = l_tot;
    __syncthreads();
  } else {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc_o[cb][e] = 0.f;
    for (int jt = 0; jt <= njt; ++jt) {
      stamp(jt);                                      // 0
      stage_alpha(jt);                                // ---- phase alpha: second half of P.V of tile jt - 1
      stamp(jt);                                      // 1
      if (jt >= 1) pv_half(std::integral_constant<int, 1>{});
      stamp(jt);                                      // 2
      dma_wait();
      pin_prefetched();
      stamp(jt);                                      // 3
      __syncthreads();
      stamp(jt);                                      // 4
      stage_beta(jt);                                 // ---- phase beta: rescale, first half of tile jt
      if (jt < njt) pv_half(std::integral_constant<int, 0>{});
      stamp(jt);                                      // 5
      stamp(jt);                                      // 6
      dma_wait();
      pin_prefetched();
      stamp(jt);                                      // 7
      __syncthreads();
    }
    __syncthreads();
    // ---- epilogue: normalise and store fp16; lane (t, hi) register r of block cb is column 32 cb + (r & 3) + 8 (r >> 2) + 4 hi
    const float l_tot = *(const lds_f32_t*)(uintptr_t)(lds0 + OFF_AL + (unsigned)((qblk * 32 + n) * 4));
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qvalid) {
      h16* op = p.out + (int64_t)qrow * p.so_t + (int64_t)h * RV + 4 * hi;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) {
          h16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (h16)(acc_o[cb][4 * r2 + e] * inv);
          *reinterpret_cast<h16x4*>(op + 32 * cb + 8 * r2) = o;
        }
    }
  }
#ifdef PL_TIMELINE
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
#endif
}

// cos / sin of the key positions as the reference's rotary cache holds them (kernel/palu_attention.py:204: rotary_emb; HF
// LlamaRotaryEmbedding: angle = fl32(pos) * inv_freq in fp32, cos / sin in fp32, cast to the activation dtype): [pos][2][64] fp16
__global__ void rope_cs_table_kernel(const float* __restrict__ inv_freq, int pos0, int npos, h16* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npos * 64) return;
  const int t = idx >> 6, i = idx & 63;
  const float ang = (float)(pos0 + t) * inv_freq[i];
  float sn, cs;
  sincosf(ang, &sn, &cs);
  out[(int64_t)t * 128 + i] = (h16)cs;
  out[(int64_t)t * 128 + 64 + i] = (h16)sn;
}

template <int NCB, int KSR, int QB>
int launch_prefill_lat(const PfLatParams& p, hipStream_t stream) {
  constexpr int smem = 3 * PL_BN * 256 + 2 * 32 * 64 * NCB + 4 * 4096 + 4 * 32 * 4 + 32 * 1024 + (QB ? 2 * 1024 * ((32 * 32 * NCB * QB / 8 + 1023) / 1024) : 0);
  auto kern = prefill_lat_kernel<NCB, KSR, QB>;
  const int rca = palu_func_max_lds(reinterpret_cast<const void*>(kern), smem);
  if (rca) return rca;
  dim3 grid(p.head_major == 2 ? p.H * p.nqt : (p.head_major ? p.H : p.nqt), p.head_major == 2 ? 1 : (p.head_major ? p.nqt : p.H), 1);
  hipLaunchKernelGGL(kern, grid, dim3(PL_THREADS), smem, stream, p);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

// workgroup order: 2 = eight heads at a time, one per XCD (a head's query tiles share its group's rows in that XCD's L2), heavy
// tiles first; 1 = head-major.  (The query chunks of the module's prompt pass -- 24 tiles x 32 heads -- ran 5 % slower in order 1:
// 99.0 vs 94.0 ms for 64k tokens, tools/time_prefill_lat_chunks.py.)  PALU_PL_ORDER=0/1/2 forces one (experiments builds).
int pl_head_major(int nqt, int H) {
  static int force = -2;
  if (force == -2) {
    const char* e = palu_exp_env("PALU_PL_ORDER");
    force = e ? atoi(e) : -1;
  }
  int hm = force >= 0 ? force : ((int64_t)nqt * H <= 256 ? 1 : 2);
  if (hm == 2 && H % 8 != 0) hm = 0;
  return hm;
}

template <int QB>
int dispatch_prefill_lat(const PfLatParams& p, int Rk, int Rv, hipStream_t s) {
  if constexpr (QB == 3) {                                        // 3-bit rows: rank_k / G = 128, rank_v / G a multiple of 128
    if (Rv == 128) return launch_prefill_lat<4, 8, 3>(p, s);
    if (Rv == 256) return launch_prefill_lat<8, 8, 3>(p, s);
    return launch_prefill_lat<12, 8, 3>(p, s);
  } else {
#define PL_CASE(NCB)                                                             \
  case NCB:                                                                      \
    return Rk == 128 ? launch_prefill_lat<NCB, 8, QB>(p, s) : launch_prefill_lat<NCB, 4, QB>(p, s);
  if (Rk == 32) {                                                // BASELINE config 1 and the small golden-fixture shape
    if constexpr (QB == 0) {
      if (Rv == 96) return launch_prefill_lat<3, 2, QB>(p, s);
    }
    return launch_prefill_lat<2, 2, QB>(p, s);
  }
  switch (Rv / 32) {
    PL_CASE(4)
    PL_CASE(6)
    PL_CASE(8)
    default: return Rk == 128 ? launch_prefill_lat<12, 8, QB>(p, s) : launch_prefill_lat<12, 4, QB>(p, s);
  }
#undef PL_CASE
  }
}

}  // namespace

// debug (-DPL_TIMELINE builds): device buffer [8 waves][64] of s_memtime stamps workgroup 0 of the next launches fills; 0 = off
extern "C" void palu_prefill_lat_timeline_buffer(void* ptr) { g_pl_timeline = (unsigned long long*)ptr; }

extern "C" size_t palu_rope_cs_table_bytes(int npos) { return npos > 0 ? (size_t)npos * 128 * sizeof(h16) : 0; }

extern "C" int palu_rope_cs_table_build(const float* inv_freq, int pos0, int npos, void* table, palu_stream_t stream) {
  PALU_REQUIRE(inv_freq && table && pos0 >= 0 && npos > 0 && (int64_t)pos0 + npos < (1 << 24), PALU_ERR_ARG,
               "rope_cs_table_build: bad arguments");
  PALU_REQUIRE(((uintptr_t)table & 15) == 0, PALU_ERR_ARG, "rope_cs_table_build: table must be 16-byte aligned");
  hipLaunchKernelGGL(rope_cs_table_kernel, dim3((unsigned)((npos * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, inv_freq,
                     pos0, npos, (h16*)table);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

extern "C" int palu_prefill_attn_lat_supported(int H, int G, int D, int Rk, int Rv) {
  if (!(H > 0 && G > 0 && H % G == 0 && D == 128)) return 0;
  if (Rk == 32 && (Rv == 64 || Rv == 96)) return 1;  // BASELINE config 1 (32 / 96, fp16 rows only) and what the reference-generated
                                                     // prefill fixtures use (tests/golden/g7_prefill.npz)
  return ((Rk == 128 || Rk == 64) && (Rv == 384 || Rv == 256 || Rv == 192 || Rv == 128)) ? 1 : 0;
}

// bits = 16 (fp16 rows), 4 or 3 (packed rows with whole-row (scale, zero) pairs): which entry takes the shape
extern "C" int palu_prefill_attn_lat_supported_bits(int H, int G, int D, int Rk, int Rv, int bits) {
  if (!palu_prefill_attn_lat_supported(H, G, D, Rk, Rv)) return 0;
  if (bits == 16) return 1;
  if (bits == 4) return Rv % 64 == 0 ? 1 : 0;
  if (bits == 3) return (Rk == 128 && Rv % 128 == 0) ? 1 : 0;
  return 0;
}

// Prompt attention of Tq queries (rotated, [H][Tq][128]; the first at absolute position `past`) over the first Tk rows of the
// latent caches xk [G][.][128], xv [G][.][Rv] (fp16, row l = position l, 16-byte aligned rows); bt = B^T [H][128][Rk] contiguous;
// cs = palu_rope_cs_table_build(inv_freq, 0, >= Tk).  out [Tq][H * Rv] fp16.  No workspace.
extern "C" int palu_prefill_attn_lat_f16(const void* q, int64_t sq_h, int64_t sq_t, const void* xk, int64_t sxk_g, int64_t sxk_l,
                                         const void* xv, int64_t sxv_g, int64_t sxv_l, const void* bt, const void* cs, void* out,
                                         int64_t so_t, int H, int G, int D, int Tq, int Tk, int Rk, int Rv, int past, int causal,
                                         float scale, palu_stream_t stream) {
  PALU_REQUIRE(q && xk && xv && bt && cs && out, PALU_ERR_ARG, "prefill_attn_lat: null pointer");
  PALU_REQUIRE(palu_prefill_attn_lat_supported(H, G, D, Rk, Rv), PALU_ERR_UNSUPPORTED,
               "prefill_attn_lat: needs head_dim 128, rank_k / G in {64, 128} with rank_v / G in {128, 192, 256, 384}, or 32 / 64 (H=%d G=%d D=%d Rk=%d Rv=%d)", H, G, D,
               Rk, Rv);
  PALU_REQUIRE(Tq >= 0 && Tk >= 0 && past >= 0 && (int64_t)past + Tq < (1 << 30), PALU_ERR_ARG, "prefill_attn_lat: bad lengths");
  if (Tq == 0) return PALU_OK;
  PALU_REQUIRE(Tk > 0, PALU_ERR_ARG, "prefill_attn_lat: no keys");
  PALU_REQUIRE(scale > 0.f, PALU_ERR_ARG, "prefill_attn_lat: scale must be positive");
  PALU_REQUIRE((((uintptr_t)q | (uintptr_t)xk | (uintptr_t)xv | (uintptr_t)bt | (uintptr_t)cs) & 15) == 0 && sq_h % 8 == 0 &&
                   sq_t % 8 == 0 && sxk_g % 8 == 0 && sxk_l % 8 == 0 && sxv_g % 8 == 0 && sxv_l % 8 == 0 && sxk_l >= Rk &&
                   sxv_l >= Rv && ((uintptr_t)out & 7) == 0 && so_t % 4 == 0,
               PALU_ERR_ARG, "prefill_attn_lat: rows must be 16-byte aligned (out 8-byte)");
  PALU_REQUIRE(((int64_t)Tk + PL_BN) * sxk_l * 2 < ((int64_t)1 << 32) && ((int64_t)Tk + PL_BN) * sxv_l * 2 < ((int64_t)1 << 32),
               PALU_ERR_UNSUPPORTED, "prefill_attn_lat: one group's latent slab must stay below 4 GiB");
  PfLatParams p;
  p.q = (const h16*)q; p.sq_h = sq_h; p.sq_t = sq_t;
  p.xk = (const h16*)xk; p.sxk_g = sxk_g; p.sxk_l = sxk_l;
  p.xv = (const h16*)xv; p.sxv_g = sxv_g; p.sxv_l = sxv_l;
  p.bt = (const h16*)bt; p.cs = (const h16*)cs;
  p.out = (h16*)out; p.so_t = so_t;
  p.H = H; p.G = G; p.gs = H / G; p.Tq = Tq; p.Tk = Tk; p.past = past; p.causal = causal ? 1 : 0;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.nqt = (Tq + PL_BM - 1) / PL_BM;
  p.head_major = pl_head_major(p.nqt, H);
  p.dbg = g_pl_timeline;
  p.kc = p.vc = nullptr; p.km = p.vm = nullptr;
  p.skc_g = p.skc_l = p.skm_g = p.skm_l = p.svc_g = p.svc_l = p.svm_g = p.svm_l = 0;
  return dispatch_prefill_lat<0>(p, Rk, Rv, (hipStream_t)stream);
}

// The same over PACKED 4-bit or 3-bit caches (quant.hip's layout: codes [G][.][R bits / 8] bytes, meta [G][.][2] fp16 (scale, zero) per
// (token, group) row; byte strides for the codes, element strides for the meta; 3-bit: rank_k / G = 128 and rank_v / G in {128, 256, 384}): the codes are de-quantised inside the kernel -- keys in the
// rebuild's registers, values into the half-tile image -- with unpack_dequant's arithmetic; no fp16 copy of the cache exists.
// bt = B^T [H][128][Rk] with the columns of every group of 8 in the order 0 4 1 5 2 6 3 7.
extern "C" int palu_prefill_attn_lat_q(const void* q, int64_t sq_h, int64_t sq_t, const void* k_codes, int64_t skc_g, int64_t skc_l,
                                       const void* k_meta, int64_t skm_g, int64_t skm_l, const void* v_codes, int64_t svc_g,
                                       int64_t svc_l, const void* v_meta, int64_t svm_g, int64_t svm_l, const void* bt_perm,
                                       const void* cs, void* out, int64_t so_t, int H, int G, int D, int Tq, int Tk, int Rk, int Rv,
                                       int bits, int past, int causal, float scale, palu_stream_t stream) {
  PALU_REQUIRE(q && k_codes && k_meta && v_codes && v_meta && bt_perm && cs && out, PALU_ERR_ARG, "prefill_attn_lat_q: null pointer");
  PALU_REQUIRE(bits == 4 || bits == 3, PALU_ERR_UNSUPPORTED, "prefill_attn_lat_q: 3- or 4-bit codes (got %d)", bits);
  PALU_REQUIRE(palu_prefill_attn_lat_supported_bits(H, G, D, Rk, Rv, bits), PALU_ERR_UNSUPPORTED,
               "prefill_attn_lat_q: needs head_dim 128 and, at 4 bit, rank_k / G in {64, 128} with rank_v / G in {128, 192, 256, 384} or 32 / 64; at 3 "
               "bit rank_k / G = 128 with rank_v / G in {128, 256, 384} (H=%d G=%d D=%d Rk=%d Rv=%d bits=%d)", H, G, D, Rk, Rv, bits);
  PALU_REQUIRE(Tq >= 0 && Tk >= 0 && past >= 0 && (int64_t)past + Tq < (1 << 30), PALU_ERR_ARG, "prefill_attn_lat_q: bad lengths");
  if (Tq == 0) return PALU_OK;
  PALU_REQUIRE(Tk > 0, PALU_ERR_ARG, "prefill_attn_lat_q: no keys");
  PALU_REQUIRE(scale > 0.f, PALU_ERR_ARG, "prefill_attn_lat_q: scale must be positive");
  PALU_REQUIRE((((uintptr_t)q | (uintptr_t)k_codes | (uintptr_t)v_codes | (uintptr_t)bt_perm | (uintptr_t)cs) & 15) == 0 &&
                   (((uintptr_t)k_meta | (uintptr_t)v_meta) & 3) == 0 && sq_h % 8 == 0 && sq_t % 8 == 0 && skc_g % 16 == 0 &&
                   skc_l % 16 == 0 && svc_g % 16 == 0 && svc_l % 16 == 0 && skm_g % 2 == 0 && skm_l % 2 == 0 && svm_g % 2 == 0 &&
                   svm_l % 2 == 0 && skc_l >= Rk * bits / 8 && svc_l >= Rv * bits / 8 && ((uintptr_t)out & 7) == 0 && so_t % 4 == 0,
               PALU_ERR_ARG, "prefill_attn_lat_q: code rows must be 16-byte aligned, meta pairs 4-byte aligned (out 8-byte)");
  PALU_REQUIRE(((int64_t)Tk + PL_BN) * skc_l < ((int64_t)1 << 32) && ((int64_t)Tk + PL_BN) * svc_l < ((int64_t)1 << 32),
               PALU_ERR_UNSUPPORTED, "prefill_attn_lat_q: one group's code slab must stay below 4 GiB");
  PfLatParams p;
  p.q = (const h16*)q; p.sq_h = sq_h; p.sq_t = sq_t;
  p.xk = (const h16*)k_codes; p.sxk_g = 0; p.sxk_l = Rk;       // (fp16 staging geometry unused)
  p.xv = (const h16*)v_codes; p.sxv_g = 0; p.sxv_l = Rv;
  p.kc = (const unsigned char*)k_codes; p.skc_g = skc_g; p.skc_l = skc_l;
  p.km = (const h16*)k_meta; p.skm_g = skm_g; p.skm_l = skm_l;
  p.vc = (const unsigned char*)v_codes; p.svc_g = svc_g; p.svc_l = svc_l;
  p.vm = (const h16*)v_meta; p.svm_g = svm_g; p.svm_l = svm_l;
  p.dbg = g_pl_timeline;
  p.bt = (const h16*)bt_perm; p.cs = (const h16*)cs;
  p.out = (h16*)out; p.so_t = so_t;
  p.H = H; p.G = G; p.gs = H / G; p.Tq = Tq; p.Tk = Tk; p.past = past; p.causal = causal ? 1 : 0;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.nqt = (Tq + PL_BM - 1) / PL_BM;
  p.head_major = pl_head_major(p.nqt, H);
  if (bits == 3) return dispatch_prefill_lat<3>(p, Rk, Rv, (hipStream_t)stream);
  return dispatch_prefill_lat<4>(p, Rk, Rv, (hipStream_t)stream);
}
