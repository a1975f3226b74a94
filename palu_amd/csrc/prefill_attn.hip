// Prefill attention over the latent value cache (the q_len > 1 branch, kernel/palu_attention.py:196-257):
//   out[t, h*Rv + c] = sum_j softmax_j( q~[h,t,:] . k~[h,j,:] * scale  [j <= past + t if causal] ) * V_lat[g, j, c]
// as a flash-style MFMA kernel: the [Tq x Tk] score matrix is never materialised (the reference builds it in
// full, palu_attention.py:205,238), P.V runs in the latent space (Rv columns per head, shared by the gs heads of a
// group, :246-251), softmax statistics are fp32 and online.
//
// Operand plan (everything "k-contiguous", so no LDS transposes):
//   S^T = K~ . Q~^T      A = K~ rows (kv), B = Q~ rows (t)            -> C layout: lane = query t, 16 kv per block
//   O^T = V^T . P^T      A = V^T rows (latent column c, kv contiguous), B = P^T = the lane's own S^T registers
// so the softmax statistics, the rescale of O and the final 1/l are all lane-local (lane = query).  The rows of
// the K~ A-fragment are read in the order "bits 2 and 3 of the row swapped", which makes the 8 registers of a
// k-step hold 8 CONSECUTIVE kv positions -- exactly what the V^T A-fragment supplies with one ds_read_b128.
// V^T ([G][Rv][Tk], kv contiguous, zero-padded to a multiple of 64) is a transient prefill workspace written by
// the host side; the latent cache itself keeps the decode layout.
//
// One 256-thread workgroup = 128 queries (32 per wave) x one head x one chunk of 32*NCB latent columns; K~/V^T
// tiles of 64 kv go through a double-buffered LDS image (XOR-swizzled 16-byte chunks, register-staged).
#include <cstdlib>
#include <type_traits>

#include "palu_common.h"

namespace {

constexpr int PF_THREADS = 256;
constexpr int PF_BM = 128;   // queries per workgroup
constexpr int PF_BN = 64;    // kv positions per tile

struct PfParams {
  const h16* q;
  const h16* k;
  const h16* vt;
  h16* out;
  int64_t sq_h, sq_t, sk_h, sk_t, sv_g, sv_c, so_t;
  int H, G, gs, Tq, Tk, Rv, past, causal;
  float scale_log2;   // scale * log2(e)
  int nqt;
  int head_major;     // dispatch order: 0 tile-major (all tiles of a head together), 1 head-major (heavy tiles of ALL heads
                      // first), 2 eight heads at a time, one per XCD, heavy tiles first (default beyond 2048 workgroups)
  // kv PANELS (palu_prefill_attn_panel_f16): k / vt hold the kv positions [kv0, kv0 + Tk) only and `past` is RELATIVE to the
  // panel (past_abs - kv0, may be negative: the whole causal logic is in panel-local indices).  The online-softmax state of
  // every (head, query) is carried from panel to panel in fp32: st_o [H][Tq][Rv] (un-normalised O), st_ml [Z][H][Tq][8]
  // (running maximum, then the partial sums of the lanes / waves that share a query; one slice per column block blockIdx.z:
  // the workgroups that split a query tile's latent columns over z all compute the same (m, l) and used to share ONE entry,
  // read at the start and overwritten at the end of the same launch -- a z-block could read what a faster sibling had
  // already advanced; with its own slice a block only ever reads what it wrote in the launch before).
  // first: start from the empty state;
  // last: normalise and store fp16 `out` instead of the state.  st_o == nullptr: the one-launch kernel.
  float* st_o;
  float* st_ml;
  int first, last;
};

// state of one lane: its running maximum, its share of the running sum (slot `ls` of the query's 8 floats) and its O columns
template <int NB>
static __device__ __forceinline__ void pf_state_load(const PfParams& p, int h, int qrow, bool qvalid, int c0, int hi, int ls,
                                                     f32x16 (&acc)[NB], float& m_run, float& l_run) {
  if (!qvalid) return;
  const float* ml = p.st_ml + (((int64_t)blockIdx.z * p.H + h) * p.Tq + qrow) * 8;
  m_run = ml[0];
  l_run = ml[1 + ls];
  const float* so = p.st_o + ((int64_t)h * p.Tq + qrow) * p.Rv + c0 + 4 * hi;
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
#pragma unroll
    for (int r2 = 0; r2 < 4; ++r2) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(so + 32 * cb + 8 * r2);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[cb][4 * r2 + e] = v[e];
    }
}
template <int NB>
static __device__ __forceinline__ void pf_state_store(const PfParams& p, int h, int qrow, bool qvalid, int c0, int hi, int ls,
                                                      const f32x16 (&acc)[NB], float m_run, float l_run) {
  if (!qvalid) return;
  float* ml = p.st_ml + (((int64_t)blockIdx.z * p.H + h) * p.Tq + qrow) * 8;
  ml[0] = m_run;                  // (every lane / wave of the query in this column block writes the same value)
  ml[1 + ls] = l_run;
  float* so = p.st_o + ((int64_t)h * p.Tq + qrow) * p.Rv + c0 + 4 * hi;
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
#pragma unroll
    for (int r2 = 0; r2 < 4; ++r2)
      *reinterpret_cast<f32x4*>(so + 32 * cb + 8 * r2) =
          f32x4{acc[cb][4 * r2], acc[cb][4 * r2 + 1], acc[cb][4 * r2 + 2], acc[cb][4 * r2 + 3]};
}

template <int NCB>
__global__ __launch_bounds__(PF_THREADS, 2) void prefill_attn_kernel(PfParams p) {
  constexpr int KS_BYTES = PF_BN * 256;            // K~ tile: 64 rows x 128 fp16
  constexpr int VS_BYTES = 32 * NCB * 128;         // V^T tile: 32*NCB rows (columns c) x 64 kv
  constexpr int VLD = NCB;                         // 16-byte chunks of the V^T tile per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks_base = smem;
  char* vs_base = smem + 2 * KS_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int n = lane & 31, hi = lane >> 5;
  // dispatch order (x fastest), heavy (late) query tiles first, chosen by the host (PfParams::head_major):
  // 1 = the heavy tiles of ALL heads first (best balance with few workgroups per CU); 2 = eight heads at a time, head
  // 8j + x on XCD x, so each XCD's L2 serves ONE head's K~ / V^T stream to its 32 CUs (best for long prompts);
  // 0 = all tiles of a head together.
  int qt_rev, h;
  if (p.head_major == 2) {                          // 8 heads at a time, one per XCD; within them heavy tiles first
    const int id = blockIdx.x;
    qt_rev = (id >> 3) % p.nqt;
    h = ((id >> 3) / p.nqt) * 8 + (id & 7);
  } else {
    qt_rev = p.head_major ? blockIdx.y : blockIdx.x;
    h = p.head_major ? blockIdx.x : blockIdx.y;
  }
  const int qt = p.nqt - 1 - qt_rev;
  const int g = h / p.gs;
  const int c0 = blockIdx.z * 32 * NCB;

  const int qrow = qt * PF_BM + w * 32 + n;
  const bool qvalid = qrow < p.Tq;
  const int qpos = p.past + qrow;

  // Q~ fragments (B operand of S^T): lane (t, hi) holds Q~[t][16ks + 8hi .. +7]
  h16x8 qf[8];
  {
    const h16* qp = p.q + (int64_t)h * p.sq_h + (int64_t)(qvalid ? qrow : 0) * p.sq_t + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + 16 * ks);
      if (!qvalid) v = u32x4{0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(h16x8, v);
    }
  }

  // number of kv tiles this workgroup needs
  int kv_end = p.Tk;
  if (p.causal) {
    const int last_q = min(p.Tq, (qt + 1) * PF_BM) - 1;
    kv_end = min(p.Tk, p.past + last_q + 1);
  }
  const int njt = (kv_end + PF_BN - 1) / PF_BN;

  // ---- staging: global -> registers -> LDS (swizzled 16-byte chunks)
  const h16* kg = p.k + (int64_t)h * p.sk_h;
  const h16* vg = p.vt + (int64_t)g * p.sv_g + (int64_t)c0 * p.sv_c;
  u32x4 kreg[4], vreg[VLD];
  auto load_k = [&](int jt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = tid + PF_THREADS * i;
      const int row = s >> 4, ch = s & 15;
      const int j = min(jt * PF_BN + row, p.Tk - 1);         // rows past the end repeat the last one (masked below)
      kreg[i] = *reinterpret_cast<const u32x4*>(kg + (int64_t)j * p.sk_t + ch * 8);
    }
  };
  auto load_v = [&](int jt) {
#pragma unroll
    for (int i = 0; i < VLD; ++i) {
      const int s = tid + PF_THREADS * i;
      const int row = s >> 3, ch = s & 7;
      vreg[i] = *reinterpret_cast<const u32x4*>(vg + (int64_t)row * p.sv_c + jt * PF_BN + ch * 8);   // zero-padded rows
    }
  };
  auto store_tile = [&](int buf) {
    char* ks = ks_base + buf * KS_BYTES;
    char* vs = vs_base + buf * VS_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = tid + PF_THREADS * i;
      const int row = s >> 4, ch = s & 15;
      *reinterpret_cast<u32x4*>(ks + row * 256 + ((ch ^ (row & 15)) << 4)) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < VLD; ++i) {
      const int s = tid + PF_THREADS * i;
      const int row = s >> 3, ch = s & 7;
      *reinterpret_cast<u32x4*>(vs + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = vreg[i];
    }
  };

  f32x16 acc_o[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc_o[cb][e] = 0.f;
  float m_run = -INFINITY;   // running max of the raw scores (scale > 0 is applied inside the exponent)
  float l_run = 0.f;         // this lane's half of the running sum (the halves share m_run)
  if (p.st_o && !p.first) pf_state_load<NCB>(p, h, qrow, qvalid, c0, hi, hi, acc_o, m_run, l_run);

  // K~ A-fragment row of this lane: bits 2 and 3 of the MFMA row swapped (see header)
  const int krow = (n & 0x13) | ((n & 4) << 1) | ((n & 8) >> 1);

  if (njt > 0) {
    load_k(0);
    load_v(0);
    store_tile(0);
  }
  __syncthreads();

  for (int jt = 0; jt < njt; ++jt) {
    const int buf = jt & 1;
    const char* ks = ks_base + buf * KS_BYTES;
    const char* vs = vs_base + buf * VS_BYTES;

    // ---- S^T = K~ . Q~^T : two 32-kv row blocks
    f32x16 sacc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[rb][e] = 0.f;
      const int row = rb * 32 + krow;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const h16x8 kf = *reinterpret_cast<const h16x8*>(ks + row * 256 + (((2 * kk + hi) ^ (row & 15)) << 4));
        sacc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], sacc[rb], 0, 0, 0);
      }
    }
    // register r of block rb in lane (t, hi) is kv = jt*64 + 32rb + 16(r>>3) + 8hi + (r&7)
    const int j0 = jt * PF_BN + 8 * hi;
    const bool need_mask = (jt * PF_BN + PF_BN > p.Tk) || (p.causal && jt * PF_BN + PF_BN - 1 > p.past + qt * PF_BM + w * 32);
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
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * p.scale_log2);             // m_run = -inf -> 0
    const float moff = -m_use * p.scale_log2;
    float lsum = 0.f;
    h16x8 pf[4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        h16x8 pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[rb][8 * s2 + e], p.scale_log2, moff));
          lsum += pv;
          pk[e] = (h16)pv;
        }
        pf[2 * rb + s2] = pk;
      }
    l_run = fmaf(l_run, alpha, lsum);
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {                 // wave-uniform: rescale only when a max moved
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[cb][e] *= alpha;
    }

    // next tile: issued once the score registers are dead (keeps the kernel at 2 workgroups per CU), landed in LDS
    // after the P.V MFMAs
    asm volatile("" ::: "memory");
    if (jt + 1 < njt) {
      load_k(jt + 1);
      load_v(jt + 1);
    }

    // ---- O^T += V^T . P^T
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int row = cb * 32 + n;
        const h16x8 vf = *reinterpret_cast<const h16x8*>(vs + row * 128 + (((2 * s + hi) ^ ((row >> 1) & 7)) << 4));
        acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], acc_o[cb], 0, 0, 0);
      }

    if (jt + 1 < njt) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (p.st_o && !p.last) {
    pf_state_store<NCB>(p, h, qrow, qvalid, c0, hi, hi, acc_o, m_run, l_run);
    return;
  }
  // ---- epilogue: 1/l (both halves), fp16 store; lane (t, hi) register r of block cb is column 32cb + (r&3) + 8(r>>2) + 4hi
  {
    const unsigned lb = __float_as_uint(l_run);
    auto sw = __builtin_amdgcn_permlane32_swap(lb, lb, false, false);
    const float l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qvalid) {
      h16* op = p.out + (int64_t)qrow * p.so_t + (int64_t)h * p.Rv + c0 + 4 * hi;
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
}

// ---------------------------------------------------------------------------------------------
// Pair variant for wide latent values (Rv = 64*NCBH, e.g. 384): 8 waves; waves w and w+4 share 32 queries.
// Each computes S^T for ONE 32-kv half of the tile and owns ONE half of the latent columns; the two exchange
// the half-tile maxima and their fp16 P fragments through LDS, so q.k^T is computed once (the 4-wave kernel
// above would run twice over the keys, once per 192-column chunk) and the softmax VALU work per wave halves.
constexpr int PFP_THREADS = 512;

template <int NCBH>
__global__ __launch_bounds__(PFP_THREADS, 1) void prefill_attn_pair_kernel(PfParams p) {
  constexpr int RV = 64 * NCBH;
  constexpr int KS_BYTES = PF_BN * 256;
  constexpr int VS_BYTES = RV * 128;
  constexpr int KLD = 2;                              // 16-byte chunks per thread: K~ tile
  constexpr int VLD = RV * 8 / PFP_THREADS;           // V^T tile
  static_assert(RV * 8 % PFP_THREADS == 0, "V^T tile must split evenly over the workgroup");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks_base = smem;
  char* vs_base = smem + 2 * KS_BYTES;
  float* mx = reinterpret_cast<float*>(vs_base + 2 * VS_BYTES);          // [4 qblk][2 half][32]
  char* px = reinterpret_cast<char*>(mx) + 4 * 2 * 32 * sizeof(float);   // [4 qblk][2 half][2 ksteps][64 lanes][16 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qblk = w & 3, half = w >> 2;
  const int n = lane & 31, hi = lane >> 5;
  int qt_rev, h;
  if (p.head_major == 2) {   // see prefill_attn_kernel
    const int id = blockIdx.x;
    qt_rev = (id >> 3) % p.nqt;
    h = ((id >> 3) / p.nqt) * 8 + (id & 7);
  } else {
    qt_rev = p.head_major ? blockIdx.y : blockIdx.x;
    h = p.head_major ? blockIdx.x : blockIdx.y;
  }
  const int qt = p.nqt - 1 - qt_rev;
  const int g = h / p.gs;
  const int c0 = half * 32 * NCBH;

  const int qrow = qt * PF_BM + qblk * 32 + n;
  const bool qvalid = qrow < p.Tq;
  const int qpos = p.past + qrow;

  h16x8 qf[8];
  {
    const h16* qp = p.q + (int64_t)h * p.sq_h + (int64_t)(qvalid ? qrow : 0) * p.sq_t + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + 16 * ks);
      if (!qvalid) v = u32x4{0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(h16x8, v);
    }
  }

  int kv_end = p.Tk;
  if (p.causal) {
    const int last_q = min(p.Tq, (qt + 1) * PF_BM) - 1;
    kv_end = min(p.Tk, p.past + last_q + 1);
  }
  const int njt = (kv_end + PF_BN - 1) / PF_BN;

  // ---- staging by LDS-DMA (buffer_load_dwordx4 ... lds, see abx_rope_kernel.h): a wave-instruction fills 64
  //      consecutive 16-byte LDS slots (4 K~ rows / 8 V^T rows); the XOR swizzle sits in the per-lane SOURCE offset,
  //      which does not depend on the piece (pieces of a wave are 8 apart), the tile/piece select is scalar.  K~ rows
  //      past the end are outside the descriptor (masked below whatever they read as); V^T is zero-padded by contract.
  const h16* kg = p.k + (int64_t)h * p.sk_h;
  const h16* vg = p.vt + (int64_t)g * p.sv_g;
  auto make_rsrc = [](const void* base, int64_t bytes) {
    u32x4 r;
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
    r[3] = 0x00020000u;
    return r;
  };
  const u32x4 krs = make_rsrc(kg, ((int64_t)(p.Tk - 1) * p.sk_t + 128) * 2);
  const u32x4 vrs = make_rsrc(vg, ((int64_t)(RV - 1) * p.sv_c + (int64_t)((p.Tk + PF_BN - 1) / PF_BN) * PF_BN) * 2);
  const int krow_l = 4 * w + (lane >> 4), vrow_l = 8 * w + (lane >> 3);
  const unsigned kvo = (unsigned)((krow_l * p.sk_t + (((lane & 15) ^ (krow_l & 15)) << 3)) * 2);
  const unsigned vvo = (unsigned)((vrow_l * p.sv_c + (((lane & 7) ^ ((vrow_l >> 1) & 7)) << 3)) * 2);
  const unsigned ktile_bytes = __builtin_amdgcn_readfirstlane((unsigned)(PF_BN * p.sk_t * 2));
  const unsigned kstep_bytes = __builtin_amdgcn_readfirstlane((unsigned)(32 * p.sk_t * 2));
  const unsigned vstep_bytes = __builtin_amdgcn_readfirstlane((unsigned)(64 * p.sv_c * 2));
  const unsigned smem_lds = (unsigned)reinterpret_cast<uintptr_t>(smem);
  auto dma = [&](unsigned dst, unsigned voff, const u32x4& rs, unsigned soff) {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(dst), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
  };
  auto dma_k = [&](int jt, int buf) {
#pragma unroll
    for (int i = 0; i < KLD; ++i)
      dma(__builtin_amdgcn_readfirstlane(smem_lds + buf * KS_BYTES + (w + 8 * i) * 1024), kvo, krs,
          __builtin_amdgcn_readfirstlane((unsigned)jt * ktile_bytes + i * kstep_bytes));
  };
  auto dma_v = [&](int jt, int buf) {
#pragma unroll
    for (int i = 0; i < VLD; ++i)
      dma(__builtin_amdgcn_readfirstlane(smem_lds + 2 * KS_BYTES + buf * VS_BYTES + (w + 8 * i) * 1024), vvo, vrs,
          __builtin_amdgcn_readfirstlane((unsigned)jt * (PF_BN * 2) + i * vstep_bytes));
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  f32x16 acc_o[NCBH];
#pragma unroll
  for (int cb = 0; cb < NCBH; ++cb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc_o[cb][e] = 0.f;
  float m_run = -INFINITY;
  float l_run = 0.f;   // this lane's share: its hi-half of this wave's kv half
  if (p.st_o && !p.first) pf_state_load<NCBH>(p, h, qrow, qvalid, c0, hi, 2 * half + hi, acc_o, m_run, l_run);

  // lane-constant LDS byte offsets of the fragments (tile buffer and column block ride in the immediates)
  const int krow = half * 32 + ((n & 0x13) | ((n & 4) << 1) | ((n & 8) >> 1));
  unsigned ka[8], va[4];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) ka[kk] = (unsigned)(krow * 256 + (((2 * kk + hi) ^ (krow & 15)) << 4));
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int ksabs = (s < 2) ? 2 * half + s : 2 * (half ^ 1) + (s - 2);     // own k-steps first, then the partner's
    va[s] = (unsigned)(2 * KS_BYTES + (c0 + n) * 128 + (((2 * ksabs + hi) ^ ((n >> 1) & 7)) << 4));
  }
  float* mx_own = mx + (qblk * 2 + half) * 32 + n;
  const float* mx_par = mx + (qblk * 2 + (half ^ 1)) * 32 + n;
  char* px_own = px + ((qblk * 2 + half) * 2) * 1024 + lane * 16;
  const char* px_par = px + ((qblk * 2 + (half ^ 1)) * 2) * 1024 + lane * 16;

  // score MFMAs of tile jt (K~ in buffer jt&1), masked, plus this wave's half-tile maximum (both hi halves)
  auto scores = [&](int jt, f32x16& sacc) {
    const char* kt = smem + (jt & 1) * KS_BYTES;
    h16x8 kf[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) kf[kk] = *reinterpret_cast<const h16x8*>(kt + ka[kk]);
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kk], qf[kk], sacc, 0, 0, 0);
  };
  auto mask_max = [&](int jt, f32x16& sacc) {
    // register r in lane (t, hi) is kv = jt*64 + 32*half + 16(r>>3) + 8hi + (r&7)
    const int j0 = jt * PF_BN + 32 * half + 8 * hi;
    const bool need_mask = (jt * PF_BN + PF_BN > p.Tk) || (p.causal && jt * PF_BN + PF_BN - 1 > p.past + qt * PF_BM + qblk * 32);
    if (need_mask) {
      const int lim = p.causal ? min(p.Tk - 1, qpos) : p.Tk - 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = j0 + 16 * (r >> 3) + (r & 7);
        if (j > lim) sacc[r] = -INFINITY;
      }
    }
    float mloc = sacc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, sacc[r]);
    const unsigned mb = __float_as_uint(mloc);
    auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  };

  // ---- software pipeline, two tiles deep: iteration jt interleaves the P.V MFMAs of tile jt-1 with the softmax
  //      VALU work of tile jt (one exp/convert slice per MFMA gap), then runs the score MFMAs of tile jt+1.
  //      K~ tile j lives in buffer j&1 (read in iteration j-1), V^T tile j in buffer j&1 (read in iteration j+1).
  f32x16 s_cur, s_nxt;
  float mloc_cur = -INFINITY;
  h16x8 pf_prev[4];                                     // P^T fragments of the previous tile (own two, partner's two)
  if (njt > 0) {
    dma_k(0, 0);
    dma_v(0, 0);
    if (njt > 1) dma_k(1, 1);
    dma_wait();
  }
  __syncthreads();
  if (njt > 0) {
    scores(0, s_cur);
    mloc_cur = mask_max(0, s_cur);
    *mx_own = mloc_cur;
  }
  __syncthreads();

  constexpr int GAPS = 4 * NCBH;
  constexpr int EPG = (16 + GAPS - 1) / GAPS;           // softmax elements per MFMA gap
  for (int jt = 0; jt <= njt; ++jt) {                   // iteration njt only drains the last tile's P.V
    const bool have_sm = jt < njt;                      // a tile to turn into probabilities
    const bool have_pv = jt > 0;                        // a previous tile to multiply with V
    const char* vt = smem + ((jt + 1) & 1) * VS_BYTES;  // V^T tile jt-1
    auto vfrag = [&](int s, int cb) { return *reinterpret_cast<const h16x8*>(vt + va[s] + cb * 32 * 128); };
    if (jt + 2 < njt) dma_k(jt + 2, jt & 1);            // K~ buffer jt&1: tile jt was consumed in iteration jt-1
    float m_new = m_run, alpha = 1.0f, moff = 0.f, lsum = 0.f;
    if (have_sm) {
      m_new = fmaxf(m_run, fmaxf(mloc_cur, *mx_par));   // partner's maximum: written before the last barrier
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      alpha = __builtin_amdgcn_exp2f((m_run - m_use) * p.scale_log2);
      moff = -m_use * p.scale_log2;
    }
    h16x8 pk[2];
    auto softmax_slice = [&](int e) {                   // element e of this lane's 16 scores of tile jt
      const float pv = __builtin_amdgcn_exp2f(fmaf(s_cur[e], p.scale_log2, moff));
      lsum += pv;
      pk[e >> 3][e & 7] = (h16)pv;
      if ((e & 7) == 7) *reinterpret_cast<h16x8*>(px_own + (e >> 3) * 1024) = pk[e >> 3];
    };
    if (have_pv) {
      h16x8 vf[2][NCBH];
#pragma unroll
      for (int cb = 0; cb < NCBH; ++cb) vf[0][cb] = vfrag(0, cb);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) {
#pragma unroll
          for (int cb = 0; cb < NCBH; ++cb) vf[(s + 1) & 1][cb] = vfrag(s + 1, cb);
        }
#pragma unroll
        for (int cb = 0; cb < NCBH; ++cb) {
          acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[s & 1][cb], pf_prev[s], acc_o[cb], 0, 0, 0);
          if (have_sm) {
#pragma unroll
            for (int q = 0; q < EPG; ++q)
              if ((s * NCBH + cb) * EPG + q < 16) softmax_slice((s * NCBH + cb) * EPG + q);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if (have_sm) {
#pragma unroll
      for (int e = 0; e < 16; ++e) softmax_slice(e);
    }
    if (have_sm) {
      l_run = fmaf(l_run, alpha, lsum);
      m_run = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // O (incl. tile jt-1) moves to the new maximum
#pragma unroll
        for (int cb = 0; cb < NCBH; ++cb)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc_o[cb][e] *= alpha;
      }
      pf_prev[0] = pk[0];
      pf_prev[1] = pk[1];
    }
    __syncthreads();                                    // B: P fragments of tile jt visible; V^T buffer (jt+1)&1 free
    if (have_sm) {
      pf_prev[2] = *reinterpret_cast<const h16x8*>(px_par);
      pf_prev[3] = *reinterpret_cast<const h16x8*>(px_par + 1024);
    }
    if (jt + 1 < njt) {
      dma_v(jt + 1, (jt + 1) & 1);                      // consumed in iteration jt+2
      scores(jt + 1, s_nxt);
      mloc_cur = mask_max(jt + 1, s_nxt);
      *mx_own = mloc_cur;
      s_cur = s_nxt;
    }
    dma_wait();
    __syncthreads();                                    // A: maxima + staged tiles visible, exchange slots free
  }

  if (p.st_o && !p.last) {
    pf_state_store<NCBH>(p, h, qrow, qvalid, c0, hi, 2 * half + hi, acc_o, m_run, l_run);
    return;
  }
  // ---- epilogue: total l = both hi halves of both waves of the pair
  {
    const unsigned lb = __float_as_uint(l_run);
    auto sw = __builtin_amdgcn_permlane32_swap(lb, lb, false, false);
    const float l_half = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    *mx_own = l_half;
    __syncthreads();
    const float l_tot = l_half + *mx_par;
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qvalid) {
      h16* op = p.out + (int64_t)qrow * p.so_t + (int64_t)h * p.Rv + c0 + 4 * hi;
#pragma unroll
      for (int cb = 0; cb < NCBH; ++cb)
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) {
          h16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (h16)(acc_o[cb][4 * r2 + e] * inv);
          *reinterpret_cast<h16x4*>(op + 32 * cb + 8 * r2) = o;
        }
    }
  }
}

template <int NCBH>
int launch_prefill_pair(const PfParams& p, hipStream_t stream) {
  constexpr int smem = 2 * (PF_BN * 256 + 64 * NCBH * 128) + 4 * 2 * 32 * (int)sizeof(float) + 4 * 2 * 2 * 1024;
  auto kern = prefill_attn_pair_kernel<NCBH>;
  const int rca = palu_func_max_lds(reinterpret_cast<const void*>(kern), smem);
  if (rca) return rca;
  dim3 grid(p.head_major == 2 ? p.H * p.nqt : (p.head_major ? p.H : p.nqt), p.head_major == 2 ? 1 : (p.head_major ? p.nqt : p.H), 1);
  hipLaunchKernelGGL(kern, grid, dim3(PFP_THREADS), smem, stream, p);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

template <int NCB>
int launch_prefill(const PfParams& p, hipStream_t stream) {
  constexpr int smem = 2 * (PF_BN * 256 + 32 * NCB * 128);
  auto kern = prefill_attn_kernel<NCB>;
  const int rca = palu_func_max_lds(reinterpret_cast<const void*>(kern), smem);
  if (rca) return rca;
  dim3 grid(p.head_major == 2 ? p.H * p.nqt : (p.head_major ? p.H : p.nqt), p.head_major == 2 ? 1 : (p.head_major ? p.nqt : p.H), p.Rv / (32 * NCB));
  hipLaunchKernelGGL(kern, grid, dim3(PF_THREADS), smem, stream, p);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

}  // namespace

static int prefill_impl(const void* q, int64_t sq_h, int64_t sq_t, const void* k, int64_t sk_h, int64_t sk_t, const void* vt,
                        int64_t sv_g, int64_t sv_c, void* out, int64_t so_t, int H, int G, int D, int Tq, int Tk, int Rv,
                        int past, int causal, float scale, float* st_o, float* st_ml, int first, int last,
                        palu_stream_t stream);

extern "C" int palu_prefill_attn_f16(const void* q, int64_t sq_h, int64_t sq_t, const void* k, int64_t sk_h, int64_t sk_t,
                                     const void* vt, int64_t sv_g, int64_t sv_c, void* out, int64_t so_t, int H, int G,
                                     int D, int Tq, int Tk, int Rv, int past, int causal, float scale,
                                     palu_stream_t stream) {
  PALU_REQUIRE(past >= 0, PALU_ERR_ARG, "prefill_attn: negative length");
  return prefill_impl(q, sq_h, sq_t, k, sk_h, sk_t, vt, sv_g, sv_c, out, so_t, H, G, D, Tq, Tk, Rv, past, causal, scale, nullptr,
                      nullptr, 1, 1, stream);
}

// One kv PANEL of a prompt pass whose keys / values do not fit the workspace at once: k / vt hold the kv positions
// [kv0, kv0 + Tk) only, `past_rel` = (absolute position of the first query) - kv0 (negative when the panel starts behind the
// first query; a panel that lies entirely in a query tile's causal future leaves that tile's state untouched).  The
// online-softmax state travels in state_o [H][Tq][Rv] fp32 and state_ml [Rv / 32][H][Tq][8] fp32 (one slice per 32-column block; size both with palu_prefill_state_bytes):
// first != 0 starts from the empty state, last != 0 normalises and writes `out` (fp16) instead of the state.  Panels of one
// (query chunk, head set) are launched in ascending kv order on one stream.  Same results as one launch over all panels
// up to the fp32 rounding of the rescale order (the running maximum moves at panel boundaries exactly as at tile boundaries).
extern "C" int palu_prefill_attn_panel_f16(const void* q, int64_t sq_h, int64_t sq_t, const void* k, int64_t sk_h,
                                           int64_t sk_t, const void* vt, int64_t sv_g, int64_t sv_c, void* out, int64_t so_t,
                                           int H, int G, int D, int Tq, int Tk, int Rv, int past_rel, int causal,
                                           float scale, void* state_o, void* state_ml, int first, int last,
                                           palu_stream_t stream) {
  PALU_REQUIRE(state_o && state_ml, PALU_ERR_ARG, "prefill_attn_panel: null state");
  PALU_REQUIRE((((uintptr_t)state_o | (uintptr_t)state_ml) & 15) == 0, PALU_ERR_ARG, "prefill_attn_panel: state must be 16-byte aligned");
  return prefill_impl(q, sq_h, sq_t, k, sk_h, sk_t, vt, sv_g, sv_c, out, so_t, H, G, D, Tq, Tk, Rv, past_rel, causal, scale,
                      (float*)state_o, (float*)state_ml, first ? 1 : 0, last ? 1 : 0, stream);
}

// bytes of (state_o, state_ml) for H heads x Tq queries
extern "C" size_t palu_prefill_state_bytes(int H, int Tq, int Rv, int which) {
  if (H <= 0 || Tq <= 0 || Rv <= 0) return 0;
  // (state_ml: one [H][Tq][8] slice per 32-column block -- the most column blocks a launch can split a query tile into)
  return which == 0 ? (size_t)H * Tq * Rv * sizeof(float) : (size_t)H * Tq * 8 * sizeof(float) * ((Rv + 31) / 32);
}

static int prefill_impl(const void* q, int64_t sq_h, int64_t sq_t, const void* k, int64_t sk_h, int64_t sk_t, const void* vt,
                        int64_t sv_g, int64_t sv_c, void* out, int64_t so_t, int H, int G, int D, int Tq, int Tk, int Rv,
                        int past, int causal, float scale, float* st_o, float* st_ml, int first, int last,
                        palu_stream_t stream) {
  PALU_REQUIRE(q && k && vt && out, PALU_ERR_ARG, "prefill_attn: null pointer");
  PALU_REQUIRE(H > 0 && G > 0 && H % G == 0, PALU_ERR_ARG, "prefill_attn: bad heads/groups H=%d G=%d", H, G);
  PALU_REQUIRE(D == 128, PALU_ERR_UNSUPPORTED, "prefill_attn: head_dim must be 128 (got %d)", D);
  PALU_REQUIRE(Tq >= 0 && Tk >= 0, PALU_ERR_ARG, "prefill_attn: negative length");
  PALU_REQUIRE(past > -(1 << 30) && (int64_t)past + Tq < (1 << 30), PALU_ERR_ARG, "prefill_attn: position offset out of range");
  if (Tq == 0) return PALU_OK;
  PALU_REQUIRE(Tk > 0, PALU_ERR_ARG, "prefill_attn: no keys");
  PALU_REQUIRE(Rv > 0 && Rv % 32 == 0, PALU_ERR_UNSUPPORTED, "prefill_attn: latent value rank per group must be a multiple of 32 (got %d)", Rv);
  PALU_REQUIRE(scale > 0.f, PALU_ERR_ARG, "prefill_attn: scale must be positive");
  PALU_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) == 0 && sq_h % 8 == 0 && sq_t % 8 == 0 &&
                   sk_h % 8 == 0 && sk_t % 8 == 0 && sv_g % 8 == 0 && sv_c % 8 == 0 && ((uintptr_t)out & 7) == 0 &&
                   so_t % 4 == 0,
               PALU_ERR_ARG, "prefill_attn: rows must be 16-byte aligned (out 8-byte)");
  PALU_REQUIRE(((int64_t)Tk + PF_BN) * sk_t * 2 < ((int64_t)1 << 32) && (int64_t)Rv * sv_c * 2 < ((int64_t)1 << 32),
               PALU_ERR_UNSUPPORTED, "prefill_attn: one head's keys / one group's values must stay below 4 GiB");
  const int tk_pad = (Tk + PF_BN - 1) / PF_BN * PF_BN;
  PALU_REQUIRE(sv_c >= tk_pad, PALU_ERR_ARG,
               "prefill_attn: vt rows must be zero-padded to a multiple of %d kv positions (sv_c=%lld < %d)", PF_BN,
               (long long)sv_c, tk_pad);
  PfParams p;
  p.q = (const h16*)q; p.k = (const h16*)k; p.vt = (const h16*)vt; p.out = (h16*)out;
  p.sq_h = sq_h; p.sq_t = sq_t; p.sk_h = sk_h; p.sk_t = sk_t; p.sv_g = sv_g; p.sv_c = sv_c; p.so_t = so_t;
  p.H = H; p.G = G; p.gs = H / G; p.Tq = Tq; p.Tk = Tk; p.Rv = Rv; p.past = past; p.causal = causal ? 1 : 0;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.nqt = (Tq + PF_BM - 1) / PF_BM;
  p.st_o = st_o; p.st_ml = st_ml; p.first = first; p.last = last;
  {
    static int force = -2;                       // PALU_PREFILL_HEAD_MAJOR = 0 / 1 overrides the size rule
    if (force == -2) {
      const char* e = palu_exp_env("PALU_PREFILL_HEAD_MAJOR");
      force = e ? atoi(e) : -1;
    }
    p.head_major = force >= 0 ? force : ((int64_t)p.nqt * H <= 2048 ? 1 : 2);
    if (p.head_major == 2 && H % 8 != 0) p.head_major = 0;
  }
  hipStream_t s = (hipStream_t)stream;
  static int use_pair = -1;
  if (use_pair < 0) {
    const char* e = palu_exp_env("PALU_PREFILL_PAIR");
    use_pair = e ? atoi(e) : 1;
  }
  // a launch too small to give every CU a workgroup (kv panels under a tight memory bound: few queries x few heads) takes the
  // 4-wave kernel with the latent columns split over blockIdx.z instead: more, smaller workgroups for the price of recomputed scores
  const bool small = st_o != nullptr && (int64_t)p.nqt * H * 2 <= palu_num_cus();
  if (small && Rv % 96 == 0) return launch_prefill<3>(p, s);
  if (small && Rv % 64 == 0) return launch_prefill<2>(p, s);
  if (use_pair && Rv % 64 == 0 && Rv <= 384) {          // two waves per 32 queries, each half of the latent columns
    switch (Rv / 64) {
      case 1: return launch_prefill_pair<1>(p, s);
      case 2: return launch_prefill_pair<2>(p, s);
      case 3: return launch_prefill_pair<3>(p, s);
      case 4: return launch_prefill_pair<4>(p, s);
      case 5: return launch_prefill_pair<5>(p, s);
      default: return launch_prefill_pair<6>(p, s);
    }
  }
  if (Rv % 192 == 0) return launch_prefill<6>(p, s);
  if (Rv % 96 == 0) return launch_prefill<3>(p, s);
  if (Rv % 64 == 0) return launch_prefill<2>(p, s);
  return launch_prefill<1>(p, s);
}
