// Prefill down-projection: latents[L, rank] = X[L, hidden] . VT^T  (kernel/palu_attention.py:167-168 with
// q_len = L; HeadwiseLowRankModule.project_to_latent :59-65) -- the one true dense GEMM of the module, on MFMA.
//
// Both operands are K-contiguous (X rows, VT rows), i.e. exactly the MFMA fragment shape (a lane holds 8
// consecutive k of one row) -- no transposes anywhere.  128x128 output tile per 256-thread workgroup
// (2x2 waves x 2x2 v_mfma_f32_32x32x16_f16 tiles), BK = 64, register-staged double-buffered LDS tiles with an
// XOR swizzle (conflict-free ds_read_b128).  The MFMA computes C^T tiles (A = VT rows, B = X rows) so that a
// lane ends up with 4 consecutive latent columns of one token: 8-byte stores, written straight into the
// [G, L, R] latent-cache layout (row offset `row0`), so prefill fills the cache without a reshape/copy.
#include "palu_common.h"

namespace {

constexpr int PG_BM = 128, PG_BN = 128, PG_BK = 64, PG_THREADS = 256;

struct PgParams {
  const h16* x;  int64_t ldx;       // [M, K]
  const h16* w;  int64_t ldw;       // [N, K]
  h16* out;      int64_t so_g, so_l; // out[(n / R) * so_g + (row0 + m) * so_l + n % R]
  int M, N, K, R, row0;
  // fused quantise + pack epilogue (project_gemm_256_kernel<NI, 3 | 4>): packed rows and (scale, zero) pairs of the latent cache
  unsigned char* codes; int64_t sc_g, sc_l;   // bytes: codes[(n / R) * sc_g + (row0 + m) * sc_l + bit stream of the row]
  h16* meta;            int64_t sm_g, sm_l;   // elements: meta[(n / R) * sm_g + (row0 + m) * sm_l + {0, 1}]
};

__device__ __forceinline__ int pg_swz(int row, int c) { return c ^ ((row >> 1) & 7); }   // 128-byte rows, 8 chunks

__global__ __launch_bounds__(PG_THREADS, 2) void project_gemm_kernel(PgParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * PG_BM * PG_BK * 2];   // [buf][A|B][128 rows][64 halfs] = 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;                 // wave grid: 2 (token) x 2 (latent column)
  const int m0 = blockIdx.x * PG_BM, n0 = blockIdx.y * PG_BN;
  constexpr int TILE = PG_BM * PG_BK * 2;              // bytes of one operand tile

  // staging: 128 rows x 8 chunks = 1024 slots per operand, 4 per thread
  u32x4 ra[4], rb[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int slot = tid + PG_THREADS * s, row = slot >> 3, c = pg_swz(row, slot & 7);
      const int xm = min(m0 + row, p.M - 1), wn_ = min(n0 + row, p.N - 1);
      ra[s] = *reinterpret_cast<const u32x4*>(p.x + (int64_t)xm * p.ldx + kt * PG_BK + c * 8);
      rb[s] = *reinterpret_cast<const u32x4*>(p.w + (int64_t)wn_ * p.ldw + kt * PG_BK + c * 8);
    }
  };
  auto sstore = [&](int buf) {
    char* a = smem + buf * 2 * TILE;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int slot = tid + PG_THREADS * s;
      *reinterpret_cast<u32x4*>(a + slot * 16) = ra[s];            // X tile
      *reinterpret_cast<u32x4*>(a + TILE + slot * 16) = rb[s];     // VT tile
    }
  };

  f32x16 acc[2][2];   // [latent-column tile][token tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K / PG_BK;
  gload(0);
  sstore(0);
  if (nk > 1) gload(1);
  const int fr = lane & 31, kb = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    if (kt + 1 < nk) sstore((kt + 1) & 1);
    if (kt + 2 < nk) gload(kt + 2);
    const char* xs = smem + (kt & 1) * 2 * TILE;
    const char* ws = xs + TILE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      h16x8 fx[2], fw[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int rx = wm * 64 + t * 32 + fr, rw = wn * 64 + t * 32 + fr;
        fx[t] = *reinterpret_cast<const h16x8*>(xs + rx * 128 + pg_swz(rx, 2 * ks + kb) * 16);
        fw[t] = *reinterpret_cast<const h16x8*>(ws + rw * 128 + pg_swz(rw, 2 * ks + kb) * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i], fx[j], acc[i][j], 0, 0, 0);
    }
  }
  // C^T tile: column (lane&31) = token, rows = latent columns (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + wm * 64 + j * 32 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + i * 32 + 8 * q + 4 * kb;
        if (n + 3 < p.N) {
          const int g = n / p.R, r = n - g * p.R;
          h16x4 v = {(h16)acc[i][j][4 * q], (h16)acc[i][j][4 * q + 1], (h16)acc[i][j][4 * q + 2], (h16)acc[i][j][4 * q + 3]};
          *reinterpret_cast<h16x4*>(p.out + (int64_t)g * p.so_g + (int64_t)(p.row0 + m) * p.so_l + r) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) {
              const int g = (n + e) / p.R, r = (n + e) - g * p.R;
              p.out[(int64_t)g * p.so_g + (int64_t)(p.row0 + m) * p.so_l + r] = (h16)acc[i][j][4 * q + e];
            }
        }
      }
    }
}

// ---- the large-shape path: 256x256 output tile per 512-thread workgroup (1 per CU), BK = 64 -----------------------
// 8 waves as 2 (token) x 4 (latent column); a wave owns 128 tokens x 64 columns = 4 x 2 MFMA tiles (128 accumulator
// registers), so a k-step costs 6 ds_read_b128 per 8 MFMAs (the 128x128 kernel above: 4 per 4).  Both operand tiles
// are staged by LDS-DMA (buffer_load_dwordx4 ... lds): a wave-instruction fills 8 rows x 128 B of the tile image, the
// XOR swizzle sits in the per-lane SOURCE offset (it depends on the wave's parity only, not on the piece), rows past
// the end are clamped in the lane offsets once, and the loop itself holds no address arithmetic: 8 DMA instructions
// per wave per k-tile with a scalar offset.  Two LDS buffers (2 x 64 KB), tile kt+1 in flight under the MFMAs of tile kt,
// `s_waitcnt vmcnt(0)` + barrier per k-tile.  Workgroup -> tile: the 8 XCDs each take every 8th token tile and run
// its column tiles back to back, so the X rows of a token tile are fetched into ONE L2.
constexpr int PL_BM = 256, PL_THREADS = 512;   // column tile: 128 * NI (NI = 2: 256 x 256; NI = 1: 256 x 128 for grids that would not fill the chip)

// QB = 3 / 4: the tile is not stored as fp16 latents but quantised and packed in the epilogue -- quantize_tensor's
// asymmetric per-(token, group) row form (palu/model/modules/quant.py:29-39 with the reference defaults group_size = 0,
// clip_ratio = 1; csrc/quant.hip has the op-by-op fp16 semantics), bit-identical to this kernel's fp16 output run through
// palu_quantize_pack.  A workgroup's 128 NI columns hold whole groups (host: 32 NI | R, R | 128 NI): the R / (32 NI) waves of
// a group exchange their row minima / maxima through the (then free) LDS, every lane derives (scale, zero) itself, codes its
// 16 NI values per token, and lane pairs (kb = 0, 1: columns 8q .. 8q+3 / 8q+4 .. 8q+7) assemble 32 codes = 16 / 12 bytes
// per store.  The fp16 latents of a packed-cache prompt pass never exist in memory.
template <int NI, int QB = 0>
__global__ __launch_bounds__(PL_THREADS, 1) void project_gemm_256_kernel(PgParams p, int ntn, int ntm) {
  extern __shared__ __attribute__((aligned(1024))) char smem_l[];   // [buf][X 256 rows | VT 128 NI rows][64 halfs] = 2 x (32 + 16 NI) KB
  constexpr int PL_BN = 128 * NI;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 2, wn = wv & 3;
  const int b = blockIdx.x, j = b >> 3;
  const int mt = (j / ntn) * 8 + (b & 7), nt = j % ntn;
  if (mt >= ntm) return;
  const int m0 = mt * PL_BM, n0 = nt * PL_BN;
  constexpr int TILE = PL_BM * PG_BK * 2;   // 32 KB: X tile
  constexpr int WTILE = PL_BN * PG_BK * 2;  // VT tile
  constexpr int BUF = TILE + WTILE;

  auto make_rsrc = [](const void* base) {
    u32x4 r;
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    r[2] = 0x7fffffffu;      // rows are clamped by hand below; k never leaves the row
    r[3] = 0x00020000u;
    return r;
  };
  const u32x4 xrs = make_rsrc(p.x + (int64_t)m0 * p.ldx), wrs = make_rsrc(p.w + (int64_t)n0 * p.ldw);
  // piece pi = wv + 8 i holds tile rows 8 pi + lane/8; chunk position lane%8 holds source chunk pos ^ key(row)
  const int prow = lane >> 3, key = (4 * (wv & 1) + (lane >> 4)) & 7, csrc = (lane & 7) ^ key;
  unsigned xvo[4], wvo[2 * NI];
#pragma unroll
  for (int i = 0; i < (2 * NI > 4 ? 2 * NI : 4); ++i) {
    const int row = 8 * (wv + 8 * i) + prow;
    if (i < 4) xvo[i] = (unsigned)(min(row, p.M - 1 - m0) * (int)p.ldx * 2 + csrc * 16);
    if (i < 2 * NI) wvo[i] = (unsigned)(min(row, p.N - 1 - n0) * (int)p.ldw * 2 + csrc * 16);
  }
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(smem_l);
  auto dma = [&](unsigned dst, unsigned voff, const u32x4& rs, unsigned soff) {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(dst), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
  };
  auto stage = [&](int kt, int buf) {
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)kt * (PG_BK * 2));
    const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF + wv * 1024));
#pragma unroll
    for (int i = 0; i < (2 * NI > 4 ? 2 * NI : 4); ++i) {
      if (i < 4) dma(d + i * 8192, xvo[i], xrs, soff);
      if (i < 2 * NI) dma(d + TILE + i * 8192, wvo[i], wrs, soff);
    }
  };

  f32x16 acc[NI][4];   // [latent-column tile][token tile]
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

  const int nk = p.K / PG_BK;
  const int fr = lane & 31, kb = lane >> 5;
  // fragment addresses: row * 128 + ((2 ks + kb) ^ key(row)) * 16; rows of a wave's tiles differ by multiples of 32,
  // so the key is the lane's own and the k-step only flips bits 5..6 of the chunk
  const int fkey = (fr >> 1) & 7;
  const unsigned xrd = (unsigned)((wm * 128 + fr) * 128), wrd = (unsigned)((wn * 32 * NI + fr) * 128);
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* xs = smem_l + (kt & 1) * BUF;
    const char* ws = xs + TILE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned co = (unsigned)(((2 * ks + kb) ^ fkey) * 16);
      h16x8 fx[4], fw[NI];
#pragma unroll
      for (int t = 0; t < 4; ++t) fx[t] = *reinterpret_cast<const h16x8*>(xs + xrd + t * 4096 + co);
#pragma unroll
      for (int t = 0; t < NI; ++t) fw[t] = *reinterpret_cast<const h16x8*>(ws + wrd + t * 4096 + co);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i], fx[jj], acc[i][jj], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (QB != 0) {
    constexpr float QMAX = (float)((1 << QB) - 1);
    float* mm = reinterpret_cast<float*>(smem_l);            // [wm 2][wn 4][128 tokens][min, max]: the tiles are consumed
    const int wpg = p.R / (32 * NI);                         // waves per latent group: 1, 2 or 4
    const int wn0 = wn / wpg * wpg;
    // round to fp16 once (the value the unfused path would have stored), row extrema of this wave's 32 NI columns
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float mx = -INFINITY, mn = INFINITY;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = (float)(h16)acc[i][jj][e];
          acc[i][jj][e] = v;
          mx = fmaxf(mx, v);
          mn = fminf(mn, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      mn = fminf(mn, __shfl_xor(mn, 32));
      if (kb == 0) {
        float* d = mm + ((wm * 4 + wn) * 128 + jj * 32 + fr) * 2;
        d[0] = mn;
        d[1] = mx;
      }
    }
    __syncthreads();
    const int nw0 = n0 + wn * 32 * NI;                       // first column of this wave
    const int g = nw0 / p.R, cg0 = nw0 - g * p.R;            // its group, its first column inside the group
    const float floor16 = (float)(h16)1e-5f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float mx = -INFINITY, mn = INFINITY;
      for (int k = 0; k < wpg; ++k) {
        const float* d = mm + ((wm * 4 + wn0 + k) * 128 + jj * 32 + fr) * 2;
        mn = fminf(mn, d[0]);
        mx = fmaxf(mx, d[1]);
      }
      // quant.py:29-38 (csrc/quant.hip quantize_row, asymmetric, clip 1): every op an fp32 op on fp16 values, rounded once
      float range = (float)(h16)(mx - mn);
      range = fmaxf(range, floor16);
      const float scale = (float)(h16)(range / QMAX);
      float zero = rintf((float)(h16)(-mn / scale));
      zero = fminf(fmaxf(zero, 0.f), QMAX);
      const int m = m0 + wm * 128 + jj * 32 + fr;
      const bool valid = m < p.M;
      if (valid && kb == 0 && wn == wn0) {
        h16* md = p.meta + (int64_t)g * p.sm_g + (int64_t)(p.row0 + m) * p.sm_l;
        md[0] = (h16)scale;
        md[1] = (h16)zero;
      }
      unsigned char* crow = p.codes + (int64_t)g * p.sc_g + (int64_t)(p.row0 + m) * p.sc_l;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        unsigned u[4], pu[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u[q] = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float qv = (float)(h16)(rintf((float)(h16)(acc[i][jj][4 * q + r] / scale)) + zero);   // quant.py:39
            qv = fminf(fmaxf(qv, 0.f), QMAX);
            u[q] |= (unsigned)qv << (QB * r);
          }
          pu[q] = (unsigned)__shfl_xor((int)u[q], 32);       // the partner lane's columns 8q + 4 .. 8q + 7
        }
        if (valid && kb == 0) {
          const int cg = cg0 + 32 * i;                       // 32 codes from column cg of the group's row
          if (QB == 4) {
            *reinterpret_cast<u32x4*>(crow + cg / 2) = u32x4{u[0] | (pu[0] << 16), u[1] | (pu[1] << 16), u[2] | (pu[2] << 16),
                                                             u[3] | (pu[3] << 16)};
          } else {
            const unsigned t0 = u[0] | (pu[0] << 12), t1 = u[1] | (pu[1] << 12), t2 = u[2] | (pu[2] << 12),
                           t3 = u[3] | (pu[3] << 12);        // 24 bits each
            unsigned* d = reinterpret_cast<unsigned*>(crow + cg * 3 / 8);
            d[0] = t0 | (t1 << 24);
            d[1] = (t1 >> 8) | (t2 << 16);
            d[2] = (t2 >> 16) | (t3 << 8);
          }
        }
      }
    }
    return;
  }
  // C^T tiles: lane column (lane & 31) = token, rows = latent columns (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a lane
  // holds 4 consecutive latent columns of one token (N % R == 0 and R % 4 == 0: never across a group boundary)
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wn * 32 * NI + i * 32 + 8 * q + 4 * kb;
      if (n >= p.N) continue;
      const int g = n / p.R;
      h16* col = p.out + (int64_t)g * p.so_g + (n - g * p.R) + (int64_t)p.row0 * p.so_l;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int m = m0 + wm * 128 + jj * 32 + fr;
        if (m >= p.M) continue;
        h16x4 v = {(h16)acc[i][jj][4 * q], (h16)acc[i][jj][4 * q + 1], (h16)acc[i][jj][4 * q + 2], (h16)acc[i][jj][4 * q + 3]};
        *reinterpret_cast<h16x4*>(col + (int64_t)m * p.so_l) = v;
      }
    }
}

}  // namespace

// tile width (NI) of the fused quantise epilogue for group rank R over N columns: whole groups per workgroup
// (32 NI | R, R | 128 NI, N % (128 NI) == 0); 0 = the shape stays on the unfused path
static int pgq_ni(int M, int N, int K, int R, int64_t ldx, int64_t ldw) {
  const bool large = M >= 512 && N >= 256 && K >= 512 && 256 * ldx * 2 < (int64_t(1) << 31) && 384 * ldw * 2 < (int64_t(1) << 31);
  if (!large || K % PG_BK != 0 || R <= 0 || N % R != 0) return 0;
  auto ok = [&](int ni) { return R % (32 * ni) == 0 && (128 * ni) % R == 0 && N % (128 * ni) == 0; };
  const int ntm = (M + PL_BM - 1) / PL_BM;
  if (ok(2) && (int64_t)ntm * (N / 256) >= palu_num_cus()) return 2;
  if (ok(1)) return 1;
  if (ok(2)) return 2;
  if (ok(3)) return 3;
  return 0;
}

extern "C" int palu_lowrank_project_gemm_q_supported(int M, int N, int K, int R, int bits) {
  if (!(bits == 3 || bits == 4)) return 0;
  return pgq_ni(M, N, K, R, K, K) != 0 ? 1 : 0;
}

// The projection with the quantise + pack epilogue: X [M, K] . VT^T [N, K] -> packed rows codes[(n / R)][row0 + m] (byte
// strides sc_g, sc_l; a row is R * bits / 8 bytes) and meta[(n / R)][row0 + m] = (scale, zero) fp16 (element strides sm_g,
// sm_l) -- asymmetric per-(token, group) rows, clip 1, group_size 0 (the reference defaults).  Bit-identical to
// palu_lowrank_project_gemm followed by palu_quantize_pack.  PALU_ERR_UNSUPPORTED for shapes the fused tile does not take
// (palu_lowrank_project_gemm_q_supported): run the two calls instead.
extern "C" int palu_lowrank_project_gemm_q(const void* x, int64_t ldx, const void* w, int64_t ldw, void* codes, int64_t sc_g,
                                           int64_t sc_l, void* meta, int64_t sm_g, int64_t sm_l, int M, int N, int K, int R,
                                           int row0, int bits, palu_stream_t stream) {
  PALU_REQUIRE(x && w && codes && meta && M >= 0 && N > 0 && K > 0 && R > 0 && row0 >= 0, PALU_ERR_ARG, "project_gemm_q: bad arguments");
  PALU_REQUIRE(bits == 3 || bits == 4, PALU_ERR_UNSUPPORTED, "project_gemm_q: bits must be 3 or 4");
  PALU_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, PALU_ERR_ARG,
               "project_gemm_q: X / VT rows must be 16-byte aligned");
  const int align = bits == 4 ? 16 : 4;
  PALU_REQUIRE(((uintptr_t)codes % align) == 0 && sc_g % align == 0 && sc_l % align == 0 && sc_l >= (int64_t)R * bits / 8,
               PALU_ERR_ARG, "project_gemm_q: packed rows must be %d-byte aligned", align);
  PALU_REQUIRE(((uintptr_t)meta & 3) == 0 && sm_g % 2 == 0 && sm_l % 2 == 0 && sm_l >= 2, PALU_ERR_ARG,
               "project_gemm_q: meta rows must be 4-byte aligned (scale, zero) pairs");
  if (M == 0) return PALU_OK;
  const int ni = pgq_ni(M, N, K, R, ldx, ldw);
  PALU_REQUIRE(ni != 0, PALU_ERR_UNSUPPORTED, "project_gemm_q: no fused tile for M=%d N=%d K=%d R=%d", M, N, K, R);
  PgParams p = {};
  p.x = (const h16*)x; p.ldx = ldx; p.w = (const h16*)w; p.ldw = ldw;
  p.M = M; p.N = N; p.K = K; p.R = R; p.row0 = row0;
  p.codes = (unsigned char*)codes; p.sc_g = sc_g; p.sc_l = sc_l;
  p.meta = (h16*)meta; p.sm_g = sm_g; p.sm_l = sm_l;
  const int ntm = (M + PL_BM - 1) / PL_BM;
  const int bn = 128 * ni, ntn = N / bn;
  const int smem = 2 * (PL_BM + bn) * PG_BK * 2;
  const dim3 grid(((ntm + 7) / 8) * 8 * ntn), block(PL_THREADS);
  hipStream_t s = (hipStream_t)stream;
#define PALU_PGQ(NIV, QBV)                                                                                                  \
  {                                                                                                                         \
    const void* fn = (const void*)project_gemm_256_kernel<NIV, QBV>;                                                        \
    PALU_REQUIRE(palu_func_max_lds(fn, smem) == 0, PALU_ERR_LAUNCH, "project_gemm_q: cannot reserve %d bytes of LDS", smem); \
    hipLaunchKernelGGL((project_gemm_256_kernel<NIV, QBV>), grid, block, smem, s, p, ntn, ntm);                             \
  }
  if (bits == 4) {
    if (ni == 1) PALU_PGQ(1, 4) else if (ni == 2) PALU_PGQ(2, 4) else PALU_PGQ(3, 4)
  } else {
    if (ni == 1) PALU_PGQ(1, 3) else if (ni == 2) PALU_PGQ(2, 3) else PALU_PGQ(3, 3)
  }
#undef PALU_PGQ
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

extern "C" int palu_lowrank_project_gemm(const void* x, int64_t ldx, const void* w, int64_t ldw, void* out, int64_t so_g,
                                         int64_t so_l, int M, int N, int K, int R, int row0, palu_stream_t stream) {
  PALU_REQUIRE(x && w && out && M >= 0 && N > 0 && K > 0 && R > 0 && row0 >= 0, PALU_ERR_ARG, "project_gemm: bad arguments");
  PALU_REQUIRE(K % PG_BK == 0, PALU_ERR_UNSUPPORTED, "project_gemm: K must be a multiple of 64 (got %d)", K);
  PALU_REQUIRE(N % R == 0 && R % 4 == 0, PALU_ERR_ARG, "project_gemm: N must be a multiple of the group rank R, R %% 4 == 0");
  PALU_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, PALU_ERR_ARG,
               "project_gemm: X / VT rows must be 16-byte aligned");
  PALU_REQUIRE(so_g % 4 == 0 && so_l % 4 == 0 && ((uintptr_t)out & 7) == 0, PALU_ERR_ARG,
               "project_gemm: output rows must be 8-byte aligned");
  if (M == 0) return PALU_OK;
  PgParams p;
  p.x = (const h16*)x; p.ldx = ldx; p.w = (const h16*)w; p.ldw = ldw;
  p.out = (h16*)out; p.so_g = so_g; p.so_l = so_l;
  p.M = M; p.N = N; p.K = K; p.R = R; p.row0 = row0;
  // large shapes: the 256x256 LDS-DMA kernel (tile offsets are 32-bit there: 256 rows of either operand < 2 GiB)
  const bool large = M >= 512 && N >= 256 && K >= 512 && 256 * ldx * 2 < (int64_t(1) << 31) && 256 * ldw * 2 < (int64_t(1) << 31);
  if (large) {
    // 256 x 256 tiles unless they leave CUs without a workgroup (8192 x 1024: 128 tiles), then 256 x 128
    // (measured: 8192 x 1024 857 vs 686 TFLOP/s; 8192 x 3072 -- 384 wide tiles -- 807 wide vs 732 narrow)
    const int ntm = (M + PL_BM - 1) / PL_BM;
    const int cus = palu_num_cus();
    const bool wide = (int64_t)ntm * ((N + 255) / 256) >= cus;
    const int bn = wide ? 256 : 128, ntn = (N + bn - 1) / bn;
    const int smem = 2 * (PL_BM + bn) * PG_BK * 2;
    const void* fn = wide ? (const void*)project_gemm_256_kernel<2> : (const void*)project_gemm_256_kernel<1>;
    PALU_REQUIRE(palu_func_max_lds(fn, smem) == 0, PALU_ERR_LAUNCH, "project_gemm: cannot reserve %d bytes of LDS", smem);
    const dim3 grid(((ntm + 7) / 8) * 8 * ntn), block(PL_THREADS);
    if (wide) hipLaunchKernelGGL(project_gemm_256_kernel<2>, grid, block, smem, (hipStream_t)stream, p, ntn, ntm);
    else hipLaunchKernelGGL(project_gemm_256_kernel<1>, grid, block, smem, (hipStream_t)stream, p, ntn, ntm);
  } else {
    dim3 grid((M + PG_BM - 1) / PG_BM, (N + PG_BN - 1) / PG_BN);
    hipLaunchKernelGGL(project_gemm_kernel, grid, dim3(PG_THREADS), 0, (hipStream_t)stream, p);
  }
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}
