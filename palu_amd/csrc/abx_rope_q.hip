// abx on QUANTISED latent keys: same kernel as abx_rope.hip with the tile staging replaced by
// packed-code loads + in-register dequantisation (abx_rope_kernel.h, QBITS = 3/4).  The dequantised
// fp16 values are bit-identical to the reference's fake-quant output (quant.py:39), so scores equal
// those of the fp16 kernel run on quantize_tensor(x).  Flops are unchanged (the kernel is MFMA-bound);
// only the HBM bytes shrink (32.5 MB at C3 instead of 139.5 MB).
#include "abx_rope_kernel.h"

namespace {
template <int NKS, int NMB, int QBITS>
int launch_abx_q(const AbxParams& p, int nwg, hipStream_t stream) {
  if ((int64_t)p.pos0 + p.L > 262144) {    // positions beyond 2^18: second-order angle correction (ORDER2)
    return launch_kernel(abx_rope_kernel<NKS, NMB, true, false, QBITS, true>, abx_smem_fast(NKS), p, nwg, stream);
  }
  return launch_kernel(abx_rope_kernel<NKS, NMB, true, false, QBITS>, abx_smem_fast(NKS), p, nwg, stream);
}
// any other rank with whole-row (scale, zero): 128-column WINDOWS through the fast 128-column kernel (masked staging of
// the missing columns; with more than one window the partial scores are accumulated in an fp32 scratch, ACC = 1 / 2)
template <int NMB, int QBITS, int ACC>
int launch_abx_q_window(const AbxParams& p, int nwg, hipStream_t stream) {
  if ((int64_t)p.pos0 + p.L > 262144)
    return launch_kernel(abx_rope_kernel<8, NMB, true, false, QBITS, true, false, ACC>, abx_smem_fast(8), p, nwg, stream);
  return launch_kernel(abx_rope_kernel<8, NMB, true, false, QBITS, false, false, ACC>, abx_smem_fast(8), p, nwg, stream);
}
__global__ void abx_q_round_kernel(const float* __restrict__ acc, int64_t acc_ld, h16* __restrict__ out, int64_t so_h, int L) {
  const int h = blockIdx.y;
  for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x)
    out[(int64_t)h * so_h + l] = (h16)acc[(int64_t)h * acc_ld + l];
}

// group-wise (scale, zero) pairs, or no scratch for a rank above 128: the chunked kernel (128-column chunks, zero-padded
// B, fragments re-read per chunk) with in-register dequantisation of every slot
template <int NMB, int QBITS>
int launch_abx_q_generic(const AbxParams& p, int nwg, hipStream_t stream) {
  return launch_kernel(abx_rope_generic_kernel<8, NMB, true, QBITS>, abx_smem_bytes(8, 2), p, nwg, stream);
}
}  // namespace

static int abx_rope_q_impl(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, const void* codes,
                           int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l, void* out,
                           int64_t so_h, int H, int G, int L, int R, int D, int bits, int group_size, const float* inv_freq,
                           int pos0, void* scratch, palu_stream_t stream);

extern "C" int palu_abx_rope_q(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, const void* codes,
                               int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l, void* out,
                               int64_t so_h, int H, int G, int L, int R, int D, int bits, const float* inv_freq,
                               int pos0, palu_stream_t stream) {
  return abx_rope_q_impl(a, sa_h, sa_d, bfrag, codes, sc_g, sc_l, meta, sm_g, sm_l, out, so_h, H, G, L, R, D, bits, 0, inv_freq,
                         pos0, nullptr, stream);
}

// the same on rows quantised in column groups (quantize_tensor(..., group_size > 0), quant.py:11-13, --lt_group_size):
// meta [G, L, R / group_size, 2]; the packed codes are laid out exactly as for whole-row quantisation
// scratch: optional fp32 array of palu_abx_scratch_bytes(H, G, L, R) bytes -- with it a whole-row-quantised rank above 128
// runs as passes of the fast kernel instead of the chunked one
extern "C" int palu_abx_rope_qg(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, const void* codes,
                                int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l, void* out,
                                int64_t so_h, int H, int G, int L, int R, int D, int bits, int group_size,
                                const float* inv_freq, int pos0, void* scratch, palu_stream_t stream) {
  if (group_size == R) group_size = 0;
  PALU_REQUIRE(group_size == 0 || (group_size > 0 && R % group_size == 0 && group_size % 8 == 0), PALU_ERR_UNSUPPORTED,
               "abx_qg: group_size %d must divide R %d and be a multiple of 8", group_size, R);
  PALU_REQUIRE(group_size == 0 || sm_l >= 2 * (R / group_size), PALU_ERR_ARG,
               "abx_qg: meta rows hold R / group_size (scale, zero) pairs");
  return abx_rope_q_impl(a, sa_h, sa_d, bfrag, codes, sc_g, sc_l, meta, sm_g, sm_l, out, so_h, H, G, L, R, D, bits, group_size,
                         inv_freq, pos0, scratch, stream);
}

static int abx_rope_q_impl(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, const void* codes,
                           int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l, void* out,
                           int64_t so_h, int H, int G, int L, int R, int D, int bits, int group_size, const float* inv_freq,
                           int pos0, void* scratch, palu_stream_t stream) {
  AbxPlan pl;
  PALU_REQUIRE(abx_plan(H, G, R, &pl), PALU_ERR_ARG, "abx_q: bad shape H=%d G=%d R=%d", H, G, R);
  PALU_REQUIRE(D == HEAD_DIM, PALU_ERR_UNSUPPORTED, "abx_q: head_dim must be 128 (got %d)", D);
  PALU_REQUIRE((bits == 4 && R % 8 == 0) || (bits == 3 && R % 32 == 0), PALU_ERR_UNSUPPORTED,
               "abx_q: bits must be 3 (R %% 32 == 0) or 4 (R %% 8 == 0); got (%d, %d)", bits, R);
  // fast path (tile staging of whole quarter rows): (4, 32|64|128), (3, 128); everything else -- the ranks the rank
  // search emits (96, 160, 224, 256, ...) and 3-bit at 32 / 64 -- goes to the two-band kernel when its rules are met (below), else to
  // the windowed / chunked one-band kernels
  const bool fast = group_size == 0 && ((bits == 4 && (R == 32 || R == 64 || R == 128)) || (bits == 3 && R == 128));
  PALU_REQUIRE(L >= 0, PALU_ERR_ARG, "abx_q: negative L");
  if (L == 0) return PALU_OK;
  PALU_REQUIRE(a && bfrag && codes && meta && out && inv_freq, PALU_ERR_ARG, "abx_q: null pointer");
  PALU_REQUIRE(((uintptr_t)codes & 3) == 0 && sc_g % 4 == 0 && sc_l % 4 == 0 && sc_l >= (int64_t)R * bits / 8,
               PALU_ERR_ARG, "abx_q: packed rows must be 4-byte aligned");
  PALU_REQUIRE(((uintptr_t)meta & 3) == 0 && sm_g % 2 == 0 && sm_l % 2 == 0 && sm_l >= 2, PALU_ERR_ARG,
               "abx_q: meta rows must be 4-byte aligned (scale, zero) pairs");
  PALU_REQUIRE(((uintptr_t)bfrag & 15) == 0, PALU_ERR_ARG, "abx_q: bfrag must be 16-byte aligned");
  PALU_REQUIRE((int64_t)pos0 + L < (1 << 24), PALU_ERR_UNSUPPORTED, "abx_q: positions must stay below 2^24");
  const int64_t ob = ((int64_t)(H - 1) * so_h + L) * 2;
  PALU_REQUIRE(ob > 0 && ob < 0xFFFFFFF0ll, PALU_ERR_UNSUPPORTED, "abx_q: out extent must be < 4 GiB");

  AbxParams p = {};
  p.a = (const h16*)a; p.sa_h = sa_h; p.sa_d = sa_d;
  p.bfrag = (const u32x4*)bfrag;
  p.xq = (const unsigned char*)codes; p.sq_g = sc_g; p.sq_l = sc_l;
  p.xmeta = (const h16*)meta; p.sm_g = sm_g; p.sm_l = sm_l;
  p.out = (h16*)out; p.so_h = so_h; p.out_bytes = (unsigned)ob;
  p.inv_freq = inv_freq;
  const int nks_frag = pl.nks_tot;       // how palu_abx_prepare_b laid the fragments out for (H, G, R)
  // windows of 128 columns through the fast kernel: whole-row metas, fragments in the chunked layout (8 k-steps per
  // window: every rank that is not 32 / 64 / 128), and an fp32 scratch when there is more than one window
  const int nwin = (R + 127) / 128;
  if (!fast && group_size == 0 && pl.chunked && (nwin == 1 || (scratch && ((uintptr_t)scratch & 15) == 0))) {
    hipStream_t sw = (hipStream_t)stream;
    const int nwg = abx_fill_params(p, pl, H, G, L, R, pos0);
    const int64_t acc_ld = ((int64_t)L + 7) & ~(int64_t)7;
    if (palu_abx2_frag_bytes(H, G, R)) {              // (rank 96 or above 128)
      // 4 heads per group: the column windows through the two-band kernel (abx_rope2.hip)
      p.bfrag2 = (const u32x4*)((const char*)bfrag + (size_t)G * pl.hb * 8 * pl.nmb * pl.nks_tot * 64 * sizeof(u32x4));
      const int rc2 = palu_abx2_try_launch_windows(&p, nwg, bits, scratch, acc_ld, sw);
      if (rc2 != PALU_ABX2_SKIP) {
        if (rc2) return rc2;
        return PALU_OK;
      }
    }
    for (int kc = 0; kc < nwin; ++kc) {
      AbxParams pk = p;
      pk.xq = (const unsigned char*)codes + (size_t)kc * 128 * bits / 8;
      pk.ncols = R - 128 * kc < 128 ? R - 128 * kc : 128;
      pk.nks_frag = nks_frag;
      pk.ks0 = 8 * kc;
      pk.acc = (float*)scratch;
      pk.acc_ld = acc_ld;
      int rc;
#define PALU_ABXQ_WIN(ACCV)                                                                                   \
  (bits == 3 ? (pl.nmb == 2 ? launch_abx_q_window<2, 3, ACCV>(pk, nwg, sw) : launch_abx_q_window<1, 3, ACCV>(pk, nwg, sw)) \
             : (pl.nmb == 2 ? launch_abx_q_window<2, 4, ACCV>(pk, nwg, sw) : launch_abx_q_window<1, 4, ACCV>(pk, nwg, sw)))
      if (nwin == 1) rc = PALU_ABXQ_WIN(0);
      else if (kc == 0) rc = PALU_ABXQ_WIN(1);
      else rc = PALU_ABXQ_WIN(2);
#undef PALU_ABXQ_WIN
      if (rc) return rc;
    }
    if (nwin > 1) {
      int bx = (L + 255) / 256;
      if (bx > 256) bx = 256;
      hipLaunchKernelGGL(abx_q_round_kernel, dim3(bx, H), dim3(256), 0, sw, (const float*)scratch, acc_ld, (h16*)out, so_h, L);
      PALU_LAUNCH_CHECK();
    }
    return PALU_OK;
  }
  if (group_size > 0 && group_size % 32 == 0 && palu_abx2_frag_bytes(H, G, R)) {
    // rows quantised in column groups at 4 heads per group: the two-band kernel picks the (scale, zero) pair per lane piece
    // (abx_rope2_kernel.h); single launch at R in {32, 64, 128}, column windows otherwise (they need the fp32 scratch)
    const int nwg2 = abx_fill_params(p, pl, H, G, L, R, pos0);
    p.nks_frag = nks_frag;
    p.qgroup = group_size;
    p.qcol0 = 0;
    p.bfrag2 = (const u32x4*)((const char*)bfrag + (size_t)G * pl.hb * 8 * pl.nmb * pl.nks_tot * 64 * sizeof(u32x4));
    int rc2 = PALU_ABX2_SKIP;
    if (R == 32 || R == 64 || R == 128) rc2 = palu_abx2_try_launch(&p, nwg2, bits, (hipStream_t)stream);
    else if (R == 96 || (scratch && ((uintptr_t)scratch & 15) == 0))
      rc2 = palu_abx2_try_launch_windows(&p, nwg2, bits, scratch, ((int64_t)L + 7) & ~(int64_t)7, (hipStream_t)stream);
    if (rc2 != PALU_ABX2_SKIP) return rc2;
    p.qgroup = 0;
    p.bfrag2 = nullptr;
  }
  if (!fast && group_size == 0 && bits == 3 && (R == 32 || R == 64) && !pl.chunked && palu_abx2_frag_bytes(H, G, R)) {
    // 3-bit rows of 32 / 64 codes at 4 heads per group: the two-band kernel stages them 12 bytes per lane (abx_rope2_kernel.h);
    // other group sizes and positions without a coefficient table stay on the chunked kernel below
    const int nwg2 = abx_fill_params(p, pl, H, G, L, R, pos0);
    p.nks_frag = nks_frag;
    p.qgroup = 0;
    p.bfrag2 = (const u32x4*)((const char*)bfrag + (size_t)G * pl.hb * 8 * pl.nmb * pl.nks_tot * 64 * sizeof(u32x4));
    const int rc2 = palu_abx2_try_launch(&p, nwg2, 3, (hipStream_t)stream);
    if (rc2 != PALU_ABX2_SKIP) return rc2;
  }
  if (!fast) {
    // plan the launch as the chunked fp16 kernel does: 128-column chunks whatever R is
    pl.chunked = true;
    pl.nkc = (R + 127) / 128;
    pl.nks_tot = 8 * pl.nkc;
  }
  const int nwg = abx_fill_params(p, pl, H, G, L, R, pos0);
  p.nks_frag = nks_frag;
  p.qgroup = group_size;
  hipStream_t s = (hipStream_t)stream;
  if (!fast) {
    if (bits == 3) return pl.nmb == 2 ? launch_abx_q_generic<2, 3>(p, nwg, s) : launch_abx_q_generic<1, 3>(p, nwg, s);
    return pl.nmb == 2 ? launch_abx_q_generic<2, 4>(p, nwg, s) : launch_abx_q_generic<1, 4>(p, nwg, s);
  }
  if (palu_abx2_frag_bytes(H, G, R)) {
    // gs = 4 at a fast rank: the two-band kernel when a coefficient table covers the positions (abx_rope2.hip)
    p.bfrag2 = (const u32x4*)((const char*)bfrag + (size_t)G * pl.hb * 8 * pl.nmb * pl.nks_tot * 64 * sizeof(u32x4));
    const int rc = palu_abx2_try_launch(&p, nwg, bits, s);
    if (rc != PALU_ABX2_SKIP) return rc;
  }
  if (bits == 3) return pl.nmb == 2 ? launch_abx_q<8, 2, 3>(p, nwg, s) : launch_abx_q<8, 1, 3>(p, nwg, s);
  switch (R) {
    case 32: return pl.nmb == 2 ? launch_abx_q<2, 2, 4>(p, nwg, s) : launch_abx_q<2, 1, 4>(p, nwg, s);
    case 64: return pl.nmb == 2 ? launch_abx_q<4, 2, 4>(p, nwg, s) : launch_abx_q<4, 1, 4>(p, nwg, s);
    default: return pl.nmb == 2 ? launch_abx_q<8, 2, 4>(p, nwg, s) : launch_abx_q<8, 1, 4>(p, nwg, s);
  }
}
