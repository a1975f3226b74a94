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
// any other rank: the chunked kernel (128-column chunks, zero-padded B) with in-register dequantisation of every slot
template <int NMB, int QBITS>
int launch_abx_q_generic(const AbxParams& p, int nwg, hipStream_t stream) {
  return launch_kernel(abx_rope_generic_kernel<8, NMB, true, QBITS>, abx_smem_bytes(8, 2), p, nwg, stream);
}
}  // namespace

static int abx_rope_q_impl(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, const void* codes,
                           int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l, void* out,
                           int64_t so_h, int H, int G, int L, int R, int D, int bits, int group_size, const float* inv_freq,
                           int pos0, palu_stream_t stream);

extern "C" int palu_abx_rope_q(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, const void* codes,
                               int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l, void* out,
                               int64_t so_h, int H, int G, int L, int R, int D, int bits, const float* inv_freq,
                               int pos0, palu_stream_t stream) {
  return abx_rope_q_impl(a, sa_h, sa_d, bfrag, codes, sc_g, sc_l, meta, sm_g, sm_l, out, so_h, H, G, L, R, D, bits, 0, inv_freq,
                         pos0, stream);
}

// the same on rows quantised in column groups (quantize_tensor(..., group_size > 0), quant.py:11-13, --lt_group_size):
// meta [G, L, R / group_size, 2]; the packed codes are laid out exactly as for whole-row quantisation
extern "C" int palu_abx_rope_qg(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, const void* codes,
                                int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l, void* out,
                                int64_t so_h, int H, int G, int L, int R, int D, int bits, int group_size,
                                const float* inv_freq, int pos0, palu_stream_t stream) {
  if (group_size == R) group_size = 0;
  PALU_REQUIRE(group_size == 0 || (group_size > 0 && R % group_size == 0 && group_size % 8 == 0), PALU_ERR_UNSUPPORTED,
               "abx_qg: group_size %d must divide R %d and be a multiple of 8", group_size, R);
  PALU_REQUIRE(group_size == 0 || sm_l >= 2 * (R / group_size), PALU_ERR_ARG,
               "abx_qg: meta rows hold R / group_size (scale, zero) pairs");
  return abx_rope_q_impl(a, sa_h, sa_d, bfrag, codes, sc_g, sc_l, meta, sm_g, sm_l, out, so_h, H, G, L, R, D, bits, group_size,
                         inv_freq, pos0, stream);
}

static int abx_rope_q_impl(const void* a, int64_t sa_h, int64_t sa_d, const void* bfrag, const void* codes,
                           int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g, int64_t sm_l, void* out,
                           int64_t so_h, int H, int G, int L, int R, int D, int bits, int group_size, const float* inv_freq,
                           int pos0, palu_stream_t stream) {
  AbxPlan pl;
  PALU_REQUIRE(abx_plan(H, G, R, &pl), PALU_ERR_ARG, "abx_q: bad shape H=%d G=%d R=%d", H, G, R);
  PALU_REQUIRE(D == HEAD_DIM, PALU_ERR_UNSUPPORTED, "abx_q: head_dim must be 128 (got %d)", D);
  PALU_REQUIRE((bits == 4 && R % 8 == 0) || (bits == 3 && R % 32 == 0), PALU_ERR_UNSUPPORTED,
               "abx_q: bits must be 3 (R %% 32 == 0) or 4 (R %% 8 == 0); got (%d, %d)", bits, R);
  // fast path (tile staging of whole quarter rows): (4, 32|64|128), (3, 128); everything else -- the ranks the rank
  // search emits (96, 160, 224, 256, ...) and 3-bit at 32 / 64 -- runs the chunked kernel
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
  if (bits == 3) return pl.nmb == 2 ? launch_abx_q<8, 2, 3>(p, nwg, s) : launch_abx_q<8, 1, 3>(p, nwg, s);
  switch (R) {
    case 32: return pl.nmb == 2 ? launch_abx_q<2, 2, 4>(p, nwg, s) : launch_abx_q<2, 1, 4>(p, nwg, s);
    case 64: return pl.nmb == 2 ? launch_abx_q<4, 2, 4>(p, nwg, s) : launch_abx_q<4, 1, 4>(p, nwg, s);
    default: return pl.nmb == 2 ? launch_abx_q<8, 2, 4>(p, nwg, s) : launch_abx_q<8, 1, 4>(p, nwg, s);
  }
}
