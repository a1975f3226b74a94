// 3/4-bit latent quantisation: quantise+pack (cache append), unpack+dequantise, raw pack/unpack.
//
// The reference only has FAKE quantisation (palu/model/modules/quant.py:5-41 returns the
// dequantised tensor; README.md:24 lists the packed kernel as TODO), so the bit layout is this
// build's (DESIGN.md "packed latent format"):
//   codes : per (group, position) row of R codes, little-endian bit stream, code j at bits
//           [j*b, (j+1)*b) of the row  (b=4: two codes per byte; b=3: 32 codes per 3 uint32)
//   meta  : per row one (scale, zero) pair of fp16  (asymmetric, whole-row groups: the reference
//           defaults lt_group_size=0, lt_sym=False, lt_clip_ratio=1.0 of utils.py:103-108)
// The integer codes and the dequantised values are BIT-EXACT with quantize_tensor's fp16
// arithmetic: every op below is the fp32 op on fp16 operands rounded once to fp16, which is what
// torch's CPU half kernels compute (and is exact-equivalent for +,-,*,/ since 24 >= 2*11+2).
#include "palu_common.h"

namespace {

static __device__ __forceinline__ float r16(float x) { return (float)(h16)x; }  // round to fp16, keep as float

// fp16 value * fp32 scalar the way torch does it: the product is rounded to fp32 FIRST, then to fp16.  Left to the
// compiler, (h16)(a * b) becomes one v_fma_mixlo_f16 (a single rounding of the exact product), which differs whenever
// the fp32-rounded product lands on an fp16 tie (seen on golden G5: 3.115234375 * 0.9f).
static __device__ __forceinline__ float mul_r32_r16(float a, float b) {
  float p = a * b;
  asm volatile("" : "+v"(p));
  return r16(p);
}

struct QpTensor {
  const h16* x;  int64_t sx_g, sx_l;
  unsigned char* codes; int64_t sc_g, sc_l;
  h16* meta;     int64_t sm_g, sm_l;
  h16* deq;      int64_t sd_g, sd_l;
  int R;
  float clip;    // lt_clip_ratio (quant.py:18-36); 1.0 = off
};

// one wave quantises + packs one (group, row) of R codes.
// SYM (quant.py:18-28): scale = clamp(amax|w|, 1e-5)[*clip] / (2^(b-1)-1), signed codes in [-2^(b-1), 2^(b-1)-1], base 0.
// The packed format stores unsigned codes, so a symmetric row is stored in offset binary: code + 2^(b-1) with
// zero = 2^(b-1) in the meta pair -- (stored - zero) * scale is the reference's code * scale bit for bit, and the
// decode kernels need no symmetric variant.
template <int BITS, bool SYM>
static __device__ __forceinline__ void quantize_row(const QpTensor& t, int g, int l, int lane) {
  const int R = t.R;
  const h16* row = t.x + g * t.sx_g + l * t.sx_l;
  constexpr float QMAX = SYM ? (float)((1 << (BITS - 1)) - 1) : (float)((1 << BITS) - 1);
  constexpr float QMIN = SYM ? -(float)(1 << (BITS - 1)) : 0.f;
  constexpr float OFFS = SYM ? (float)(1 << (BITS - 1)) : 0.f;     // stored code = code + OFFS
  float mx = -INFINITY, mn = INFINITY;
  for (int j = lane; j < R; j += 64) {
    float v = (float)row[j];
    mx = fmaxf(mx, v);
    mn = fminf(mn, v);
  }
  mx = wave_max(mx);
  mn = -wave_max(-mn);
  const float floor16 = (float)(h16)1e-5f;   // the clamp constant is cast to fp16 (a subnormal)
  float scale, zero;
  if (SYM) {
    // quant.py:18-24: w_max = amax|w|.clamp(min=1e-5) [* clip]; scales = w_max / q_max; base = 0
    float top = fmaxf(fmaxf(mx, -mn), floor16);
    // Half tensor * Python scalar: fp32 product of the fp16 value and (float)clip, rounded once (torch semantics)
    if (t.clip < 1.0f) top = mul_r32_r16(top, t.clip);
    scale = r16(top / QMAX);
    zero = 0.f;
  } else {
    // quant.py:29-38: [max, min *= clip]; scales = (max-min).clamp(min=1e-5)/q_max ; base = round(-min/scales).clamp(0, q_max)
    if (t.clip < 1.0f) {
      mx = mul_r32_r16(mx, t.clip);
      mn = mul_r32_r16(mn, t.clip);
    }
    float range = r16(mx - mn);
    range = fmaxf(range, floor16);
    scale = r16(range / QMAX);
    zero = rintf(r16(-mn / scale));
    zero = fminf(fmaxf(zero, 0.f), QMAX);
  }
  if (lane == 0) {
    h16* m = t.meta + g * t.sm_g + l * t.sm_l;
    m[0] = (h16)scale;
    m[1] = (h16)(zero + OFFS);
  }
  unsigned char* crow = t.codes + g * t.sc_g + l * t.sc_l;
  h16* drow = t.deq ? t.deq + g * t.sd_g + l * t.sd_l : nullptr;
  // a lane packs 8 consecutive codes (one 3- or 4-byte unit) per step
  for (int u = lane; u < (R >> 3); u += 64) {
    unsigned bits = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float w = (float)row[8 * u + e];
      // quant.py:39: clamp(round(w/scales) + base, q_min, q_max)
      float q = r16(rintf(r16(w / scale)) + zero);
      q = fminf(fmaxf(q, QMIN), QMAX);
      bits |= (unsigned)(q + OFFS) << (BITS * e);
      if (drow) drow[8 * u + e] = (h16)(r16(q - zero) * scale);   // (q - base) * scales, fp16
    }
    unsigned char* d = crow + u * BITS;
    if (BITS == 4) {
      *reinterpret_cast<unsigned*>(d) = bits;
    } else {
      d[0] = (unsigned char)bits;
      d[1] = (unsigned char)(bits >> 8);
      d[2] = (unsigned char)(bits >> 16);
    }
  }
}

// waves [0, G*nrows) -> tensor a, the next G*nrows_b -> tensor b (the K and V latent rows of one decode step)
template <int BITS, bool SYM>
__global__ __launch_bounds__(256) void quantize_pack_kernel(QpTensor a, QpTensor b, int G, int nrows, int nrows_b) {
  const int lane = threadIdx.x & 63;
  int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t na = (int64_t)G * nrows;
  if (wid < na) {
    quantize_row<BITS, SYM>(a, (int)(wid / nrows), (int)(wid % nrows), lane);
  } else {
    wid -= na;
    if (wid < (int64_t)G * nrows_b) quantize_row<BITS, SYM>(b, (int)(wid / nrows_b), (int)(wid % nrows_b), lane);
  }
}

template <int BITS>
__global__ __launch_bounds__(256) void unpack_dequant_kernel(const unsigned char* __restrict__ codes, int64_t sc_g,
                                                             int64_t sc_l, const h16* __restrict__ meta, int64_t sm_g,
                                                             int64_t sm_l, h16* __restrict__ out, int64_t so_g,
                                                             int64_t so_l, int G, int nrows, int R) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per 8 codes
  const int upr = R >> 3;
  if (idx >= (int64_t)G * nrows * upr) return;
  const int u = (int)(idx % upr);
  const int64_t rl = idx / upr;
  const int g = (int)(rl / nrows), l = (int)(rl % nrows);
  const unsigned char* d = codes + g * sc_g + l * sc_l + u * BITS;
  unsigned bits = (BITS == 4) ? *reinterpret_cast<const unsigned*>(d)
                              : ((unsigned)d[0] | ((unsigned)d[1] << 8) | ((unsigned)d[2] << 16));
  const h16* m = meta + g * sm_g + l * sm_l;
  const float scale = (float)m[0], zero = (float)m[1];
  h16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float q = (float)((bits >> (BITS * e)) & ((1u << BITS) - 1));
    o[e] = (h16)(r16(q - zero) * scale);
  }
  *reinterpret_cast<h16x8*>(out + g * so_g + l * so_l + 8 * u) = o;
}

template <int BITS>
__global__ void pack_codes_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int64_t nunits) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= nunits) return;
  unsigned bits = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) bits |= ((unsigned)in[8 * u + e] & ((1u << BITS) - 1)) << (BITS * e);
  unsigned char* d = out + u * BITS;
#pragma unroll
  for (int k = 0; k < BITS; ++k) d[k] = (unsigned char)(bits >> (8 * k));
}

template <int BITS>
__global__ void unpack_codes_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int64_t nunits) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= nunits) return;
  const unsigned char* d = in + u * BITS;
  unsigned bits = 0;
#pragma unroll
  for (int k = 0; k < BITS; ++k) bits |= (unsigned)d[k] << (8 * k);
#pragma unroll
  for (int e = 0; e < 8; ++e) out[8 * u + e] = (unsigned char)((bits >> (BITS * e)) & ((1u << BITS) - 1));
}

bool quant_shape_ok(int bits, int R) {
  if (bits == 4) return R > 0 && R % 8 == 0;
  if (bits == 3) return R > 0 && R % 32 == 0;
  return false;
}

}  // namespace

extern "C" size_t palu_packed_row_bytes(int R, int bits) { return quant_shape_ok(bits, R) ? (size_t)R * bits / 8 : 0; }

namespace {
int launch_quantize(const QpTensor& ta, const QpTensor& tb, int G, int nrows, int nrows_b, int bits, hipStream_t s,
                    bool sym = false) {
  const int64_t waves = (int64_t)G * (nrows + nrows_b);
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  if (bits == 4) {
    if (sym) hipLaunchKernelGGL((quantize_pack_kernel<4, true>), grid, block, 0, s, ta, tb, G, nrows, nrows_b);
    else hipLaunchKernelGGL((quantize_pack_kernel<4, false>), grid, block, 0, s, ta, tb, G, nrows, nrows_b);
  } else {
    if (sym) hipLaunchKernelGGL((quantize_pack_kernel<3, true>), grid, block, 0, s, ta, tb, G, nrows, nrows_b);
    else hipLaunchKernelGGL((quantize_pack_kernel<3, false>), grid, block, 0, s, ta, tb, G, nrows, nrows_b);
  }
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}
}  // namespace

extern "C" int palu_quantize_pack(const void* x, int64_t sx_g, int64_t sx_l, void* codes, int64_t sc_g, int64_t sc_l,
                                  void* meta, int64_t sm_g, int64_t sm_l, void* dequant, int64_t sd_g, int64_t sd_l,
                                  int G, int nrows, int R, int bits, palu_stream_t stream) {
  return palu_quantize_pack_ex(x, sx_g, sx_l, codes, sc_g, sc_l, meta, sm_g, sm_l, dequant, sd_g, sd_l, G, nrows, R, bits,
                               0, 1.0f, stream);
}

extern "C" int palu_quantize_pack_ex(const void* x, int64_t sx_g, int64_t sx_l, void* codes, int64_t sc_g, int64_t sc_l,
                                     void* meta, int64_t sm_g, int64_t sm_l, void* dequant, int64_t sd_g, int64_t sd_l,
                                     int G, int nrows, int R, int bits, int sym, float clip_ratio,
                                     palu_stream_t stream) {
  PALU_REQUIRE(x && codes && meta && G > 0 && nrows >= 0, PALU_ERR_ARG, "quantize_pack: bad arguments");
  PALU_REQUIRE(clip_ratio > 0.f && clip_ratio <= 1.0f, PALU_ERR_ARG, "quantize_pack: clip_ratio must be in (0, 1]");
  PALU_REQUIRE(quant_shape_ok(bits, R), PALU_ERR_UNSUPPORTED,
               "quantize_pack: bits must be 3 (R %% 32 == 0) or 4 (R %% 8 == 0), got bits=%d R=%d", bits, R);
  PALU_REQUIRE(bits != 4 || (sc_g % 4 == 0 && sc_l % 4 == 0 && ((uintptr_t)codes & 3) == 0), PALU_ERR_ARG,
               "quantize_pack: 4-bit rows must be 4-byte aligned");
  PALU_REQUIRE(sm_l >= 2 || nrows <= 1, PALU_ERR_ARG, "quantize_pack: meta rows hold (scale, zero)");
  if (nrows == 0) return PALU_OK;
  QpTensor t = {(const h16*)x, sx_g, sx_l, (unsigned char*)codes, sc_g, sc_l, (h16*)meta, sm_g, sm_l,
                (h16*)dequant, sd_g, sd_l, R, clip_ratio};
  return launch_quantize(t, t, G, nrows, 0, bits, (hipStream_t)stream, sym != 0);
}

// internal (decode_step.hip): quantise+pack the new K and V latent rows of one step in ONE launch
int palu_quantize_pack_kv(const void* k, int64_t sk_g, void* k_codes, int64_t skc_g, void* k_meta, int64_t skm_g, int Rk,
                          const void* v, int64_t sv_g, void* v_codes, int64_t svc_g, void* v_meta, int64_t svm_g, int Rv,
                          int G, int bits, palu_stream_t stream) {
  PALU_REQUIRE(quant_shape_ok(bits, Rk) && quant_shape_ok(bits, Rv), PALU_ERR_UNSUPPORTED,
               "quantize_pack_kv: unsupported bits=%d Rk=%d Rv=%d", bits, Rk, Rv);
  QpTensor tk = {(const h16*)k, sk_g, 0, (unsigned char*)k_codes, skc_g, 0, (h16*)k_meta, skm_g, 0, nullptr, 0, 0, Rk, 1.0f};
  QpTensor tv = {(const h16*)v, sv_g, 0, (unsigned char*)v_codes, svc_g, 0, (h16*)v_meta, svm_g, 0, nullptr, 0, 0, Rv, 1.0f};
  return launch_quantize(tk, tv, G, 1, 1, bits, (hipStream_t)stream);
}

extern "C" int palu_unpack_dequant(const void* codes, int64_t sc_g, int64_t sc_l, const void* meta, int64_t sm_g,
                                   int64_t sm_l, void* out, int64_t so_g, int64_t so_l, int G, int nrows, int R, int bits,
                                   palu_stream_t stream) {
  PALU_REQUIRE(codes && meta && out && G > 0 && nrows >= 0, PALU_ERR_ARG, "unpack_dequant: bad arguments");
  PALU_REQUIRE(quant_shape_ok(bits, R), PALU_ERR_UNSUPPORTED, "unpack_dequant: unsupported bits=%d R=%d", bits, R);
  PALU_REQUIRE(so_g % 8 == 0 && so_l % 8 == 0 && ((uintptr_t)out & 15) == 0, PALU_ERR_ARG,
               "unpack_dequant: output rows must be 16-byte aligned");
  PALU_REQUIRE(bits != 4 || (sc_g % 4 == 0 && sc_l % 4 == 0 && ((uintptr_t)codes & 3) == 0), PALU_ERR_ARG,
               "unpack_dequant: 4-bit rows must be 4-byte aligned");
  if (nrows == 0) return PALU_OK;
  const int64_t n = (int64_t)G * nrows * (R / 8);
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (bits == 4)
    hipLaunchKernelGGL(unpack_dequant_kernel<4>, grid, block, 0, s, (const unsigned char*)codes, sc_g, sc_l,
                       (const h16*)meta, sm_g, sm_l, (h16*)out, so_g, so_l, G, nrows, R);
  else
    hipLaunchKernelGGL(unpack_dequant_kernel<3>, grid, block, 0, s, (const unsigned char*)codes, sc_g, sc_l,
                       (const h16*)meta, sm_g, sm_l, (h16*)out, so_g, so_l, G, nrows, R);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

extern "C" int palu_pack_codes(const void* codes_u8, void* packed, int64_t ncodes, int bits, palu_stream_t stream) {
  PALU_REQUIRE(codes_u8 && packed && ncodes >= 0 && ncodes % 8 == 0 && (bits == 3 || bits == 4), PALU_ERR_ARG,
               "pack_codes: need bits in {3,4} and a multiple of 8 codes");
  if (ncodes == 0) return PALU_OK;
  const int64_t nu = ncodes / 8;
  dim3 grid((unsigned)((nu + 255) / 256)), block(256);
  if (bits == 4)
    hipLaunchKernelGGL(pack_codes_kernel<4>, grid, block, 0, (hipStream_t)stream, (const unsigned char*)codes_u8,
                       (unsigned char*)packed, nu);
  else
    hipLaunchKernelGGL(pack_codes_kernel<3>, grid, block, 0, (hipStream_t)stream, (const unsigned char*)codes_u8,
                       (unsigned char*)packed, nu);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

extern "C" int palu_unpack_codes(const void* packed, void* codes_u8, int64_t ncodes, int bits, palu_stream_t stream) {
  PALU_REQUIRE(codes_u8 && packed && ncodes >= 0 && ncodes % 8 == 0 && (bits == 3 || bits == 4), PALU_ERR_ARG,
               "unpack_codes: need bits in {3,4} and a multiple of 8 codes");
  if (ncodes == 0) return PALU_OK;
  const int64_t nu = ncodes / 8;
  dim3 grid((unsigned)((nu + 255) / 256)), block(256);
  if (bits == 4)
    hipLaunchKernelGGL(unpack_codes_kernel<4>, grid, block, 0, (hipStream_t)stream, (const unsigned char*)packed,
                       (unsigned char*)codes_u8, nu);
  else
    hipLaunchKernelGGL(unpack_codes_kernel<3>, grid, block, 0, (hipStream_t)stream, (const unsigned char*)packed,
                       (unsigned char*)codes_u8, nu);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}
