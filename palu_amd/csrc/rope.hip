// In-place rotary embedding of a [H, T, 128] fp16 tensor with HF-4.37.2 semantics (the prompt branch,
// kernel/palu_attention.py:204-205: `cos, sin = rotary_emb(x, seq_len)`; `apply_rotary_pos_emb`):
//   angle = fl32(pos) * inv_freq[i]  (fp32),  cos/sin in fp32, CAST TO fp16 (the table is cast to the activation dtype),
//   out[d]      = fp16( fp16(x[d]      * cos) + fp16(-x[d + 64] * sin) )
//   out[d + 64] = fp16( fp16(x[d + 64] * cos) + fp16( x[d]      * sin) )      (half-split pairs, rotate_half)
// i.e. every product and the sum round to fp16 like the reference's fp16 tensor ops.  Row t has position pos0 + t.
// HBM-bound elementwise pass (one read + one write of the tensor).
#include "palu_common.h"

namespace {

__global__ __launch_bounds__(256) void rope_inplace_kernel(h16* x, int64_t sx_h, int64_t sx_t, int T, int pos0,
                                                           const float* inv_freq) {
  // thread = (row t, 8 consecutive pairs): 8 x (d, d+64) -> two 16-byte loads and stores
  const int h = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = idx >> 3, c = idx & 7;
  if (t >= T) return;
  h16* row = x + (int64_t)h * sx_h + (int64_t)t * sx_t;
  h16x8 lo = *reinterpret_cast<const h16x8*>(row + 8 * c);
  h16x8 hi = *reinterpret_cast<const h16x8*>(row + 64 + 8 * c);
  const float pos = (float)(pos0 + t);
  h16x8 olo, ohi;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma clang fp contract(off)   // two fp16 products and one fp16 sum, never an fma
    const float ang = pos * inv_freq[8 * c + e];
    float sn, cs;
    sincosf(ang, &sn, &cs);
    const h16 c16 = (h16)cs, s16 = (h16)sn;
    const h16 a = lo[e], b = hi[e];
    olo[e] = (h16)(a * c16) + (h16)((h16)(-b) * s16);
    ohi[e] = (h16)(b * c16) + (h16)(a * s16);
  }
  *reinterpret_cast<h16x8*>(row + 8 * c) = olo;
  *reinterpret_cast<h16x8*>(row + 64 + 8 * c) = ohi;
}

}  // namespace

extern "C" int palu_rope_f16(void* x, int64_t sx_h, int64_t sx_t, int H, int T, int D, int pos0, const float* inv_freq,
                             palu_stream_t stream) {
  PALU_REQUIRE(x && inv_freq, PALU_ERR_ARG, "rope: null pointer");
  PALU_REQUIRE(D == 128, PALU_ERR_UNSUPPORTED, "rope: head_dim must be 128 (got %d)", D);
  PALU_REQUIRE(H >= 0 && T >= 0 && pos0 >= 0 && (int64_t)pos0 + T < (1 << 24), PALU_ERR_ARG, "rope: bad shape / positions");
  PALU_REQUIRE(((uintptr_t)x & 15) == 0 && sx_h % 8 == 0 && sx_t % 8 == 0, PALU_ERR_ARG, "rope: rows must be 16-byte aligned");
  if (H == 0 || T == 0) return PALU_OK;
  dim3 grid((T * 8 + 255) / 256, H);
  hipLaunchKernelGGL(rope_inplace_kernel, grid, dim3(256), 0, (hipStream_t)stream, (h16*)x, sx_h, sx_t, T, pos0, inv_freq);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}
