/*
 * palu_hip.h -- C ABI of the MI355X (gfx950) low-rank-KV attention decode path.
 *
 * The reference (shadowpa0327/Palu) is pure Python; its boundary for this path is the Python
 * call `kernel.abx_rope.abx(a, b, x)` (kernel/abx_rope.py:114-150, called at
 * kernel/palu_attention.py:219) plus the stock torch ops of the decode branch
 * (kernel/palu_attention.py:207-257).  This header is what a ctypes/cffi stub binds instead
 * (see INTEGRATION.md): plain pointers + sizes, caller-owned DEVICE buffers, a HIP stream.
 *
 * Conventions for every entry point:
 *   - pointers are device pointers unless the name ends in _host;
 *   - strides are in ELEMENTS of the pointed-to type;
 *   - work is enqueued on `stream` (a hipStream_t cast to void*; NULL = default stream),
 *     nothing synchronises, nothing allocates: every call is hipGraph-capturable
 *     (run_latency_attention.py:81-90 captures the step in a graph);
 *   - returns 0 on success or a negative PALU_ERR_* code; palu_last_error() gives the text.
 *     Never throws, never aborts.
 */
#ifndef PALU_HIP_H
#define PALU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PALU_OK 0
#define PALU_ERR_ARG (-1)      /* bad shape / stride / alignment / null pointer          */
#define PALU_ERR_UNSUPPORTED (-2) /* valid but not implemented (e.g. head_dim != 128)    */
#define PALU_ERR_LAUNCH (-3)   /* HIP reported a launch error                            */

typedef void* palu_stream_t;

const char* palu_last_error(void);
int palu_version(void);

/* ------------------------------------------------------------------------------------------
 * RoPE frequencies.  kernel/pytorch_reference.py:4: inv_freq[i] = 1 / theta^(2i/D) in fp32.
 * Host helper (fills D/2 floats).  The kernels take the table as a device pointer so that a
 * caller can pass exactly the values its framework computed.
 */
int palu_rope_inv_freq_host(float theta, int head_dim, float* out_host);

/* ------------------------------------------------------------------------------------------
 * abx: fused reconstruct-K -> RoPE -> q.K^T   (replaces kernel/abx_rope.py:44-150)
 *
 *   out[h, l] = sum_d a[h,d] * RoPE_{pos0+l}( sum_r x[h/gs, l, r] * b[h, r, d] )[d]
 *
 * a: [H, D] fp16 (the already-rotated query; reference shape [H,1,D]), b: [H, R, D] fp16,
 * x: [G, L, R] fp16 latent keys (row l = absolute position pos0 + l), out: [H, L] fp16.
 * No 1/sqrt(D) (palu_attention.py:219 divides afterwards).  D must be 128 (the reference
 * hard-codes it, abx_rope.py:21-22,125); R % 8 == 0; H % G == 0; L >= 0 arbitrary (masked tail).
 *
 * `b` is a weight: it is re-laid-out once into MFMA A-operand fragments by palu_abx_prepare_b
 * (bfrag must hold palu_abx_bfrag_bytes() bytes) and the hot call consumes the fragments.
 * x rows must be 16-byte aligned (sx_g % 8 == 0, sx_l % 8 == 0, innermost stride 1).
 */
size_t palu_abx_bfrag_bytes(int H, int G, int R);
/* Numerics switch of the R in {32,64,128} fast path (process-wide, returns the previous value):
 * 1 (default) folds the query into the B fragments once per launch -- one extra fp16 operand
 * rounding, same size as the oracle's own rounding of K (abx_rope.py:164); 0 keeps q in fp32. */
int palu_abx_set_fold(int enable);
int palu_abx_prepare_b(const void* b, int64_t sb_h, int64_t sb_r, int64_t sb_d,
                       int H, int G, int R, int D, void* bfrag, palu_stream_t stream);
int palu_abx_rope_f16(const void* a, int64_t sa_h, int64_t sa_d,
                      const void* bfrag,
                      const void* x, int64_t sx_g, int64_t sx_l,
                      void* out, int64_t so_h,
                      int H, int G, int L, int R, int D,
                      const float* inv_freq, int pos0, palu_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PALU_HIP_H */
