// Shared device/host helpers for the Palu decode-path HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/palu_hip.h"

typedef _Float16 h16;
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
typedef __attribute__((ext_vector_type(4))) _Float16 h16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define PALU_WAVE 64

// Experiment knobs (wave priorities, workgroups per CU, timeline dumps, ...) are read from the environment only in a build
// with -DPALU_EXPERIMENTS (PALU_EXTRA_CFLAGS=-DPALU_EXPERIMENTS python -m palu_amd.build --force); the shipped library runs
// the measured defaults.  The kernel-selection switches that tests and A/B runs use stay: PALU_ABX_TWO_BAND, PALU_FUSED_ATTN,
// PALU_PVQ_DIRECT, PALU_PV_DIRECT.
#include <stdlib.h>
static inline const char* palu_exp_env(const char* name) {
#ifdef PALU_EXPERIMENTS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Set by every entry point on failure; read with palu_last_error().
void palu_set_error(const char* fmt, ...);

#define PALU_REQUIRE(cond, code, ...)        \
  do {                                       \
    if (!(cond)) {                           \
      palu_set_error(__VA_ARGS__);           \
      return (code);                         \
    }                                        \
  } while (0)

#define PALU_LAUNCH_CHECK()                                             \
  do {                                                                  \
    hipError_t e_ = hipGetLastError();                                  \
    if (e_ != hipSuccess) {                                             \
      palu_set_error("HIP launch failed: %s", hipGetErrorString(e_));   \
      return PALU_ERR_LAUNCH;                                           \
    }                                                                   \
  } while (0)

int palu_num_cus();   // cached hipDeviceProp multiProcessorCount of the current device
int palu_func_max_lds(const void* func, int bytes);   // MaxDynamicSharedMemorySize once per (kernel, current device); PALU_OK or error

// internal cross-TU helper (abx_rope.hip): the two-band fragments inside a palu_abx_prepare_b allocation (null: none for this shape)
const void* palu_abx_two_band_frags(const void* bfrag, int H, int G, int R);

// internal cross-TU helper (quant.hip): both new latent rows of a decode step in one launch
int palu_quantize_pack_kv(const void* k, int64_t sk_g, void* k_codes, int64_t skc_g, void* k_meta, int64_t skm_g, int Rk,
                          const void* v, int64_t sv_g, void* v_codes, int64_t svc_g, void* v_meta, int64_t svm_g, int Rv,
                          int G, int bits, palu_stream_t stream);

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// split workspace of the P.V kernels (decode_pv.hip, decode_fused.hip):  stats [H][2] (padded) | part [H][ns][Rv] | ml [H][ns][2]
// -- the (max, sum) pairs the probs kernel and the callers read sit at offset 0 whatever split count a kernel chose
static inline size_t pv_ws_stats_floats(int H) { return ((size_t)2 * H + 63) / 64 * 64; }

