// Latent P.V on the matrix cores: wave-private staging of a column slice of the V latents.
//
//   ctx[h, c] = sum_l P[h, l] * V[l, c]        (kernel/palu_attention.py:246-251, one latent group)
//
// as  D^T = V^T . P^T  on v_mfma_f32_16x16x32_f16: M = 16 latent columns (one "col-tile"), N = 16 head
// slots (gs real heads, the rest duplicates whose results are ignored), K = 32 cache rows (one "unit").
// A wave owns NT adjacent col-tiles (NT in {1,2,4}: a 32/64/128-byte slice of every V row) for ALL rows of
// its workgroup's range and keeps the 4*NT accumulators in registers, so the V stream needs no cross-wave
// synchronisation at all:
//   HBM --buffer_load_dwordx4 ... lds (LDS-DMA, no VGPRs, no VALU)--> wave-private LDS ring of units
//       --ds_read_b64_tr_b16 (hardware transpose read: row-major V -> K-contiguous MFMA A operand)--> MFMA
// LDS image of a unit: [32 rows][NT*32 bytes]; a DMA wave-instruction fills 1 KiB lane-linearly, so the
// 32-byte segments of a row are XOR-swizzled through the per-lane SOURCE offset (key(row)) such that the 8
// row segments a 32-lane half reads in one transpose-read fall on 8 different bank octets (conflict free).
// K index of the MFMA (kgroup q = lane/16, element e) <-> unit row 16*(e>>2) + 4*q + (e&3): the transpose
// read of a 16-lane group covers 4 consecutive rows; the probabilities are read in the same order.
#pragma once
#include "palu_common.h"

namespace pvm {

constexpr int UNIT_ROWS = 32;

template <int NT>
struct Cfg {
  static_assert(NT == 1 || NT == 2 || NT == 4, "a wave owns 1, 2 or 4 col-tiles");
  static constexpr int RBV = 32 * NT;          // bytes per LDS row
  static constexpr int UB = UNIT_ROWS * RBV;   // bytes per unit (1 KiB per col-tile)
  static constexpr int CPR = 2 * NT;           // 16-byte chunks per row
  static constexpr int RPI = 64 / CPR;         // rows per DMA wave-instruction
  // swizzle key of a unit row (depends on row & 7 only, and is the same for row and row + 16)
  static __device__ __forceinline__ int key(int row) { return NT == 4 ? (row >> 1) & 3 : (NT == 2 ? (row >> 2) & 1 : 0); }
};

// lane constants of a wave's V path
template <int NT>
struct Lane {
  unsigned dma_voff;   // (lane / CPR) * row_bytes + col0 * 2 + swizzled chunk * 16: source offset inside a DMA piece
  unsigned rd;         // transpose-read byte offset of col-tile 0 inside a unit (rows 0..15 half; + 16*RBV for rows
                       // 16..31); col-tile ct is at rd ^ (ct << 5): the XOR swizzle only touches bits 5-6
  unsigned p_off;      // byte offset of this lane's probabilities inside a [heads][rows] fp16 tile row block
};

template <int NT>
static __device__ __forceinline__ Lane<NT> make_lane(int lane, int col0, unsigned row_bytes, int p_row_stride_bytes) {
  using C = Cfg<NT>;
  Lane<NT> L;
  const int row_in = lane / C::CPR, pchunk = lane % C::CPR;
  const int lchunk = pchunk ^ (C::key(row_in) << 1);
  L.dma_voff = (unsigned)row_in * row_bytes + (unsigned)(col0 * 2 + lchunk * 16);
  const int q = lane >> 4, i = lane & 15;
  const int r = 4 * q + (i >> 2);
  L.rd = (unsigned)(r * C::RBV + (C::key(r) * 32) + 8 * (i & 3));
  // B operand: head slot n = lane & 15 -> head n & 3 (duplicates for n >= 4), rows 4q.. of the unit
  L.p_off = (unsigned)((lane & 3) * p_row_stride_bytes + 4 * q * 2);
  return L;
}

template <bool NONTEMPORAL = false>
static __device__ __forceinline__ void dma_piece(unsigned lds_dst, unsigned voff, const u32x4& rsrc, unsigned soff) {
  if (NONTEMPORAL) {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen nt lds"
        :
        : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
        : "memory");
  } else {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
        : "memory");
  }
}

// Issue the NT DMA pieces of the unit whose first row is row0 (wave-uniform) into LDS bytes [lds_dst, +UB).
// Rows >= L re-read row L-1 (finite data; their probabilities are zero), never anything past the slab.
template <int NT, bool NONTEMPORAL = false>
static __device__ __forceinline__ void dma_unit(const Lane<NT>& ln, const u32x4& rsrc, unsigned lds_dst, int row0, int L,
                                                unsigned row_bytes, int lane) {
  using C = Cfg<NT>;
  if (row0 + UNIT_ROWS <= L) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
      dma_piece<NONTEMPORAL>(lds_dst + (unsigned)(i * 1024), ln.dma_voff,
                rsrc, __builtin_amdgcn_readfirstlane((unsigned)(row0 + i * C::RPI) * row_bytes));
  } else {
    const int row_in = lane / C::CPR;      // rare path (last unit of the cache): lane constants recomputed here
    const unsigned col = ln.dma_voff - (unsigned)row_in * row_bytes;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int row = min(row0 + i * C::RPI + row_in, L - 1);
      dma_piece(lds_dst + (unsigned)(i * 1024), (unsigned)row * row_bytes + col, rsrc, 0u);
    }
  }
}

typedef __attribute__((address_space(3))) h16x4 lds_h16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

static __device__ __forceinline__ h16x4 tr_read(unsigned addr) {
  return __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)addr));
}

// One unit (32 rows) of P.V for the wave's NT col-tiles.  unit_addr: LDS byte address of the unit's ring slot (a
// multiple of 128);
// p_addr: LDS byte address of P[head 0][first row of the unit] (fp16, heads p_row_stride apart, folded into ln.p_off).
template <int NT>
static __device__ __forceinline__ void pv_unit(f32x4 (&acc)[NT], const Lane<NT>& ln, unsigned unit_addr, unsigned p_addr) {
  using C = Cfg<NT>;
  const h16x4 plo = *(const lds_h16x4*)(uintptr_t)(p_addr + ln.p_off);
  const h16x4 phi = *(const lds_h16x4*)(uintptr_t)(p_addr + ln.p_off + 32);
  const h16x8 pf = __builtin_shufflevector(plo, phi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const unsigned a = (unit_addr + ln.rd) ^ (unsigned)(ct << 5);
    const h16x4 lo = tr_read(a);
    const h16x4 hi = tr_read(a + 16 * C::RBV);
    const h16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, acc[ct], 0, 0, 0);
  }
}

}  // namespace pvm
