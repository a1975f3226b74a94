// Host side of the two-band score kernel (abx_rope2_kernel.h): fragment preparation, the low-band RoPE coefficient
// table and its registry, and the launch decision.  The kernel computes what abx_rope_kernel computes (the reference's
// `_abx_fwd`, kernel/abx_rope.py:79-111); the entry points of abx_rope.hip / abx_rope_q.hip select it when they can.
#include <mutex>
#include <vector>

#include "abx_rope2_kernel.h"

namespace {

struct RopeTable {
  const float* inv_freq;   // device pointer the callers pass as `inv_freq`: the registry key
  const u32x4* tab;
  int tile_first, ntiles;  // 128-position tiles covered: [tile_first, tile_first + ntiles)
  float f_low;             // inv_freq[32] as the host computed it: bounds the low band's angles
};
std::mutex g_tab_mutex;
std::vector<RopeTable> g_tabs;
// what palu_rope_table_build wrote where: the T1 / T2 parts of a table sit behind ITS coefficient tiles, so a table must be
// registered with the range it was built for (a prefix or sub-range pointed the position-split kernel at the wrong
// start tables -- silently wrong scores)
struct BuiltTable { const void* table; int tile_first, ntiles; };
std::vector<BuiltTable> g_built;

int two_band_enabled() {     // PALU_ABX_TWO_BAND=0 keeps every launch on abx_rope_kernel (A/B measurements)
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PALU_ABX_TWO_BAND");
    v = e ? (atoi(e) != 0) : 1;
  }
  return v;
}

int g_position_split = -1;       // -1: not read yet; PALU_ABX_SPLIT=0 in the environment starts with the pair-split kernel
int position_split_enabled() {
  if (g_position_split < 0) {
    const char* e = getenv("PALU_ABX_SPLIT");
    g_position_split = e ? (atoi(e) != 0) : 1;
  }
  return g_position_split;
}

// The position-split kernel pays a longer prologue (every wave takes all high-band fragments) for a main loop that needs
// 3/4 of the cycles: it wins once a wave (4 per CU) has ONE 128-position tile (16k positions x 8 groups: 17.1 against 17.9 us,
// 32k: 26.4 against 29.2, 64k: 43.3 against 48.7; a quarter of a tile per wave: equal; profiles/r05_abx_split_vs_pair.txt).
// g_split_min_tiles: tiles per wave from which it is selected (palu_abx_set_position_split(n >= 1) sets it; n < 0: 0 = always)
int g_split_min_tiles = 1;
bool position_split_preferred(const AbxParams& p) {
  const int on = position_split_enabled();
  if (!on) return false;
  int nch = palu_num_cus() / p.G;
  if (nch < 1) nch = 1;
  const int64_t ntiles = ((int64_t)p.L + TL - 1) / TL;
  return ntiles >= (int64_t)g_split_min_tiles * 4 * nch;
}

template <int NKS, int QBITS>
int launch2(const AbxParams& p, int nwg, hipStream_t stream) {
  if (QBITS == 0 && p.acc == nullptr && p.ncols == 0 && position_split_preferred(p)) return palu_abx3_launch(&p, NKS, stream);
  if (p.qfold) {                                     // pre-folded fragments exist for the position-split kernel only
    palu_set_error("abx: pre-folded fragments given to a launch that does not take the position-split kernel");
    return PALU_ERR_UNSUPPORTED;
  }
  if (p.acc == nullptr) return launch_kernel(abx_rope2_kernel<NKS, QBITS, 0>, abx2_smem(NKS), p, nwg, stream);
  if (p.win_pass == 0) return launch_kernel(abx_rope2_kernel<NKS, QBITS, 1>, abx2_smem(NKS), p, nwg, stream);   // first window: store
  if (p.win_pass == 1) return launch_kernel(abx_rope2_kernel<NKS, QBITS, 2>, abx2_smem(NKS), p, nwg, stream);   // middle ones: add
  return launch_kernel(abx_rope2_kernel<NKS, QBITS, 3>, abx2_smem(NKS), p, nwg, stream);                    // last: add, round, store
}

// column windows of rank 96 and of the ranks above 128 (ranks are multiples of 32: palu/rank_search.py:11-17): 128-wide ones, then the
// remainder -- 32 or 64 at their own width, 96 as a 128-wide window with 96 valid columns (measured at C2's shape: a pass
// costs ~30 / 35 / 51 us at width 32 / 64 / 128, so 64 + 32 loses to one padded 128)
int abx2_windows(int R, int* w, int* valid) {
  if (R % 32 || (R <= 128 && R != 96)) return 0;
  int n = 0, c = 0;
  while (R - c >= 128) { w[n] = 128; valid[n++] = 128; c += 128; }
  const int rem = R - c;
  if (rem == 96) { w[n] = 128; valid[n++] = 96; }
  else if (rem) { w[n] = rem; valid[n++] = rem; }
  return n;
}

}  // namespace

size_t palu_abx2_frag_bytes(int H, int G, int R) {
  if (G <= 0 || H != 4 * G) return 0;
  if (R == 32 || R == 64 || R == 128) return abx2_frag_u32x4(G, R / 16) * sizeof(u32x4);
  int wdt[64], val[64];
  const int nw = R <= 128 * 60 ? abx2_windows(R, wdt, val) : 0;
  size_t n = 0;
  for (int i = 0; i < nw; ++i) n += abx2_frag_u32x4(G, wdt[i] / 16);     // one fragment set per column window
  return n * sizeof(u32x4);
}

// fragments of the two-band kernel, written behind the abx_rope_kernel fragments by palu_abx_prepare_b
int palu_abx2_prepare_b(const void* b, int64_t sb_h, int64_t sb_r, int64_t sb_d, int H, int G, int R, void* frag2,
                        hipStream_t stream) {
  if (!palu_abx2_frag_bytes(H, G, R)) return PALU_OK;
  int wdt[64], val[64];
  int nw = abx2_windows(R, wdt, val);
  if (nw == 0) { nw = 1; wdt[0] = val[0] = R; }
  u32x4* dst = (u32x4*)frag2;
  int c0 = 0;
  for (int i = 0; i < nw; ++i) {
    const int nks = wdt[i] / 16;
    const int64_t n_hi = (int64_t)G * 8 * nks * 64;
    const int64_t total = (int64_t)abx2_frag_u32x4(G, nks);
    hipLaunchKernelGGL(abx2_prepare_b_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const h16*)b + (int64_t)c0 * sb_r, sb_h, sb_r, sb_d, G, val[i], nks, dst, n_hi, total);   // rows >= val: 0
    PALU_LAUNCH_CHECK();
    dst += total;
    c0 += val[i];
  }
  return PALU_OK;
}

// table = [coefficient tiles: 1 KB per 128 positions][T1: 256 B per tile][T2: 33 x 256 B] (abx_rope2_kernel.h)
static inline size_t tab_tiles_bytes(int ntiles) { return (size_t)ntiles * 2 * 32 * sizeof(u32x4); }
static inline size_t tab_t1_bytes(int ntiles) { return (size_t)ntiles * ABX2_T1_TILE_FLOATS * sizeof(float); }

extern "C" size_t palu_rope_table_bytes(int npos) {
  if (npos <= 0) return 0;
  const int ntiles = (npos + TL - 1) / TL;
  return tab_tiles_bytes(ntiles) + tab_t1_bytes(ntiles) + ABX2_T2_FLOATS * sizeof(float);
}

extern "C" int palu_rope_table_build(const float* inv_freq, int pos_first, int npos, void* table, palu_stream_t stream) {
  PALU_REQUIRE(inv_freq && table, PALU_ERR_ARG, "rope_table_build: null pointer");
  PALU_REQUIRE(pos_first >= 0 && pos_first % TL == 0 && npos > 0, PALU_ERR_ARG,
               "rope_table_build: pos_first must be a multiple of %d and npos > 0 (got %d, %d)", TL, pos_first, npos);
  PALU_REQUIRE(((uintptr_t)table & 15) == 0, PALU_ERR_ARG, "rope_table_build: table must be 16-byte aligned");
  const int ntiles = (npos + TL - 1) / TL;
  const int64_t total = (int64_t)ntiles * 64;
  hipLaunchKernelGGL(abx2_rope_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, inv_freq,
                     pos_first / TL, ntiles, (u32x4*)table);
  PALU_LAUNCH_CHECK();
  float* t1 = (float*)((char*)table + tab_tiles_bytes(ntiles));
  const int64_t total2 = (int64_t)ntiles * 32 + 33 * 32;
  hipLaunchKernelGGL(abx2_rope_start_kernel, dim3((unsigned)((total2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, inv_freq,
                     pos_first / TL, ntiles, t1, (float*)((char*)t1 + tab_t1_bytes(ntiles)));
  PALU_LAUNCH_CHECK();
  {
    std::lock_guard<std::mutex> lk(g_tab_mutex);
    bool found = false;
    for (auto& b : g_built)
      if (b.table == table) { b = BuiltTable{table, pos_first / TL, ntiles}; found = true; }
    if (!found) {
      if (g_built.size() > 256) g_built.erase(g_built.begin());
      g_built.push_back(BuiltTable{table, pos_first / TL, ntiles});
    }
  }
  return PALU_OK;
}

extern "C" int palu_rope_table_register(const float* inv_freq, const void* table, int pos_first, int npos, float inv_freq_32) {
  PALU_REQUIRE(inv_freq && table && pos_first >= 0 && pos_first % TL == 0 && npos > 0, PALU_ERR_ARG,
               "rope_table_register: bad arguments");
  PALU_REQUIRE(inv_freq_32 > 0.f, PALU_ERR_ARG, "rope_table_register: inv_freq[32] must be positive");
  std::lock_guard<std::mutex> lk(g_tab_mutex);
  for (const auto& b : g_built)
    if (b.table == table)
      PALU_REQUIRE(b.tile_first == pos_first / TL && b.ntiles == (npos + TL - 1) / TL, PALU_ERR_ARG,
                   "rope_table_register: the table was built for positions [%d, %d) and must be registered with that range "
                   "(got pos_first %d, npos %d): its start tables follow its own coefficient tiles",
                   b.tile_first * TL, (b.tile_first + b.ntiles) * TL, pos_first, npos);
  for (auto& t : g_tabs)
    if (t.inv_freq == inv_freq) {
      t = RopeTable{inv_freq, (const u32x4*)table, pos_first / TL, (npos + TL - 1) / TL, inv_freq_32};
      return PALU_OK;
    }
  g_tabs.push_back(RopeTable{inv_freq, (const u32x4*)table, pos_first / TL, (npos + TL - 1) / TL, inv_freq_32});
  return PALU_OK;
}

extern "C" int palu_rope_table_unregister(const float* inv_freq) {
  std::lock_guard<std::mutex> lk(g_tab_mutex);
  for (size_t i = 0; i < g_tabs.size(); ++i)
    if (g_tabs[i].inv_freq == inv_freq) {
      g_tabs.erase(g_tabs.begin() + (long)i);
      return PALU_OK;
    }
  return PALU_OK;
}

// 1 when a launch with these positions would take the two-band kernel (introspection for tests and the bench)
extern "C" int palu_abx_two_band_selected(const float* inv_freq, int H, int G, int L, int R, int pos0) {
  if (!two_band_enabled() || !palu_abx2_frag_bytes(H, G, R) || L <= 0 || pos0 < 0 || pos0 % TL) return 0;
  if (R > 128 && ((int64_t)L + 3 * 128) * R * 2 >= ((int64_t)1 << 31)) return 0;
  // positions: the high band corrects the oracle's fp32 angle rounding to first order in the residual (<= 2^-6 rad below 2^19:
  // the neglected term stays below 1.3e-4 on the highest-frequency pair alone); the tables hold 2^18 + 4096 positions -- a
  // 256k prompt and what a generation appends
  if ((int64_t)pos0 + L > 266240) return 0;
  std::lock_guard<std::mutex> lk(g_tab_mutex);
  for (const auto& t : g_tabs)
    if (t.inv_freq == inv_freq) {
      const int first = pos0 / TL, last = (pos0 + L + TL - 1) / TL;
      // psi_max = 64 f_32 <= 0.7 rad keeps the degree-7 Taylor remainder below 0.7^8 / 8! = 1.4e-6 (theta >= ~8400 at D = 128);
      // f_32 (pos0 + L) < 2700 rad: the low band uses the exact angle l f, the oracle rounds l f to fp32 first -- <= 2^-13 rad
      // below 4096 rad, on the band's highest frequency only.  Measured at the end of a 262 145-position cache (2621 rad,
      // tests/test_two_band_gpu.py::test_two_band_at_256k_positions): error vs fp64 and rms at the oracle's own level
      return first >= t.tile_first && last <= t.tile_first + t.ntiles && 64.0f * t.f_low <= 0.7f &&
             t.f_low * (float)(pos0 + L) < 2700.0f;
    }
  return 0;
}

// Process-wide switch between the two forms of the two-band kernel (returns the previous value): 1 (default) the
// position-split kernel (abx_rope3_kernel.h) where it applies, 0 the pair-split kernel (abx_rope2_kernel.h) everywhere
extern "C" int palu_abx_set_position_split(int enable) {
  const int o = position_split_enabled() ? (g_split_min_tiles > 0 ? g_split_min_tiles : -1) : 0;
  g_position_split = enable ? 1 : 0;
  if (enable > 0) g_split_min_tiles = enable;        // from n tiles per wave on
  if (enable < 0) g_split_min_tiles = 0;             // every shape the kernel takes (tests, A/B measurements)
  return o;
}

// 1 when an fp16 launch with these positions takes the position-split form (introspection for tests and the bench)
extern "C" int palu_abx_position_split_selected(const float* inv_freq, int H, int G, int L, int R, int pos0) {
  if (!(R == 32 || R == 64 || R == 128) || !palu_abx_two_band_selected(inv_freq, H, G, L, R, pos0)) return 0;
  AbxParams p = {};
  p.G = G;
  p.L = L;
  return position_split_preferred(p) ? 1 : 0;
}

static unsigned long long* g_abx2_dbg = nullptr;
// debug: device buffer of >= 176 KB that workgroup 0 of the next two-band launches dumps its first W image into (0 = off)
extern "C" void palu_abx2_debug_buffer(void* ptr) { g_abx2_dbg = (unsigned long long*)ptr; }

int palu_abx2_try_launch(const void* params, int nwg, int bits, hipStream_t stream) {
  AbxParams p = *reinterpret_cast<const AbxParams*>(params);
  p.dbg = g_abx2_dbg;
  // (a windowed pass arrives with R = the window's width, acc set and win_pass = 0 for the first window, 1 for the middle ones, 2 for the last)
  if (!p.bfrag2 || p.ncols < 0 || p.ncols >= p.R || p.qgroup < 0 || p.qgroup % 32 != 0 || p.HB != 1 || p.gs != 4) return PALU_ABX2_SKIP;
  if (bits == 0 && p.qgroup != 0) return PALU_ABX2_SKIP;
  if (!(p.R == 32 || p.R == 64 || p.R == 128)) return PALU_ABX2_SKIP;
  if (!palu_abx_two_band_selected(p.inv_freq, p.H, p.G, p.L, p.R, p.pos0)) return PALU_ABX2_SKIP;
  {
    std::lock_guard<std::mutex> lk(g_tab_mutex);
    const RopeTable* t = nullptr;
    for (const auto& e : g_tabs)
      if (e.inv_freq == p.inv_freq) t = &e;
    if (!t) return PALU_ABX2_SKIP;
    p.rope_tab = t->tab;
    p.tab_tile0 = p.pos0 / TL - t->tile_first;
    p.rope_t1 = (const float*)((const char*)t->tab + tab_tiles_bytes(t->ntiles));
    p.rope_t2 = (const float*)((const char*)p.rope_t1 + tab_t1_bytes(t->ntiles));
  }
  if (bits == 0) {
    switch (p.R) {
      case 32: return launch2<2, 0>(p, nwg, stream);
      case 64: return launch2<4, 0>(p, nwg, stream);
      default: return launch2<8, 0>(p, nwg, stream);
    }
  }
  if (bits == 3) {
    switch (p.R) {          // (narrower 3-bit rows are staged by 2 / 1 lanes per row: 32 codes = 12 bytes each)
      case 32: return launch2<2, 3>(p, nwg, stream);
      case 64: return launch2<4, 3>(p, nwg, stream);
      default: return launch2<8, 3>(p, nwg, stream);
    }
  }
  switch (p.R) {
    case 32: return launch2<2, 4>(p, nwg, stream);
    case 64: return launch2<4, 4>(p, nwg, stream);
    default: return launch2<8, 4>(p, nwg, stream);
  }
}

// Rank 96 or a rank above 128 with 4 heads per group as passes of the two-band kernel over its column windows (128, 128, ..., 64, 32),
// fp32 partial scores in `scratch` (palu_abx_scratch_bytes); the last pass adds its window, rounds and stores `out`.  `params`: AbxParams of the whole
// problem (x / xq, bfrag2 = the window fragment sets, R = the full rank).  PALU_ABX2_SKIP when the shape or the positions
// do not allow it.  bits = 0 / 3 / 4.
int palu_abx2_try_launch_windows(const void* params, int nwg, int bits, void* scratch, int64_t acc_ld, hipStream_t stream) {
  const AbxParams p0 = *reinterpret_cast<const AbxParams*>(params);
  int wdt[64], val[64];
  const int nw = abx2_windows(p0.R, wdt, val);
  if (nw == 0 || (nw > 1 && !scratch) || !p0.bfrag2 || p0.HB != 1 || p0.gs != 4 || p0.ncols != 0) return PALU_ABX2_SKIP;
  for (int i = 0; i < nw; ++i)
    if (!palu_abx_two_band_selected(p0.inv_freq, p0.H, p0.G, p0.L, wdt[i], p0.pos0)) return PALU_ABX2_SKIP;
  if (bits == 0 && ((int64_t)p0.L + 3 * 128) * p0.sx_l * 2 >= ((int64_t)1 << 31)) return PALU_ABX2_SKIP;
  // a padded last window of fp16 rows (96 valid of 128 columns) over-reads into what follows a row and multiplies it by zero
  // fragment rows: fine while that is the next row's (finite) latents, i.e. rows packed back to back -- behind a wider row
  // stride sits padding nobody initialised (0 x NaN = NaN): those launches take the one-band kernel's masked staging
  for (int i = 0; i < nw; ++i)
    if (bits == 0 && val[i] != wdt[i] && p0.sx_l != p0.R) return PALU_ABX2_SKIP;
  const u32x4* frag = p0.bfrag2;
  int c0 = 0;
  for (int i = 0; i < nw; ++i) {
    AbxParams p = p0;
    p.R = wdt[i];
    p.ncols = val[i] == wdt[i] ? 0 : val[i];
    p.bfrag2 = frag;
    p.acc = nw > 1 ? (float*)scratch : nullptr;      // (rank 96: one padded window straight to `out`)
    p.acc_ld = acc_ld;
    p.win_pass = i == 0 ? 0 : i + 1 < nw ? 1 : 2;
    p.qcol0 = c0;                                    // (per-column-group metas: the pair index counts from the row's start)
    if (bits == 0) p.x = p0.x + c0;
    else p.xq = p0.xq + (size_t)c0 * bits / 8;
    const int rc = palu_abx2_try_launch(&p, nwg, bits, stream);
    if (rc != PALU_OK) return rc == PALU_ABX2_SKIP && i > 0 ? PALU_ERR_LAUNCH : rc;
    frag += abx2_frag_u32x4(p0.G, wdt[i] / 16);
    c0 += wdt[i];
  }
  return PALU_OK;
}
