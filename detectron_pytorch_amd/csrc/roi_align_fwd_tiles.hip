// roi_align_fwd_tiles.hip -- RoIAlign forward (Caffe2 semantics, roi_align_kernel.cu:16-121), NCHW, ONE launch, no
// workspace: the TILE-CENTRIC forward for gfx950.
//
// Why.  A per-RoI gather (roi_align_fwd_records) moves every RoI's window into LDS on its own: on the config-2 input
// the windows overlap 2.2x and a ~70-byte row segment drags in 1.5 cache lines, so 430 MB pass from L2 to the L1s for
// 60 MB of distinct pixels, and a separate launch has to sort the RoIs along a sweep to keep even that local.  Here the
// roles are swapped: a workgroup owns a TH x TW tile of ONE feature map for 32 channels, loads it (plus a 4-pixel halo
// to the right and below) ONCE with LDS-DMA, and computes every output bin whose first sample's lower tap lies in the
// tile -- whichever RoI it belongs to.  Every feature byte is fetched once per channel group (+ halo), neighbouring
// tiles run side by side, and nothing has to be sorted, ranked or prepared.
//
// Two ways to get the per-tile tables:
//   * no scratch (mi_roi_align_forward): every workgroup scans the RoIs and builds the tables of its tile itself;
//   * with scratch (mi_roi_align_forward_ws): roi_align_tiles_prepare, one small workgroup per TILE, does that once and
//     leaves "descriptor blocks" in the workspace; the eight channel groups of a tile just DMA the block beside the image.
//
//   workgroup = (tile, 32-channel group), 512 lanes.  blockIdx % ncg = channel group: with round-robin dispatch an
//       XCD's L2 only ever sees "its" channel slabs.
//   1. every lane fetches the RoIs it will test (lane = RoI), THEN the tile DMA is issued (the loads are ordered in
//      front of the DMA so that waiting for them does not wait for the image): wave w moves piece w of all 32 planes.
//      Rows / columns past the map repeat the last row / column, so that "upper tap = lower tap + 1" holds for a
//      sample clamped to the border too (the reference reads the border pixel twice, weights 1 and 0).
//   2. scan: lane = RoI, multiplications and compares only -- level, image, and the clamped RoI rectangle (+-1) against
//      the tile.  Survivors (a superset of the RoIs with a bin here) are compacted into LDS.
//   3. per batch of <= 16 (8) survivors: geometry (one lane per RoI), axis tables (lane = (RoI, axis, sample): the
//      reference's fp32 operations for coordinate, taps and weights, roi_align_kernel.cu:74-110,16-52), then per
//      (RoI, axis, bin) the footprint check and the range [pa, pb) of bins that belong to this tile, then one
//      descriptor per work unit = (RoI, bin row).
//   4. a unit is computed by 32 lanes = 32 channels (planes with an odd stride: one ds_read2_b32 of a half-wave hits
//      32 banks).  Per sample column ONE address serves four loads: rows y, y+1 at column x (ds_read2_b32) and at
//      x + 1 (immediate offset); the pair (row y, row y+1) is a packed operand, so a bin costs 4 address adds and
//      11 packed FMAs / multiplies per channel.  A bin row leaves as a 16-byte + 12-byte store per channel (7 bins) --
//      every output element is written exactly once, by one tile.
//   RoIs the tables cannot describe (a sample outside the [-1, size] band, more samples per axis than the tables
//   hold, a bin whose taps span more than the halo) are computed by the tile that owns their first anchor with the
//   reference's operation order straight from global memory (bit-exact); RoIs of a non-existent image get zeros from
//   tile (index mod tiles).
//
// Arithmetic: taps and weights exactly the reference's; the sum is evaluated per tap row as
//   sum_iy { hy/count, ly/count } * ( sum_ix hx * F[.][xlo] + lx * F[.][xlo + 1] )   with FMAs,
// i.e. fp32 rounding differences only (measured ~5e-7 on unit-variance data; contract 1e-4).
#include "common.h"
#include "lds_dma.h"
#include "roi_align_device.h"

#include <type_traits>

namespace mi {
namespace {

constexpr int kCt = 32;         // channels per workgroup
constexpr int kThreads = 512;
constexpr int kNWaves = kThreads / 64;
constexpr int kGroups = kThreads / 32;  // units in flight
constexpr int kHalo = 4;        // rows below / columns right of the tile that a bin's taps may reach
constexpr int kCandCap = 64;    // scan survivors held in LDS per pass
constexpr int kMaxSamples = 32; // samples per axis the generic tables hold

struct Cand {  // 32 bytes
  float b, x1, y1, x2, y2;
  int id, pad0, pad1;
};
struct TabEnt {  // one axis sample: LDS byte offset of its lower tap (clamped into the image), weights (the y weights
  int off;       // already divided by count), lower tap relative to the tile origin in pixels (unclamped)
  float hw, lw;
  int lo_rel;
};
struct AxEnt {  // one scan survivor, one axis (32 bytes)
  float start, bin;
  int g, flags, pa, pb, r, pad;
};
enum : int { kEntNotFast = 1, kEntZero = 2 };
constexpr int kEB = 16;       // RoIs per batch (the unit prefix lives in 16 lanes)
constexpr int kPreBlocks = 2; // descriptor blocks roi_align_tiles_prepare leaves per tile (survivors 0..31)

// Everything the units need to know about one batch of <= kEB RoIs of a tile; the same bytes in LDS and in the workspace.
template <int S>
struct DescBlock {
  TabEnt tabs[2][kEB][S];  // [axis][RoI][sample]; axis 0 = y
  AxEnt axes[2][kEB];
  int misc[16];            // [0] RoIs of the batch, [1] more batches follow, [2] scan survivors of the tile
};

template <int kSR, int kA, int TH, int TW>
struct TileCfg {
  static constexpr int kRows = TH + kHalo, kPitch = TW + kHalo;
  static constexpr int kPx = kRows * kPitch;
  static constexpr int kPlane = kPx | 1;  // odd: 32 planes -> 32 banks
  static constexpr int kPieces = (kPx + 63) / 64;
  static constexpr int S = (kA > 0 && kSR > 0) ? kA * kSR : kMaxSamples;  // table entries per axis and RoI
  static constexpr int PB = kA > 0 ? kA : kMaxSamples;                    // bins per axis
  using Block = DescBlock<S>;
  static constexpr size_t kImgBytes = (size_t)kCt * kPlane * 4;
  static constexpr size_t kLdsBytes = kImgBytes + sizeof(Block) + 2 * kCandCap * sizeof(Cand) + 64;
  static constexpr size_t kPrepLdsBytes = sizeof(Block) + kPreBlocks * kEB * sizeof(Cand) + 64;
  static_assert(kPitch + 1 < 256, "ds_read2_b32 offsets");
  static_assert(sizeof(Block) % 16 == 0 && kImgBytes % 4 == 0, "16-byte DMA pieces");
};

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

// LDS words at byte address a + 4 * kCol and one image row further down: the two tap rows of one column
template <int kPitch, int kCol>
__device__ __forceinline__ v2f lds_rows(unsigned byte_addr) {
  const lds_cfloat_t q = reinterpret_cast<lds_cfloat_t>((uintptr_t)byte_addr);
  v2f r;
  r.x = q[kCol];
  r.y = q[kCol + kPitch];
  return r;
}
__device__ __forceinline__ v2f splat(float w) { return (v2f){w, w}; }
// LDS traffic of ONE wavefront is processed in order: between two phases of a wave that communicate through LDS only the
// compiler has to be kept from reordering them
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- one tile -------------------------------------------------------------------------------------------------------
struct TileCtx {
  int tile_global, lvl, n, x0, y0, height, width;
  float scale;
  const float* feat;
};
// (integer divisions run on the vector unit: readfirstlane brings the wave-uniform results back to SGPRs, otherwise
// everything derived from them occupies vector registers)
template <int TH, int TW>
__device__ __forceinline__ TileCtx decode_tile(const LevelTable& lv, int tile_global) {
  // The level table is only ever indexed with compile-time constants: a run-time index into a kernel argument turns
  // into a chain of dependent scalar loads (one K$ miss each, ~2 us before the first useful instruction); this way the
  // whole table arrives with the first loads and the level is picked by selects.
  TileCtx tc;
  tc.tile_global = tile_global;
  int lvl = 0, base = 0;
  tc.height = lv.height[0];
  tc.width = lv.width[0];
  tc.scale = lv.scale[0];
  tc.feat = lv.feat[0];
#pragma unroll
  for (int l = 1; l < kMaxLevels; l++) {
    if (l < lv.count && tile_global >= lv.tile_base[l]) {
      lvl = l;
      base = lv.tile_base[l];
      tc.height = lv.height[l];
      tc.width = lv.width[l];
      tc.scale = lv.scale[l];
      tc.feat = lv.feat[l];
    }
  }
  const int tile = tile_global - base;
  tc.lvl = lvl;
  const int tiles_x = (tc.width + TW - 1) / TW, tiles_y = (tc.height + TH - 1) / TH;
  tc.n = uniform(tile / (tiles_x * tiles_y));
  const int trem = tile - tc.n * tiles_x * tiles_y;
  const int tyi = uniform(trem / tiles_x), txi = trem - tyi * tiles_x;
  tc.x0 = txi * TW;
  tc.y0 = tyi * TH;
  return tc;
}

// One RoI against a tile, multiplications and compares only: true for every RoI that has a bin there or that the tile
// owns (a superset; the tables decide).  A bin's anchor is the lower tap of its first sample, whose coordinate lies in
// [start, start + length]; taps are clamped to the map.
template <int TH, int TW>
__device__ __forceinline__ bool tile_test(const TileCtx& tc, int levels_count, int batch, int ntiles, int i, float rb,
                                          float rx1, float ry1, float rx2, float ry2, int rl) {
  const int b = (int)rb;
  if (b < 0 || b >= batch) return (i % ntiles) == tc.tile_global;  // zeros, written by tile (i mod tiles)
  const int l = min(max(rl, 0), levels_count - 1);
  if (l != tc.lvl || b != tc.n) return false;
  const float h_max = (float)(tc.height - 1), w_max = (float)(tc.width - 1);
  const float sy = ry1 * tc.scale, sx = rx1 * tc.scale;
  const float ey = sy + fmaxf(ry2 * tc.scale - sy, 1.f), ex = sx + fmaxf(rx2 * tc.scale - sx, 1.f);
  const float ya = fminf(fmaxf(sy, 0.f), h_max) - 1.f, yb = fminf(fmaxf(ey, 0.f), h_max) + 1.f;
  const float xa = fminf(fmaxf(sx, 0.f), w_max) - 1.f, xb = fminf(fmaxf(ex, 0.f), w_max) + 1.f;
  return ya < (float)(tc.y0 + TH) && yb >= (float)tc.y0 && xa < (float)(tc.x0 + TW) && xb >= (float)tc.x0;
}

// Scan by ONE wave in RoI order: survivors with ordinal in [win_lo, win_lo + cap) go to list[0 .. cap) (and to a second
// copy `list2` when given); returns the number of survivors of the tile.  Four RoIs per lane are fetched together.
template <int TH, int TW>
__device__ __forceinline__ int scan_ordered(const TileCtx& tc, int levels_count, int batch, int ntiles,
                                            const float* __restrict__ rois, const int* __restrict__ levels,
                                            int num_rois, int win_lo, int cap, Cand* list, Cand* list2) {
  const int lane = threadIdx.x & 63;
  int count = 0;
  constexpr int kPer = 4;
  for (int base = 0; base < num_rois; base += 64 * kPer) {
    float rv[kPer][5];
    int rl[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int i = base + k * 64 + lane;
      rl[k] = 0;
#pragma unroll
      for (int j = 0; j < 5; j++) rv[k][j] = 0.f;
      if (i < num_rois) {
#pragma unroll
        for (int j = 0; j < 5; j++) rv[k][j] = rois[(long long)i * 5 + j];
        if (levels != nullptr) rl[k] = levels[i];
      }
    }
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int i = base + k * 64 + lane;
      const bool hit = i < num_rois && tile_test<TH, TW>(tc, levels_count, batch, ntiles, i, rv[k][0], rv[k][1], rv[k][2],
                                                          rv[k][3], rv[k][4], rl[k]);
      const unsigned long long m = __ballot(hit);
      const int ord = count + __popcll(m & ((1ull << lane) - 1ull)) - win_lo;
      if (hit && ord >= 0 && ord < cap) {
        Cand cr;
        cr.b = rv[k][0], cr.x1 = rv[k][1], cr.y1 = rv[k][2], cr.x2 = rv[k][3], cr.y2 = rv[k][4], cr.id = i, cr.pad0 = 0,
        cr.pad1 = 0;
        list[ord] = cr;
        if (list2 != nullptr) list2[ord] = cr;
      }
      count += __popcll(m);
    }
  }
  return count;
}

// One batch of <= kEB survivors, ONE axis, by ONE wave: axis entries, sample tables (the reference's fp32 operations,
// roi_align_kernel.cu:74-110, 16-52), and per bin the footprint check and the range [pa, pb) of bins whose anchor lies in
// the tile.  mycand: the wave's survivor list; myax / mytab: axis `axis` of the descriptor block.
template <int kSR, int kA, int TH, int TW>
__device__ __forceinline__ void build_axis(const TileCtx& tc, int axis, int batch, int ah, int aw, int sr,
                                           const Cand* mycand, AxEnt* myax, TabEnt* mytab, int b0, int ne) {
  using Cfg = TileCfg<kSR, kA, TH, TW>;
  constexpr int kPitch = Cfg::kPitch, S = Cfg::S, PB = Cfg::PB;
  const int lane = threadIdx.x & 63;
  const int aligned = axis == 0 ? ah : aw, size = axis == 0 ? tc.height : tc.width, origin = axis == 0 ? tc.y0 : tc.x0;
    if constexpr (kSR == 2 && kA > 0) {
      // compile-time 2 x 2 grid: one pass.  lane = (RoI, sample); the two samples of a bin sit in neighbouring lanes, so
      // the footprint of a bin and its tile come from a lane exchange instead of a second pass over LDS.
      if (lane < ne) {
        const Cand cr = mycand[b0 + lane];
        const int b = (int)cr.b;
        AxEnt en;
        en.start = en.bin = 0.f;
        en.g = 2;
        en.flags = (b < 0 || b >= batch) ? kEntZero : 0;
        en.pa = 0x7fff;
        en.pb = 0;
        en.r = cr.id;
        en.pad = 0;
        myax[lane] = en;
      }
      wave_sync();
      constexpr int kPasses = (kEB * S + 63) / 64;
#pragma unroll 1
      for (int pass = 0; pass < kPasses; pass++) {  // (unrolling the passes was measured: slower, 49 us against 40)
        const int t = pass * 64 + lane;
        if (t >= ne * S) continue;
        const int e = t / S, s = t - e * S;
        const Cand cr = mycand[b0 + e];
        const float lo_c = axis == 0 ? cr.y1 : cr.x1, hi_c = axis == 0 ? cr.y2 : cr.x2;
        // roi_align_kernel.cu:79-88, 106-110
        const float start = lo_c * tc.scale;
        const float len = fmaxf(hi_c * tc.scale - start, 1.f);
        const float bin = len / (float)kA;
        const int p = s >> 1, i = s & 1;
        float v = start + (float)p * bin + ((float)i + .5f) * bin / 2.f;
        int flag = (v < -1.0f || v > (float)size) ? (int)kEntNotFast : 0;
        // roi_align_kernel.cu:27-52
        if (v <= 0) v = 0;
        int lo = (int)v;
        float lw, hw;
        if (lo >= size - 1) {
          lo = size - 1;  // the reference's upper tap is the same pixel; here it is the copy one row / column on
          lw = 0.f;
          hw = 1.f;
        } else {
          lw = v - (float)lo;
          hw = 1.f - lw;
        }
        TabEnt te;
        te.lo_rel = lo - origin;
        if (axis == 0) {
          te.off = min(max(te.lo_rel, 0), Cfg::kRows - 2) * kPitch * 4;
          te.hw = hw / 4.f;  // count = 2 * 2 (roi_align_kernel.cu:101)
          te.lw = lw / 4.f;
        } else {
          te.off = min(max(te.lo_rel, 0), kPitch - 2) * 4;
          te.hw = hw;
          te.lw = lw;
        }
        mytab[e * S + s] = te;
        const int other = __shfl_xor(te.lo_rel, 1);  // the bin's second sample (S and 64 are even: same pass)
        if (i == 0) {
          if (other + 1 - te.lo_rel > kHalo) flag |= kEntNotFast;
          if (te.lo_rel >= 0 && te.lo_rel < (axis == 0 ? TH : TW)) {
            atomicMin(&myax[e].pa, p);
            atomicMax(&myax[e].pb, p + 1);
          }
        }
        if (flag) atomicOr(&myax[e].flags, flag);
      }
            return;
    }
    // geometry: one lane per RoI (roi_align_kernel.cu:76-98)
    if (lane < ne) {
      const Cand cr = mycand[b0 + lane];
      const int b = (int)cr.b;
      const float lo_c = axis == 0 ? cr.y1 : cr.x1, hi_c = axis == 0 ? cr.y2 : cr.x2;
      const float lo_o = axis == 0 ? cr.x1 : cr.y1, hi_o = axis == 0 ? cr.x2 : cr.y2;
      AxEnt en;
      en.start = lo_c * tc.scale;
      const float len = fmaxf(hi_c * tc.scale - en.start, 1.f);
      en.bin = len / (float)aligned;
      en.g = sr > 0 ? sr : (int)ceilf(len / (float)aligned);
      const float start_o = lo_o * tc.scale;
      const float len_o = fmaxf(hi_o * tc.scale - start_o, 1.f);
      const int aligned_o = axis == 0 ? aw : ah;
      const int g_o = sr > 0 ? sr : (int)ceilf(len_o / (float)aligned_o);
      en.flags = (b < 0 || b >= batch) ? kEntZero : 0;
      // sampling grids the tables cannot hold (only possible with an adaptive grid) make the RoI a slow one
      if (!(en.g >= 1 && en.g <= S && aligned * en.g <= S)) en.flags |= kEntNotFast;
      en.pa = 0x7fff;
      en.pb = 0;
      en.r = cr.id;
      en.pad = __float_as_int((float)en.g * (float)g_o);  // count (roi_align_kernel.cu:101)
      myax[lane] = en;
    }
    wave_sync();
    // tables: lane = (RoI, sample)
    for (int t = lane; t < ne * S; t += 64) {
      const int e = t / S, s = t - e * S;
      const AxEnt en = myax[e];
      const bool g_ok = !(en.flags & kEntNotFast);
      const int gs = g_ok ? en.g : 1;          // a slow RoI still gets its first entry: the owner test reads it
      const int ns = g_ok ? aligned * en.g : 1;
      if (s < ns) {
        const int p = s / gs, i = s - p * gs;
        // roi_align_kernel.cu:106-110
        float v = en.start + (float)p * en.bin + ((float)i + .5f) * en.bin / (float)en.g;
        if (v < -1.0f || v > (float)size) atomicOr(&myax[e].flags, (int)kEntNotFast);
        // roi_align_kernel.cu:27-52
        if (v <= 0) v = 0;
        int lo = (int)v;
        float lw, hw;
        if (lo >= size - 1) {
          lo = size - 1;  // the reference's upper tap is the same pixel; here it is the copy one row / column on
          lw = 0.f;
          hw = 1.f;
        } else {
          lw = v - (float)lo;
          hw = 1.f - lw;
        }
        // the entries of bins that belong to other tiles are never used for a result: their offset is clamped into the
        // image so that the units can compute whole rows with fixed indices
        TabEnt te;
        te.lo_rel = lo - origin;
        if (axis == 0) {
          const float count = __int_as_float(en.pad);
          te.off = min(max(te.lo_rel, 0), Cfg::kRows - 2) * kPitch * 4;
          te.hw = hw / count;
          te.lw = lw / count;
        } else {
          te.off = min(max(te.lo_rel, 0), kPitch - 2) * 4;
          te.hw = hw;
          te.lw = lw;
        }
        mytab[e * S + s] = te;
      }
    }
    wave_sync();
        // bins: lane = (RoI, bin): footprint within the halo?  which bins belong to this tile?
    for (int t = lane; t < ne * PB; t += 64) {
      const int e = t / PB, p = t - e * PB;
      const AxEnt en = myax[e];
      if (p < aligned && !(en.flags & kEntNotFast)) {
        const int lo_rel = mytab[e * S + p * en.g].lo_rel, hi_rel = mytab[e * S + p * en.g + en.g - 1].lo_rel + 1;
        if (hi_rel - lo_rel > kHalo) atomicOr(&myax[e].flags, (int)kEntNotFast);
        if (lo_rel >= 0 && lo_rel < (axis == 0 ? TH : TW)) {
          atomicMin(&myax[e].pa, p);
          atomicMax(&myax[e].pb, p + 1);
        }
      }
    }
}

// ---- one batch of a tile, all waves of the workgroup: every (RoI, bin row) unit of the descriptor block `blk` is computed
// from the image at LDS byte address img_c (this lane's channel plane) by 32 lanes = 32 channels; then the RoIs the tile
// owns but the tables cannot describe.  kNG = 32-lane groups of the workgroup. ------------------------------------------
template <int kSR, int kA, int TH, int TW, int kNG>
__device__ __forceinline__ void compute_batch(const TileCtx& tc, const typename TileCfg<kSR, kA, TH, TW>::Block* blk,
                                              unsigned img_c, const float* __restrict__ rois, float* __restrict__ out,
                                              int channels, int c0, int ah, int aw, int sr, int ablate) {
  using Cfg = TileCfg<kSR, kA, TH, TW>;
  constexpr int kPitch = Cfg::kPitch, S = Cfg::S, EB = kEB;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), cl = tid & 31;
  const TabEnt* ytab = &blk->tabs[0][0][0];
  const TabEnt* xtab = &blk->tabs[1][0][0];
  const AxEnt* axes = &blk->axes[0][0];
  const int ne = blk->misc[0];
  const int bins = ah * aw;
  const float* __restrict__ feat = tc.feat;
  const int height = tc.height, width = tc.width;
  const float spatial_scale = tc.scale;
  // ---- every wave: per-RoI state in lanes 0..15, prefix of the unit counts ----
  int e_r = 0, e_pa0 = 0, e_pw = 0, e_flags = 0, e_slow = 0, incl = 0;
  {
    int nun = 0;
    if (lane < ne) {
      const AxEnt ey = axes[lane], ex = axes[EB + lane];
      e_flags = ey.flags | ex.flags;
      e_r = ey.r;
      e_pa0 = ey.pa;
      e_pw = (ex.pa & 0xff) | (ex.pb << 8);
      if (!(e_flags & (kEntNotFast | kEntZero))) {
        if (ey.pb > ey.pa && ex.pb > ex.pa) nun = ey.pb - ey.pa;
      } else if (e_flags & kEntZero) {
        e_slow = 1;
      } else {
        // the tile that holds the first anchor computes the whole RoI
        const int ly = ytab[lane * S].lo_rel, lx = xtab[lane * S].lo_rel;
        e_slow = (ly >= 0 && ly < TH && lx >= 0 && lx < TW) ? 1 : 0;
      }
    }
    incl = nun;
#pragma unroll
    for (int d = 1; d < EB; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
  }
  const int nunits = __builtin_amdgcn_readlane(incl, EB - 1);
  const unsigned long long slow_mask = __ballot(e_slow != 0);

  // ---- units: 32 lanes = 32 channels per (RoI, bin row) ----
  for (int ub = 0; ub < nunits; ub += kNG) {
    const int u_lo = ub + 2 * wave, u_hi = u_lo + 1;  // the units of this wave's two halves
    // the RoI of a unit = number of RoIs whose units all come before it
    const int e_lo = __popcll(__ballot(lane < EB && incl <= u_lo)), e_hi = __popcll(__ballot(lane < EB && incl <= u_hi));
    const int u = lane < 32 ? u_lo : u_hi;
    const int e = min(lane < 32 ? e_lo : e_hi, EB - 1);
    const int r = __shfl(e_r, e), pa0 = __shfl(e_pa0, e), pw = __shfl(e_pw, e);
    const int before = __shfl(incl, max(e - 1, 0));
    if (u < nunits) {
      const int ph = pa0 + (u - (e > 0 ? before : 0));
      const int pwa = pw & 0xff, pwb = pw >> 8;
      float* __restrict__ dst = out + (((long long)r * channels + c0 + cl) * ah + ph) * aw;
      if constexpr (kA > 0) {
        static_assert(kSR == 2, "the unrolled path is written for 2 x 2 samples");
        const TabEnt ya = ytab[e * S + ph * 2], yb = ytab[e * S + ph * 2 + 1];
        const unsigned row_a = img_c + (unsigned)ya.off, row_b = img_c + (unsigned)yb.off;
        const v2f wya = {ya.hw, ya.lw}, wyb = {yb.hw, yb.lw};
        const TabEnt* xe = xtab + e * S;
        float acc[kA];
        if (ablate & 2) {
#pragma unroll
          for (int j = 0; j < kA; j++) acc[j] = 0.f;
        } else {
          // one bin at a time; the two x entries of the next bin are fetched before the taps of this one are used (one
          // LDS latency per bin instead of two).  Bins of the row that belong to another tile cost only that fetch.
          TabEnt xc0 = xe[0], xc1 = xe[1];
#pragma unroll
          for (int j = 0; j < kA; j++) {
            const TabEnt x0e = xc0, x1e = xc1;
            const bool mine = j >= pwa && j < pwb;
            v2f t00, t01, t10, t11, u00, u01, u10, u11;
            if (mine) {
              const unsigned a0 = row_a + (unsigned)x0e.off, a1 = row_a + (unsigned)x1e.off;
              const unsigned b0a = row_b + (unsigned)x0e.off, b1a = row_b + (unsigned)x1e.off;
              t00 = lds_rows<kPitch, 0>(a0);
              t01 = lds_rows<kPitch, 1>(a0);
              t10 = lds_rows<kPitch, 0>(a1);
              t11 = lds_rows<kPitch, 1>(a1);
              u00 = lds_rows<kPitch, 0>(b0a);
              u01 = lds_rows<kPitch, 1>(b0a);
              u10 = lds_rows<kPitch, 0>(b1a);
              u11 = lds_rows<kPitch, 1>(b1a);
            }
            if (j + 1 < kA) {
              xc0 = xe[2 * j + 2];
              xc1 = xe[2 * j + 3];
            }
            acc[j] = 0.f;
            if (mine) {
              v2f sa = t00 * x0e.hw;
              sa = __builtin_elementwise_fma(t01, splat(x0e.lw), sa);
              sa = __builtin_elementwise_fma(t10, splat(x1e.hw), sa);
              sa = __builtin_elementwise_fma(t11, splat(x1e.lw), sa);
              v2f sb = u00 * x0e.hw;
              sb = __builtin_elementwise_fma(u01, splat(x0e.lw), sb);
              sb = __builtin_elementwise_fma(u10, splat(x1e.hw), sb);
              sb = __builtin_elementwise_fma(u11, splat(x1e.lw), sb);
              v2f a2 = sa * wya;
              a2 = __builtin_elementwise_fma(sb, wyb, a2);
              acc[j] = a2.x + a2.y;
            }
          }
        }
        if (ablate & 4) {
        } else if (pwa == 0 && pwb == kA) {
          if constexpr (kA == 7) {
            *reinterpret_cast<f4u*>(dst) = f4u{acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f3u*>(dst + 4) = f3u{acc[4], acc[5], acc[6]};
          } else if constexpr (kA == 14) {
            *reinterpret_cast<f4u*>(dst) = f4u{acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f4u*>(dst + 4) = f4u{acc[4], acc[5], acc[6], acc[7]};
            *reinterpret_cast<f4u*>(dst + 8) = f4u{acc[8], acc[9], acc[10], acc[11]};
            *reinterpret_cast<f2u*>(dst + 12) = f2u{acc[12], acc[13]};
          } else {
#pragma unroll
            for (int j = 0; j < kA; j++) dst[j] = acc[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < kA; j++)
            if (j >= pwa && j < pwb) dst[j] = acc[j];
        }
      } else {
        // any pooled size / sampling grid: bin by bin
        const int gh = axes[e].g, gw = axes[EB + e].g;
        for (int pwi = pwa; pwi < pwb; pwi++) {
          v2f a2 = {0.f, 0.f};
          for (int iy = 0; iy < gh; iy++) {
            const TabEnt y = ytab[e * S + ph * gh + iy];
            v2f sy = {0.f, 0.f};
            for (int ix = 0; ix < gw; ix++) {
              const TabEnt x = xtab[e * S + pwi * gw + ix];
              const unsigned a = img_c + (unsigned)y.off + (unsigned)x.off;
              sy = __builtin_elementwise_fma(lds_rows<kPitch, 0>(a), splat(x.hw), sy);
              sy = __builtin_elementwise_fma(lds_rows<kPitch, 1>(a), splat(x.lw), sy);
            }
            a2 = __builtin_elementwise_fma(sy, (v2f){y.hw, y.lw}, a2);
          }
          dst[pwi] = a2.x + a2.y;
        }
      }
    }
  }

  // ---- RoIs this tile owns that the tables cannot describe: reference operation order from global memory ----
  for (unsigned long long todo = slow_mask; todo != 0ull; todo &= todo - 1ull) {
    const int e = (int)__builtin_ctzll(todo);
    const int r = __builtin_amdgcn_readlane(e_r, e);
    const int fl = __builtin_amdgcn_readlane(e_flags, e);
    float* __restrict__ dst = out + ((long long)r * channels + c0) * bins;
    if (fl & kEntZero) {
      for (int i = tid; i < kCt * bins; i += (kNG * 32)) dst[i] = 0.f;
      continue;
    }
    const RoiGeom g = roi_geometry(rois + (long long)r * 5, spatial_scale, ah, aw, sr);
    const float* src = feat + ((long long)g.batch_ind * channels + c0) * height * width;
    for (int i = tid; i < kCt * bins; i += (kNG * 32)) {
      const int c = i / bins, bin = i - c * bins;
      const int ph = bin / aw, pw = bin - ph * aw;
      const float* plane = src + (long long)c * height * width;
      float output_val = 0.f;
      for (int iy = 0; iy < g.grid_h; iy++) {
        const float y = sample_y(g, ph, iy);
        for (int ix = 0; ix < g.grid_w; ix++) {
          const float x = sample_x(g, pw, ix);
          const Taps t = sample_taps(height, width, y, x);
          float val = 0.f;
          if (t.y_low >= 0) {
            const float v1 = plane[t.y_low * width + t.x_low], v2 = plane[t.y_low * width + t.x_high];
            const float v3 = plane[t.y_high * width + t.x_low], v4 = plane[t.y_high * width + t.x_high];
            val = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(t.w1, v1), __fmul_rn(t.w2, v2)), __fmul_rn(t.w3, v3)),
                            __fmul_rn(t.w4, v4));
          }
          output_val = __fadd_rn(output_val, val);
        }
      }
      dst[i] = output_val / g.count;
    }
  }
}

// ---- roi_align_tiles_prepare: one 256-lane workgroup per TILE; leaves the descriptor blocks of the tile's first
// kPreBlocks * kEB scan survivors (in RoI order) in the workspace ------------------------------------------------------
constexpr int kPrepThreads = 256;
template <int kSR, int kA, int TH, int TW>
__global__ void __launch_bounds__(kPrepThreads)
roi_align_tiles_prepare(const LevelTable lv, const float* __restrict__ rois, const int* __restrict__ levels, int num_rois,
                        int batch, int ah_arg, int aw_arg, int sr_arg, int ntiles, unsigned char* __restrict__ desc) {
  using Cfg = TileCfg<kSR, kA, TH, TW>;
  using Block = typename Cfg::Block;
  const int ah = kA > 0 ? kA : ah_arg, aw = kA > 0 ? kA : aw_arg;
  const int sr = kSR > 0 ? kSR : sr_arg;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Block* blk = reinterpret_cast<Block*>(smem);
  Cand* cand = reinterpret_cast<Cand*>(smem + sizeof(Block));  // [kPreBlocks * kEB]
  int* wcount = reinterpret_cast<int*>(cand + kPreBlocks * kEB);
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const TileCtx tc = decode_tile<TH, TW>(lv, (int)blockIdx.x);
  constexpr int kCap = kPreBlocks * kEB, kW = kPrepThreads / 64;

  // scan, lane = RoI; ordinals follow the RoI index
  int total = 0;
  for (int base = 0; base < num_rois; base += kPrepThreads) {
    const int i = base + tid;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f, r4 = 0.f;
    int rl = 0;
    if (i < num_rois) {
      const float* q = rois + (long long)i * 5;
      r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3], r4 = q[4];
      if (levels != nullptr) rl = levels[i];
    }
    const bool hit = i < num_rois && tile_test<TH, TW>(tc, lv.count, batch, ntiles, i, r0, r1, r2, r3, r4, rl);
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    int off = total, all = 0;
#pragma unroll
    for (int w = 0; w < kW; w++) {
      const int cnt = wcount[w];
      off += w < wave ? cnt : 0;
      all += cnt;
    }
    const int ord = off + __popcll(m & ((1ull << lane) - 1ull));
    if (hit && ord < kCap) {
      Cand cr;
      cr.b = r0, cr.x1 = r1, cr.y1 = r2, cr.x2 = r3, cr.y2 = r4, cr.id = i, cr.pad0 = 0, cr.pad1 = 0;
      cand[ord] = cr;
    }
    total += all;
    __syncthreads();
  }
  for (int k = 0; k < kPreBlocks; k++) {
    const int ne = min(max(total - k * kEB, 0), kEB);
    if (k > 0 && ne == 0) break;  // the main kernel reads block k only when block k - 1 says "more"
    if (wave < 2) build_axis<kSR, kA, TH, TW>(tc, wave, batch, ah, aw, sr, cand, blk->axes[wave], &blk->tabs[wave][0][0], k * kEB, ne);
    if (tid == 0) {
      blk->misc[0] = ne;
      blk->misc[1] = total > (k + 1) * kEB ? 1 : 0;
      blk->misc[2] = total;
    }
    __syncthreads();
    const int4* src = reinterpret_cast<const int4*>(blk);
    int4* dst = reinterpret_cast<int4*>(desc + ((size_t)blockIdx.x * kPreBlocks + k) * sizeof(Block));
    for (int i = tid; i < (int)(sizeof(Block) / 16); i += kPrepThreads) dst[i] = src[i];
    __syncthreads();
  }
}

// kSR > 0 and kA > 0: sampling_ratio == kSR, aligned_height == aligned_width == kA at compile time.
//
// Roles: waves 2-7 issue the image DMA.  With `desc` wave 0 DMA's the tile's descriptor block beside it; without, wave 0
// builds everything that depends on the y axis and wave 1 everything that depends on the x axis.
template <int kSR, int kA, int TH, int TW>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))  // two workgroups per CU
roi_align_fwd_tiles(const LevelTable lv, const float* __restrict__ rois, const int* __restrict__ levels,
                    float* __restrict__ out, int num_rois, int batch, int channels, int ah_arg, int aw_arg,
                    int sr_arg, int ntiles, const unsigned char* __restrict__ desc, long long* __restrict__ timeline,
                    int ablate_arg) {
  const int ablate = MI_ABLATE(ablate_arg);  // tuning builds only: 1 no image DMA, 2 no tap reads / arithmetic, 4 no stores
  // tuning aid (tools/timeline_tiles.py): clock stamps of lane 0 of every workgroup, null in normal operation
  const auto stamp = [&](int k) {
    if (timeline != nullptr && threadIdx.x == 0) timeline[(long long)blockIdx.x * 8 + k] = (long long)clock64();
  };
  stamp(0);
  if (ablate & 8) return;  // tuning builds: launch and dispatch only
  using Cfg = TileCfg<kSR, kA, TH, TW>;
  using Block = typename Cfg::Block;
  constexpr int kPitch = Cfg::kPitch, kPlane = Cfg::kPlane, S = Cfg::S, EB = kEB;
  const int ah = kA > 0 ? kA : ah_arg, aw = kA > 0 ? kA : aw_arg;
  const int sr = kSR > 0 ? kSR : sr_arg;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* img = reinterpret_cast<float*>(smem);
  Block* blk = reinterpret_cast<Block*>(smem + Cfg::kImgBytes);
  Cand* cands = reinterpret_cast<Cand*>(smem + Cfg::kImgBytes + sizeof(Block));  // [axis][kCandCap]: a copy per axis wave
  int* ctr = reinterpret_cast<int*>(cands + 2 * kCandCap);                       // [0] survivors, [1] waves done scanning
  int* misc = blk->misc;
  TabEnt* tabs = &blk->tabs[0][0][0];
  AxEnt* axes = &blk->axes[0][0];

  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int ncg = channels / kCt;
  const int tile_global = uniform((int)blockIdx.x / ncg);
  const int cg = (int)blockIdx.x - tile_global * ncg;
  const TileCtx tc = decode_tile<TH, TW>(lv, tile_global);
  const float* __restrict__ feat = tc.feat;
  const int height = tc.height, width = tc.width, n = tc.n, x0 = tc.x0, y0 = tc.y0;
  const int c0 = cg * kCt;
  const unsigned plane_bytes = (unsigned)height * (unsigned)width * 4u;
  const bool pre = desc != nullptr;

  const int axis = wave & 1;  // meaningful in waves 0 (y) and 1 (x)
  Cand* mycand = cands + axis * kCandCap;

  if (!pre) {
    // ---- fast scan: every lane tests the RoIs tid, tid + 512, ... (fetched together, BEFORE the DMA is issued: issued
    // first, the image delays the RoIs of the DMA waves -- measured 50 us against 40) and appends its survivors to both
    // axis lists through an LDS counter.  The order of the list is arbitrary; it only matters when the list overflows,
    // and then the axis waves rescan in order.
    if (tid == 0) {
      ctr[0] = 0;
      ctr[1] = 0;
    }
    __syncthreads();
    constexpr int kPer = 2;
    for (int base = 0; base < num_rois; base += kThreads * kPer) {
      float rv[kPer][5];
      int rl[kPer];
#pragma unroll
      for (int k = 0; k < kPer; k++) {
        const int i = base + k * kThreads + tid;
        rl[k] = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) rv[k][j] = 0.f;
        if (i < num_rois) {
#pragma unroll
          for (int j = 0; j < 5; j++) rv[k][j] = rois[(long long)i * 5 + j];
          if (levels != nullptr) rl[k] = levels[i];
        }
      }
#pragma unroll
      for (int k = 0; k < kPer; k++) {
        const int i = base + k * kThreads + tid;
        const bool hit = i < num_rois && tile_test<TH, TW>(tc, lv.count, batch, ntiles, i, rv[k][0], rv[k][1], rv[k][2],
                                                            rv[k][3], rv[k][4], rl[k]);
        const unsigned long long m = __ballot(hit);
        if (m != 0ull) {
          int slot = 0;
          if (lane == 0) slot = atomicAdd(&ctr[0], __popcll(m));
          slot = __builtin_amdgcn_readfirstlane(slot) + __popcll(m & ((1ull << lane) - 1ull));
          if (hit && slot < kCandCap) {
            Cand cr;
            cr.b = rv[k][0], cr.x1 = rv[k][1], cr.y1 = rv[k][2], cr.x2 = rv[k][3], cr.y2 = rv[k][4], cr.id = i, cr.pad0 = 0,
            cr.pad1 = 0;
            cands[slot] = cr;
            cands[kCandCap + slot] = cr;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicAdd(&ctr[1], 1);
  }
  if (wave >= 2) {  // after the scan: in the DMA waves the RoIs must not queue behind the image (in-order vmcnt)
  // ---- tile image: [channel][row][kPitch] by LDS-DMA, lanes flattened over (row, column), clamped to the map:
  // rows / columns past it repeat the last row / column, so that "upper tap = lower tap + 1" holds for a sample
  // clamped to the border too (the reference reads the border pixel twice, weights 1 and 0) ----
  const float* slab = feat + ((long long)n * channels + c0) * height * width;
  const srd_t srd = make_srd(slab, (unsigned)kCt * plane_bytes);  // the range check includes the scalar offset
  const unsigned img_lds = lds_addr_uniform(img);
  // instruction j = (piece, channel): the six DMA waves take j = wave - 2, wave + 4, ...
  // Interior tiles of maps with 16-byte aligned rows move 16 bytes per lane (a quarter of the requests); a tile that
  // reaches past the last column needs the per-pixel clamp of the dword form.
  const bool vec4 = (width & 3) == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0 && x0 + kPitch <= width &&
                    (kPitch & 3) == 0 && (TW & 3) == 0;
  if (vec4) {
    constexpr int kCPR = kPitch / 4, kChunks = Cfg::kRows * kCPR, kP4 = (kChunks + 63) / 64;
    for (int k = 0; k < ((ablate & 1) ? 0 : kP4); k++) {
      const int q = k * 64 + lane;
      const int row = q / kCPR, j = q - row * kCPR;
      const unsigned voff = (unsigned)(min(y0 + row, height - 1) * width + x0 + 4 * j) * 4u;
      if (q < kChunks) {
        for (int c = (wave - 2 + (kNWaves - 2) * 64 - k * kCt) % (kNWaves - 2); c < kCt; c += kNWaves - 2)
          dma_dwordx4(srd, img_lds + (unsigned)(c * kPlane + k * 256) * 4u, voff, (unsigned)c * plane_bytes);
      }
    }
  } else {
    for (int k = 0; k < ((ablate & 1) ? 0 : Cfg::kPieces); k++) {
      const int p = k * 64 + lane;
      const int row = p / kPitch, col = p - row * kPitch;
      const unsigned voff = (unsigned)(min(y0 + row, height - 1) * width + min(x0 + col, width - 1)) * 4u;
      if (p < Cfg::kPx) {
        for (int c = (wave - 2 + (kNWaves - 2) * 64 - k * kCt) % (kNWaves - 2); c < kCt; c += kNWaves - 2)
          dma_dword(srd, img_lds + (unsigned)(c * kPlane + k * 64) * 4u, voff, (unsigned)c * plane_bytes);
      }
    }
  }
  }
  stamp(1);

  const int cl = tid & 31;
  const unsigned img_c = (unsigned)(uintptr_t)(lds_cfloat_t)(img + cl * kPlane);  // LDS byte address of this lane's plane

  // Batches of <= 16 survivors.  With descriptors: blocks 0 .. kPreBlocks-1 come from the workspace; survivors past them
  // (a tile under a pile of RoIs) are scanned and built here, in RoI order, like everything without descriptors.
  int win_lo = 0, b0 = 0, total = 0, ncand = 0, nblk = 0;  // waves 0 and 1
  for (bool first = true;; first = false) {
    if (wave < 2) {
      if (pre && nblk < kPreBlocks) {
        if (wave == 0) {
          const unsigned char* src = desc + ((size_t)tile_global * kPreBlocks + nblk) * sizeof(Block);
          const srd_t bsrd = make_srd(src, (unsigned)sizeof(Block));
          const unsigned dst_lds = lds_addr_uniform(blk);
          for (int k = 0; k * 1024 < (int)sizeof(Block); k++)
            if (k * 1024 + lane * 16 < (int)sizeof(Block)) dma_dwordx4(bsrd, dst_lds + (unsigned)k * 1024u, (unsigned)(k * 1024 + lane * 16), 0u);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        nblk++;
        if (nblk == kPreBlocks) {  // what follows the blocks starts at survivor kPreBlocks * kEB
          win_lo = kPreBlocks * EB - kCandCap;
          ncand = 0;
          b0 = 0;
        }
      } else {
        if (pre && total == 0) total = misc[2];  // (set by the last block; > kPreBlocks * kEB or we would not be here)
        if (!pre && first) {
          while (__atomic_load_n(&ctr[1], __ATOMIC_RELAXED) < kNWaves) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          total = __atomic_load_n(&ctr[0], __ATOMIC_RELAXED);
          if (total > kCandCap)  // overflow: windows of the list in RoI order
            total = scan_ordered<TH, TW>(tc, lv.count, batch, ntiles, rois, levels, num_rois, 0, kCandCap, mycand, nullptr);
          ncand = min(total, kCandCap);
        } else {
          b0 += EB;
          if (b0 >= ncand) {
            win_lo += kCandCap;
            total = scan_ordered<TH, TW>(tc, lv.count, batch, ntiles, rois, levels, num_rois, win_lo, kCandCap, mycand, nullptr);
            ncand = min(total - win_lo, kCandCap);
            b0 = 0;
          }
        }
        const int ne = min(EB, ncand - b0);
        build_axis<kSR, kA, TH, TW>(tc, axis, batch, ah, aw, sr, mycand, axes + axis * EB, tabs + axis * EB * S, b0, ne);
        if (tid == 0) {
          misc[0] = ne;
          misc[1] = (b0 + EB < ncand || win_lo + kCandCap < total) ? 1 : 0;
          misc[2] = total;
        }
      }
      if (first) stamp(3);
    } else if (first) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();  // tables built / fetched, image landed
    if (first) stamp(4);
    if (ablate & 16) break;  // tuning builds: nothing after the barrier
    const bool more = misc[1] != 0;
    if (first && timeline != nullptr && tid == 0) timeline[(long long)blockIdx.x * 8 + 7] = misc[2];

    compute_batch<kSR, kA, TH, TW, kGroups>(tc, blk, img_c, rois, out, channels, c0, ah, aw, sr, ablate);
    if (first) stamp(5);
    if (!more) break;
    __syncthreads();  // tables and entries are reused by the next batch
  }
  stamp(6);
}

// ---- roi_align_fwd_tiles_stream: the pooling kernel of the workspace path as a persistent, double-buffered pipeline.
// A 78 KB tile image lets only two 512-lane workgroups share a CU, and their phases (fetch, barrier, compute) cannot fill
// each other's gaps: no unit is more than half busy.  Here ONE 1024-lane workgroup per CU walks its items = (tile, channel
// group), channel group = blockIdx % ncg (one XCD's L2 keeps seeing the same channel slabs): while item k is computed out
// of one LDS image + descriptor block, all waves have the DMA of item k + 1 (image and block) in flight into the other;
// one barrier per item.  Descriptor blocks come from roi_align_tiles_prepare. -------------------------------------------
long long* g_tiles_timeline = nullptr;
constexpr int kStreamThreads = 1024;
constexpr int kStreamCandCap = 16;
template <int kSR, int kA, int TH, int TW>
__global__ void __launch_bounds__(kStreamThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
roi_align_fwd_tiles_stream(const LevelTable lv, const float* __restrict__ rois, const int* __restrict__ levels,
                           float* __restrict__ out, int num_rois, int batch, int channels, int ah_arg, int aw_arg,
                           int sr_arg, int ntiles, const unsigned char* __restrict__ desc, int ablate_arg,
                           long long* __restrict__ timeline) {
  const int ablate = MI_ABLATE(ablate_arg);
  // tuning aid (tools/timeline_tiles.py STREAM=1): clock stamps of lane 0 around the workgroup's THIRD item
  int item_no = 0;
  const auto stamp = [&](int k) {
    if (timeline != nullptr && threadIdx.x == 0 && item_no == 2) timeline[(long long)blockIdx.x * 8 + k] = (long long)clock64();
  };
  using Cfg = TileCfg<kSR, kA, TH, TW>;
  using Block = typename Cfg::Block;
  constexpr int kPitch = Cfg::kPitch, kPlane = Cfg::kPlane, EB = kEB;
  constexpr int kNW = kStreamThreads / 64, kNG = kStreamThreads / 32;
  constexpr int kCap = kStreamCandCap;  // survivors per pass of the fallback scan
  const int ah = kA > 0 ? kA : ah_arg, aw = kA > 0 ? kA : aw_arg;
  const int sr = kSR > 0 ? kSR : sr_arg;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [image 0][image 1][block 0][block 1][survivor lists of the in-kernel fallback]
  float* img0 = reinterpret_cast<float*>(smem);
  Block* blk0 = reinterpret_cast<Block*>(smem + 2 * Cfg::kImgBytes);
  Cand* cands = reinterpret_cast<Cand*>(smem + 2 * Cfg::kImgBytes + 2 * sizeof(Block));  // [axis][kCap]

  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), cl = tid & 31;
  const int ncg = channels / kCt;
  const int cg = (int)blockIdx.x % ncg;
  const int tile_first = (int)blockIdx.x / ncg, tile_step = (int)gridDim.x / ncg;
  const int c0 = cg * kCt;

  // all waves: the DMA of one item into buffer `buf`: instruction j = (piece, channel) of the image for j = wave, wave + 16,
  // ..., and the 1 KB pieces of the descriptor block
  const auto issue = [&](int tile_global, int buf) {
    const TileCtx tc = decode_tile<TH, TW>(lv, tile_global);
    const int height = tc.height, width = tc.width;
    const unsigned plane_bytes = (unsigned)height * (unsigned)width * 4u;
    const float* slab = tc.feat + ((long long)tc.n * channels + c0) * height * width;
    const srd_t srd = make_srd(slab, (unsigned)kCt * plane_bytes);  // the range check includes the scalar offset
    const unsigned img_lds = lds_addr_uniform(img0) + (unsigned)buf * (unsigned)Cfg::kImgBytes;
    // Interior tiles of maps with 16-byte aligned rows move 16 bytes per lane (a quarter of the requests); a tile that
    // reaches past the last column needs the per-pixel clamp of the dword form.  Rows / columns past the map repeat the
    // last row / column: "upper tap = lower tap + 1" then also holds for a sample clamped to the border.
    const bool vec4 = (width & 3) == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0 && tc.x0 + kPitch <= width &&
                      (kPitch & 3) == 0 && (TW & 3) == 0;
    if (ablate & 1) {
    } else if (vec4) {
      constexpr int kCPR = kPitch / 4, kChunks = Cfg::kRows * kCPR, kP4 = (kChunks + 63) / 64;
#pragma unroll
      for (int k = 0; k < kP4; k++) {
        const int q = k * 64 + lane;
        const int row = q / kCPR, j = q - row * kCPR;
        const unsigned voff = (unsigned)(min(tc.y0 + row, height - 1) * width + tc.x0 + 4 * j) * 4u;
        if (q < kChunks) {
          for (int c = (wave + kNW * 64 - k * kCt) % kNW; c < kCt; c += kNW)
            dma_dwordx4(srd, img_lds + (unsigned)(c * kPlane + k * 256) * 4u, voff, (unsigned)c * plane_bytes);
        }
      }
    } else {
      for (int k = 0; k < Cfg::kPieces; k++) {
        const int p = k * 64 + lane;
        const int row = p / kPitch, col = p - row * kPitch;
        const unsigned voff = (unsigned)(min(tc.y0 + row, height - 1) * width + min(tc.x0 + col, width - 1)) * 4u;
        if (p < Cfg::kPx) {
          for (int c = (wave + kNW * 64 - k * kCt) % kNW; c < kCt; c += kNW)
            dma_dword(srd, img_lds + (unsigned)(c * kPlane + k * 64) * 4u, voff, (unsigned)c * plane_bytes);
        }
      }
    }
    const unsigned char* src = desc + (size_t)tile_global * kPreBlocks * sizeof(Block);
    const srd_t bsrd = make_srd(src, (unsigned)sizeof(Block));
    const unsigned dst_lds = lds_addr_uniform(blk0) + (unsigned)buf * (unsigned)sizeof(Block);
    for (int k = wave; k * 1024 < (int)sizeof(Block); k += kNW)
      if (k * 1024 + lane * 16 < (int)sizeof(Block))
        dma_dwordx4(bsrd, dst_lds + (unsigned)k * 1024u, (unsigned)(k * 1024 + lane * 16), 0u);
  };

  if (tile_first >= ntiles) return;
  issue(tile_first, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tile_first + tile_step < ntiles) issue(tile_first + tile_step, 1);
  int buf = 0, tile_global = tile_first;
  TileCtx tc = decode_tile<TH, TW>(lv, tile_global);
  // state of a tile with more than 16 scan survivors (waves 0 and 1 use the last three)
  const int axis = wave & 1;
  Cand* mycand = cands + axis * kCap;
  int nblk = 1, win_lo = kPreBlocks * EB - kCap, b0 = 0, ncand = 0;
  for (;;) {  // one iteration per batch of <= 16 RoIs; all but a few tiles have one batch
    Block* blk = blk0 + buf;
    const unsigned img_c = (unsigned)(uintptr_t)(lds_cfloat_t)(img0 + (size_t)buf * (Cfg::kImgBytes / 4) + cl * kPlane);
    stamp(1);
    compute_batch<kSR, kA, TH, TW, kNG>(tc, blk, img_c, rois, out, channels, c0, ah, aw, sr, ablate);
    stamp(2);
    if (timeline != nullptr && tid == 0 && item_no == 2) timeline[(long long)blockIdx.x * 8 + 7] = blk->misc[2];
    if (blk->misc[1] != 0) {
      // ---- further batches of this tile (the second one precomputed, the rest built here in RoI order by waves 0 and 1,
      // as in roi_align_fwd_tiles) go through the same LDS block, synchronously ----
      const int total = blk->misc[2];
      __syncthreads();  // everybody is done with the block
      if (nblk < kPreBlocks) {
        if (wave == 0) {
          const unsigned char* src = desc + ((size_t)tile_global * kPreBlocks + nblk) * sizeof(Block);
          const srd_t bsrd = make_srd(src, (unsigned)sizeof(Block));
          const unsigned dst_lds = lds_addr_uniform(blk);
          for (int k = 0; k * 1024 < (int)sizeof(Block); k++)
            if (k * 1024 + lane * 16 < (int)sizeof(Block))
              dma_dwordx4(bsrd, dst_lds + (unsigned)k * 1024u, (unsigned)(k * 1024 + lane * 16), 0u);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        nblk++;
      } else if (wave < 2) {
        b0 += EB;
        if (b0 >= ncand) {
          win_lo += kCap;
          scan_ordered<TH, TW>(tc, lv.count, batch, ntiles, rois, levels, num_rois, win_lo, kCap, mycand, nullptr);
          ncand = min(total - win_lo, kCap);
          b0 = 0;
        }
        const int ne = min(EB, ncand - b0);
        build_axis<kSR, kA, TH, TW>(tc, axis, batch, ah, aw, sr, mycand, blk->axes[axis], &blk->tabs[axis][0][0], b0, ne);
        if (tid == 0) {
          blk->misc[0] = ne;
          blk->misc[1] = (b0 + EB < ncand || win_lo + kCap < total) ? 1 : 0;
          blk->misc[2] = total;
        }
      }
      __syncthreads();
      continue;
    }
    // ---- next item ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // it has landed
    stamp(3);
    __syncthreads();                                  // and everybody is done with this one
    stamp(4);
    item_no++;
    stamp(0);
    tile_global += tile_step;
    if (tile_global >= ntiles) break;
    buf ^= 1;
    tc = decode_tile<TH, TW>(lv, tile_global);
    nblk = 1, win_lo = kPreBlocks * EB - kCap, b0 = 0, ncand = 0;
    if (tile_global + tile_step < ntiles) issue(tile_global + tile_step, buf ^ 1);
  }
}

constexpr int kTH = 12, kTW = 28;

template <int kSR, int kA>
int launch_one(const LevelTable& lv, const float* rois, const int* levels, float* out, int num_rois, int batch,
               int channels, int ah, int aw, int sr, int ntiles, unsigned char* desc, hipStream_t stream) {
  using Cfg = TileCfg<kSR, kA, kTH, kTW>;
  static const bool attr = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&roi_align_fwd_tiles<kSR, kA, kTH, kTW>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::kLdsBytes) == hipSuccess;
  }();
  (void)attr;
  if (desc != nullptr) {
    roi_align_tiles_prepare<kSR, kA, kTH, kTW><<<ntiles, kPrepThreads, Cfg::kPrepLdsBytes, stream>>>(
        lv, rois, levels, num_rois, batch, ah, aw, sr, ntiles, desc);
    const int rc = check_launch("roi_align_tiles_prepare");
    if (rc != MI_OK) return rc;
  }
  constexpr size_t kStreamLds =
      2 * Cfg::kImgBytes + 2 * sizeof(typename Cfg::Block) + 2 * kStreamCandCap * sizeof(Cand);
  if constexpr (kStreamLds <= 160 * 1024) {  // two images and two blocks must fit one CU
    if (desc != nullptr && !tuning().tiles_no_stream) {
      static const bool attr2 = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&roi_align_fwd_tiles_stream<kSR, kA, kTH, kTW>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStreamLds) == hipSuccess;
      }();
      (void)attr2;
      // one workgroup per CU (256), a whole number of tiles per pass so that blockIdx % ncg is the channel group
      const int ncg = channels / kCt;
      const int tiles_per_pass = ncg >= 256 ? 1 : (256 / ncg < ntiles ? 256 / ncg : ntiles);
      roi_align_fwd_tiles_stream<kSR, kA, kTH, kTW><<<tiles_per_pass * ncg, kStreamThreads, kStreamLds, stream>>>(
          lv, rois, levels, out, num_rois, batch, channels, ah, aw, sr, ntiles, desc, tuning().ablate, g_tiles_timeline);
      return check_launch("roi_align_fwd_tiles_stream");
    }
  }
  roi_align_fwd_tiles<kSR, kA, kTH, kTW><<<ntiles * (channels / kCt), kThreads, Cfg::kLdsBytes, stream>>>(
      lv, rois, levels, out, num_rois, batch, channels, ah, aw, sr, ntiles, desc, g_tiles_timeline, tuning().ablate);
  return check_launch("roi_align_fwd_tiles");
}

int count_tiles(LevelTable& lv, int batch) {
  lv.tile_base[0] = 0;
  for (int l = 0; l < lv.count; l++)
    lv.tile_base[l + 1] =
        lv.tile_base[l] + ((lv.width[l] + kTW - 1) / kTW) * ((lv.height[l] + kTH - 1) / kTH) * batch;
  return lv.tile_base[lv.count];
}

size_t block_bytes(int aligned_height, int aligned_width, int sampling_ratio) {
  if (sampling_ratio == 2 && aligned_height == 7 && aligned_width == 7) return sizeof(DescBlock<14>);
  if (sampling_ratio == 2 && aligned_height == 14 && aligned_width == 14) return sizeof(DescBlock<28>);
  return sizeof(DescBlock<kMaxSamples>);
}

}  // namespace

void roi_align_fwd_tiles_set_timeline(long long* device_buffer) { g_tiles_timeline = device_buffer; }

bool roi_align_fwd_tiles_supported(int channels, int height, int width, int aligned_height, int aligned_width) {
  return channels > 0 && channels % kCt == 0 && height > 0 && width > 0 && aligned_height > 0 && aligned_width > 0 &&
         aligned_height <= kMaxSamples && aligned_width <= kMaxSamples &&
         (long long)kCt * height * width * 4 < (1LL << 31);
}

// bytes of scratch with which the forward runs as roi_align_tiles_prepare + roi_align_fwd_tiles
size_t roi_align_fwd_tiles_workspace_bytes(LevelTable lv, int batch, int aligned_height, int aligned_width,
                                           int sampling_ratio) {
  return (size_t)count_tiles(lv, batch) * kPreBlocks * block_bytes(aligned_height, aligned_width, sampling_ratio);
}

// workspace: nullptr (or too small) = one launch, every workgroup builds its own tables
int launch_roi_align_fwd_tiles_levels(LevelTable lv, const float* rois, const int* levels, float* output, int batch,
                                      int channels, int num_rois, int aligned_height, int aligned_width,
                                      int sampling_ratio, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  const int ntiles = count_tiles(lv, batch);
  if ((long long)ntiles * (channels / kCt) >= (1LL << 31)) {
    set_error("roi_align_fwd_tiles: grid too large");
    return MI_ERR_BAD_ARGUMENT;
  }
  unsigned char* desc = static_cast<unsigned char*>(workspace);
  if (desc != nullptr && (workspace_bytes < (size_t)ntiles * kPreBlocks * block_bytes(aligned_height, aligned_width, sampling_ratio) ||
                          (reinterpret_cast<uintptr_t>(desc) & 15) != 0 || tuning().no_ws))
    desc = nullptr;
  if (sampling_ratio == 2 && aligned_height == 7 && aligned_width == 7)
    return launch_one<2, 7>(lv, rois, levels, output, num_rois, batch, channels, 7, 7, 2, ntiles, desc, stream);
  if (sampling_ratio == 2 && aligned_height == 14 && aligned_width == 14)
    return launch_one<2, 14>(lv, rois, levels, output, num_rois, batch, channels, 14, 14, 2, ntiles, desc, stream);
  return launch_one<0, 0>(lv, rois, levels, output, num_rois, batch, channels, aligned_height, aligned_width,
                          sampling_ratio, ntiles, desc, stream);
}

int launch_roi_align_fwd_tiles(const float* features, const float* rois, float* output, int batch, int channels,
                               int height, int width, int num_rois, int aligned_height, int aligned_width,
                               float spatial_scale, int sampling_ratio, void* workspace, size_t workspace_bytes,
                               hipStream_t stream) {
  return launch_roi_align_fwd_tiles_levels(single_level(features, nullptr, batch, height, width, spatial_scale), rois,
                                           nullptr, output, batch, channels, num_rois, aligned_height, aligned_width,
                                           sampling_ratio, workspace, workspace_bytes, stream);
}

}  // namespace mi
