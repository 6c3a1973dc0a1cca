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
constexpr int kPreload = 2;     // RoIs per lane fetched in front of the DMA

struct Cand {  // 32 bytes
  float b, x1, y1, x2, y2;
  int id, pad0, pad1;
};
struct TabEnt {  // one axis sample: LDS byte offset of its lower tap (clamped into the image), weights (the y weights
  int off;       // already divided by count), lower tap relative to the tile origin in pixels (unclamped)
  float hw, lw;
  int lo_rel;
};
struct Ent {  // one scan survivor
  float start[2], bin[2];
  int g[2], pa[2], pb[2];
  int r, flags, unit_base, nunits, slow, pad;
};
struct Unit {  // one (RoI, bin row)
  int e, ph, r, pw;  // pw = pwa | pwb << 8
};
enum : int { kEntNotFast = 1, kEntZero = 2 };

template <int kSR, int kA, int TH, int TW>
struct TileCfg {
  static constexpr int kRows = TH + kHalo, kPitch = TW + kHalo;
  static constexpr int kPx = kRows * kPitch;
  static constexpr int kPlane = kPx | 1;  // odd: 32 planes -> 32 banks
  static constexpr int kPieces = (kPx + 63) / 64;
  static constexpr int S = (kA > 0 && kSR > 0) ? kA * kSR : kMaxSamples;  // table entries per axis and RoI
  static constexpr int EB = S <= 16 ? 16 : 8;                            // RoIs per batch
  static constexpr int PB = kA > 0 ? kA : kMaxSamples;                    // bins per axis the lane mappings provide for
  static constexpr int kMaxUnits = EB * PB;
  static constexpr size_t kImgBytes = (size_t)kCt * kPlane * 4;
  static constexpr size_t kTabBytes = (size_t)EB * S * 2 * sizeof(TabEnt);
  static constexpr size_t kLdsBytes =
      kImgBytes + kTabBytes + kCandCap * sizeof(Cand) + EB * sizeof(Ent) + kMaxUnits * sizeof(Unit) + 64 * 4;
  static_assert(kPitch + 1 < 256, "ds_read2_b32 offsets");
  static_assert(EB * 2 * S <= kThreads && EB * 2 * PB <= kThreads && EB * PB <= kThreads, "one pass per phase");
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

// kSR > 0 and kA > 0: sampling_ratio == kSR, aligned_height == aligned_width == kA at compile time.
template <int kSR, int kA, int TH, int TW>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))  // two workgroups per CU
roi_align_fwd_tiles(const LevelTable lv, const float* __restrict__ rois, const int* __restrict__ levels,
                    float* __restrict__ out, int num_rois, int batch, int channels, int ah_arg, int aw_arg,
                    int sr_arg, int ntiles, long long* __restrict__ timeline) {
  // tuning aid (tools/timeline_tiles.py): clock stamps of lane 0 of every workgroup, null in normal operation
  const auto stamp = [&](int k) {
    if (timeline != nullptr && threadIdx.x == 0) timeline[(long long)blockIdx.x * 8 + k] = (long long)clock64();
  };
  stamp(0);
  using Cfg = TileCfg<kSR, kA, TH, TW>;
  constexpr int kPitch = Cfg::kPitch, kPlane = Cfg::kPlane, S = Cfg::S, EB = Cfg::EB, PB = Cfg::PB;
  const int ah = kA > 0 ? kA : ah_arg, aw = kA > 0 ? kA : aw_arg;
  const int sr = kSR > 0 ? kSR : sr_arg;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* img = reinterpret_cast<float*>(smem);
  TabEnt* ytab = reinterpret_cast<TabEnt*>(smem + Cfg::kImgBytes);  // [EB][S]
  TabEnt* xtab = ytab + EB * S;                                     // [EB][S]
  Cand* cand = reinterpret_cast<Cand*>(xtab + EB * S);
  Ent* ents = reinterpret_cast<Ent*>(cand + kCandCap);
  Unit* units = reinterpret_cast<Unit*>(ents + EB);
  int* misc = reinterpret_cast<int*>(units + Cfg::kMaxUnits);  // [0..7] wave counts, [8] units of the batch

  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  // ---- this workgroup's tile ----
  // (integer divisions run on the vector unit: readfirstlane brings the wave-uniform results back to SGPRs, otherwise
  // everything derived from them occupies vector registers)
  const int ncg = channels / kCt;
  const int tile_global = uniform((int)blockIdx.x / ncg);
  const int cg = (int)blockIdx.x - tile_global * ncg;
  int tile = tile_global, lvl = 0;
  while (lvl + 1 < lv.count && tile >= lv.tile_base[lvl + 1]) lvl++;
  tile -= lv.tile_base[lvl];
  const float* __restrict__ feat = lv.feat[lvl];
  const int height = lv.height[lvl], width = lv.width[lvl];
  const float spatial_scale = lv.scale[lvl];
  const int tiles_x = (width + TW - 1) / TW, tiles_y = (height + TH - 1) / TH;
  const int n = uniform(tile / (tiles_x * tiles_y));
  const int trem = tile - n * tiles_x * tiles_y;
  const int tyi = uniform(trem / tiles_x), txi = trem - tyi * tiles_x;
  const int x0 = txi * TW, y0 = tyi * TH;
  const int c0 = cg * kCt;
  const int bins = ah * aw;
  const unsigned plane_bytes = (unsigned)height * (unsigned)width * 4u;

  // ---- the RoIs this lane will test first: fetched in front of the DMA ----
  float pre[kPreload][5];
  int prel[kPreload];
#pragma unroll
  for (int k = 0; k < kPreload; k++) {
    const int i = tid + k * kThreads;
    prel[k] = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) pre[k][j] = 0.f;
    if (i < num_rois) {
#pragma unroll
      for (int j = 0; j < 5; j++) pre[k][j] = rois[(long long)i * 5 + j];
      if (levels != nullptr) prel[k] = levels[i];
    }
  }
#pragma unroll
  for (int k = 0; k < kPreload; k++) {  // the values must have arrived before the DMA is issued (in-order vmcnt)
#pragma unroll
    for (int j = 0; j < 5; j++) asm volatile("" ::"v"(pre[k][j]));
    asm volatile("" ::"v"(prel[k]));
  }
  stamp(1);

  // ---- tile image: [channel][row][kPitch] by LDS-DMA, lanes flattened over (row, column), clamped to the map ----
  {
    const float* slab = feat + ((long long)n * channels + c0) * height * width;
    const srd_t srd = make_srd(slab, (unsigned)kCt * plane_bytes);  // the range check includes the scalar offset
    const unsigned img_lds = lds_addr_uniform(img);
    for (int k = wave; k < Cfg::kPieces; k += kNWaves) {
      const int p = k * 64 + lane;
      const int row = p / kPitch, col = p - row * kPitch;
      const unsigned voff = (unsigned)(min(y0 + row, height - 1) * width + min(x0 + col, width - 1)) * 4u;
      if (p < Cfg::kPx) {
#pragma unroll
        for (int c = 0; c < kCt; c++)
          dma_dword(srd, img_lds + (unsigned)(c * kPlane + k * 64) * 4u, voff, (unsigned)c * plane_bytes);
      }
    }
  }
  stamp(2);

  const int grp = tid >> 5, cl = tid & 31;
  const unsigned img_c = (unsigned)(uintptr_t)(lds_cfloat_t)(img + cl * kPlane);  // LDS byte address of this lane's plane

  // one RoI against this tile, multiplications and compares only: true for every RoI that has a bin here or that this
  // tile owns (a superset; the tables decide).  A bin's anchor is the lower tap of its first sample, whose coordinate
  // lies in [start, start + length]; taps are clamped to the map.
  const float tile_y_lo = (float)y0, tile_y_hi = (float)(y0 + TH), tile_x_lo = (float)x0, tile_x_hi = (float)(x0 + TW);
  const float h_max = (float)(height - 1), w_max = (float)(width - 1);
  const auto test = [&](int i, float rb, float rx1, float ry1, float rx2, float ry2, int rl) -> bool {
    const int b = (int)rb;
    if (b < 0 || b >= batch) return (i % ntiles) == tile_global;  // zeros, written by tile (i mod tiles)
    const int l = min(max(rl, 0), lv.count - 1);
    if (l != lvl || b != n) return false;
    const float sy = ry1 * spatial_scale, sx = rx1 * spatial_scale;
    const float ey = sy + fmaxf(ry2 * spatial_scale - sy, 1.f), ex = sx + fmaxf(rx2 * spatial_scale - sx, 1.f);
    const float ya = fminf(fmaxf(sy, 0.f), h_max) - 1.f, yb = fminf(fmaxf(ey, 0.f), h_max) + 1.f;
    const float xa = fminf(fmaxf(sx, 0.f), w_max) - 1.f, xb = fminf(fmaxf(ex, 0.f), w_max) + 1.f;
    return ya < tile_y_hi && yb >= tile_y_lo && xa < tile_x_hi && xb >= tile_x_lo;
  };

  // ---- scan: survivors with ordinal in [win_lo, win_lo + kCandCap) go to LDS (ordinals follow the RoI index);
  // returns the number of survivors of the tile ----
  // (first: the RoIs fetched in front of the DMA are used; later passes -- tiles with more than kCandCap survivors --
  // read them again, so that the registers are free while the units run)
  const auto scan = [&](int win_lo, auto first_pass) -> int {
    constexpr bool kFirst = decltype(first_pass)::value;
    int total = 0;
    for (int base = 0; base < num_rois; base += kThreads) {
      const int i = base + tid;
      float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f, r4 = 0.f;
      int rl = 0;
      if (kFirst && base == 0) {
        r0 = pre[0][0], r1 = pre[0][1], r2 = pre[0][2], r3 = pre[0][3], r4 = pre[0][4], rl = prel[0];
      } else if (kFirst && kPreload > 1 && base == kThreads) {
        r0 = pre[kPreload - 1][0], r1 = pre[kPreload - 1][1], r2 = pre[kPreload - 1][2], r3 = pre[kPreload - 1][3],
        r4 = pre[kPreload - 1][4], rl = prel[kPreload - 1];
      } else if (i < num_rois) {
        const float* q = rois + (long long)i * 5;
        r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3], r4 = q[4];
        if (levels != nullptr) rl = levels[i];
      }
      const bool hit = i < num_rois && test(i, r0, r1, r2, r3, r4, rl);
      const unsigned long long m = __ballot(hit);
      if (lane == 0) misc[wave] = __popcll(m);
      __syncthreads();
      int off = total, all = 0;
#pragma unroll
      for (int w = 0; w < kNWaves; w++) {
        const int cnt = misc[w];
        off += w < wave ? cnt : 0;
        all += cnt;
      }
      const int ord = off + __popcll(m & ((1ull << lane) - 1ull)) - win_lo;
      if (hit && ord >= 0 && ord < kCandCap) {
        Cand cr;
        cr.b = r0, cr.x1 = r1, cr.y1 = r2, cr.x2 = r3, cr.y2 = r4, cr.id = i, cr.pad0 = 0, cr.pad1 = 0;
        cand[ord] = cr;
      }
      total += all;
      __syncthreads();
    }
    return total;
  };
  int total = scan(0, std::true_type());
  stamp(3);
  if (timeline != nullptr && tid == 0) timeline[(long long)blockIdx.x * 8 + 7] = total;

  bool image_ready = false;
  for (int win_lo = 0;; win_lo += kCandCap) {
    if (win_lo > 0) total = scan(win_lo, std::false_type());
    const int ncand = min(total - win_lo, kCandCap);

    for (int b0 = 0; b0 < ncand; b0 += EB) {
      const int ne = min(EB, ncand - b0);
      // ---- geometry: one lane per RoI (roi_align_kernel.cu:76-98) ----
      if (tid < ne) {
        const Cand cr = cand[b0 + tid];
        const int b = (int)cr.b;
        Ent en;
        en.start[0] = cr.y1 * spatial_scale;
        en.start[1] = cr.x1 * spatial_scale;
        const float len_h = fmaxf(cr.y2 * spatial_scale - en.start[0], 1.f);
        const float len_w = fmaxf(cr.x2 * spatial_scale - en.start[1], 1.f);
        en.bin[0] = len_h / (float)ah;
        en.bin[1] = len_w / (float)aw;
        en.g[0] = sr > 0 ? sr : (int)ceilf(len_h / (float)ah);
        en.g[1] = sr > 0 ? sr : (int)ceilf(len_w / (float)aw);
        en.pa[0] = en.pa[1] = 0x7fff;
        en.pb[0] = en.pb[1] = 0;
        en.r = cr.id;
        en.flags = (b < 0 || b >= batch) ? kEntZero : 0;
        // sampling grids the tables cannot hold (only possible with an adaptive grid) make the RoI a slow one
        if (!(en.g[0] >= 1 && en.g[0] <= S && ah * en.g[0] <= S && en.g[1] >= 1 && en.g[1] <= S && aw * en.g[1] <= S))
          en.flags |= kEntNotFast;
        en.unit_base = en.nunits = en.slow = en.pad = 0;
        ents[tid] = en;
      }
      __syncthreads();
      // ---- tables: lane = (RoI, axis, sample); axis 0 = y ----
      {
        const int e = tid / (2 * S), rem = tid - e * (2 * S);
        const int axis = rem / S, s = rem - axis * S;
        if (e < ne) {
          const Ent& en = ents[e];
          const int g = en.g[axis], aligned = axis == 0 ? ah : aw, size = axis == 0 ? height : width;
          const bool g_ok = !(en.flags & kEntNotFast);
          const int gs = g_ok ? g : 1;          // a slow RoI still gets its first entry: the owner test reads it
          const int ns = g_ok ? aligned * g : 1;
          if (s < ns) {
            const int p = s / gs, i = s - p * gs;
            // roi_align_kernel.cu:106-110
            float v = en.start[axis] + (float)p * en.bin[axis] + ((float)i + .5f) * en.bin[axis] / (float)g;
            if (v < -1.0f || v > (float)size) atomicOr(&ents[e].flags, (int)kEntNotFast);
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
            // the entries of bins that belong to other tiles are never used for a result: their offset is clamped into
            // the image so that the units can compute whole rows with fixed indices
            TabEnt t;
            if (axis == 0) {
              const float count = (float)en.g[0] * (float)en.g[1];
              t.lo_rel = lo - y0;
              t.off = min(max(t.lo_rel, 0), Cfg::kRows - 2) * kPitch * 4;
              t.hw = hw / count;
              t.lw = lw / count;
              ytab[e * S + s] = t;
            } else {
              t.lo_rel = lo - x0;
              t.off = min(max(t.lo_rel, 0), kPitch - 2) * 4;
              t.hw = hw;
              t.lw = lw;
              xtab[e * S + s] = t;
            }
          }
        }
      }
      __syncthreads();
      // ---- bins: lane = (RoI, axis, bin): footprint within the halo?  which bins belong to this tile? ----
      {
        const int e = tid / (2 * PB), rem = tid - e * (2 * PB);
        const int axis = rem / PB, p = rem - axis * PB;
        if (e < ne && p < (axis == 0 ? ah : aw) && !(ents[e].flags & kEntNotFast)) {
          const int g = ents[e].g[axis];
          const TabEnt* tab = axis == 0 ? ytab : xtab;
          const int lo_rel = tab[e * S + p * g].lo_rel, hi_rel = tab[e * S + p * g + g - 1].lo_rel + 1;
          if (hi_rel - lo_rel > kHalo) atomicOr(&ents[e].flags, (int)kEntNotFast);
          if (lo_rel >= 0 && lo_rel < (axis == 0 ? TH : TW)) {
            atomicMin(&ents[e].pa[axis], p);
            atomicMax(&ents[e].pb[axis], p + 1);
          }
        }
      }
      __syncthreads();
      // ---- units per RoI, prefix ----
      if (tid < 64) {
        int nun = 0;
        if (tid < ne) {
          const Ent& en = ents[tid];
          if (!(en.flags & (kEntNotFast | kEntZero))) {
            if (en.pb[0] > en.pa[0] && en.pb[1] > en.pa[1]) nun = en.pb[0] - en.pa[0];
          } else if (en.flags & kEntZero) {
            ents[tid].slow = 1;
          } else {
            // the tile that holds the first anchor computes the whole RoI
            const int ly = ytab[tid * S].lo_rel, lx = xtab[tid * S].lo_rel;
            ents[tid].slow = (ly >= 0 && ly < TH && lx >= 0 && lx < TW) ? 1 : 0;
          }
        }
        int incl = nun;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(incl, d);
          if (lane >= d) incl += o;
        }
        if (tid < ne) {
          ents[tid].nunits = nun;
          ents[tid].unit_base = incl - nun;
        }
        if (tid == 63) misc[8] = incl;
      }
      __syncthreads();
      {
        // unit descriptors: lane = (RoI, row of the RoI inside this tile)
        const int ue = tid / PB, uk = tid - ue * PB;
        if (ue < ne) {
          const Ent& en = ents[ue];
          if (uk < en.nunits) {
            Unit u;
            u.e = ue, u.ph = en.pa[0] + uk, u.r = en.r, u.pw = en.pa[1] | (en.pb[1] << 8);
            units[en.unit_base + uk] = u;
          }
        }
      }
      if (!image_ready) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        image_ready = true;
      }
      __syncthreads();

      // ---- units: 32 lanes = 32 channels per (RoI, bin row) ----
      const int nunits = misc[8];
      if (win_lo == 0 && b0 == 0) stamp(4);
      for (int u = grp; u < nunits; u += kGroups) {
        const Unit un = units[u];
        const int e = un.e, ph = un.ph;
        const int pwa = un.pw & 0xff, pwb = un.pw >> 8;
        float* __restrict__ dst = out + (((long long)un.r * channels + c0 + cl) * ah + ph) * aw;
        if constexpr (kA > 0) {
          static_assert(kSR == 2, "the unrolled path is written for 2 x 2 samples");
          const TabEnt ya = ytab[e * S + ph * 2], yb = ytab[e * S + ph * 2 + 1];
          const unsigned row_a = img_c + (unsigned)ya.off, row_b = img_c + (unsigned)yb.off;
          const v2f wya = {ya.hw, ya.lw}, wyb = {yb.hw, yb.lw};
          const TabEnt* xe = xtab + e * S;
          float acc[kA];
#pragma unroll
          for (int pw0 = 0; pw0 < kA; pw0 += 2) {
            constexpr int kNB = 2;
            const int nb = kA - pw0 < kNB ? kA - pw0 : kNB;
            if (pw0 < pwb && pw0 + nb > pwa) {
              v2f t[kNB][2][2][2];  // [bin][iy][ix][x tap] = {row y, row y + 1}
              float hx[kNB][2], lx[kNB][2];
#pragma unroll
              for (int j = 0; j < kNB; j++) {
                if (j < nb) {
#pragma unroll
                  for (int ix = 0; ix < 2; ix++) {
                    const TabEnt x = xe[(pw0 + j) * 2 + ix];
                    hx[j][ix] = x.hw;
                    lx[j][ix] = x.lw;
                    const unsigned aa = row_a + (unsigned)x.off, ab = row_b + (unsigned)x.off;
                    t[j][0][ix][0] = lds_rows<kPitch, 0>(aa);
                    t[j][0][ix][1] = lds_rows<kPitch, 1>(aa);
                    t[j][1][ix][0] = lds_rows<kPitch, 0>(ab);
                    t[j][1][ix][1] = lds_rows<kPitch, 1>(ab);
                  }
                }
              }
#pragma unroll
              for (int j = 0; j < kNB; j++) {
                if (j < nb) {
                  v2f sa = t[j][0][0][0] * hx[j][0];
                  sa = __builtin_elementwise_fma(t[j][0][0][1], splat(lx[j][0]), sa);
                  sa = __builtin_elementwise_fma(t[j][0][1][0], splat(hx[j][1]), sa);
                  sa = __builtin_elementwise_fma(t[j][0][1][1], splat(lx[j][1]), sa);
                  v2f sb = t[j][1][0][0] * hx[j][0];
                  sb = __builtin_elementwise_fma(t[j][1][0][1], splat(lx[j][0]), sb);
                  sb = __builtin_elementwise_fma(t[j][1][1][0], splat(hx[j][1]), sb);
                  sb = __builtin_elementwise_fma(t[j][1][1][1], splat(lx[j][1]), sb);
                  v2f a2 = sa * wya;
                  a2 = __builtin_elementwise_fma(sb, wyb, a2);
                  acc[pw0 + j] = a2.x + a2.y;
                }
              }
            } else {
#pragma unroll
              for (int j = 0; j < kNB; j++)
                if (j < nb) acc[pw0 + j] = 0.f;
            }
          }
          if (pwa == 0 && pwb == kA) {
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
          const int gh = ents[e].g[0], gw = ents[e].g[1];
          for (int pw = pwa; pw < pwb; pw++) {
            v2f a2 = {0.f, 0.f};
            for (int iy = 0; iy < gh; iy++) {
              const TabEnt y = ytab[e * S + ph * gh + iy];
              v2f sy = {0.f, 0.f};
              for (int ix = 0; ix < gw; ix++) {
                const TabEnt x = xtab[e * S + pw * gw + ix];
                const unsigned a = img_c + (unsigned)y.off + (unsigned)x.off;
                sy = __builtin_elementwise_fma(lds_rows<kPitch, 0>(a), splat(x.hw), sy);
                sy = __builtin_elementwise_fma(lds_rows<kPitch, 1>(a), splat(x.lw), sy);
              }
              a2 = __builtin_elementwise_fma(sy, (v2f){y.hw, y.lw}, a2);
            }
            dst[pw] = a2.x + a2.y;
          }
        }
      }
      if (win_lo == 0 && b0 == 0) stamp(5);

      // ---- RoIs this tile owns that the tables cannot describe: reference operation order from global memory ----
      for (int e = 0; e < ne; e++) {
        if (!ents[e].slow) continue;
        const int r = ents[e].r;
        float* __restrict__ dst = out + ((long long)r * channels + c0) * bins;
        if (ents[e].flags & kEntZero) {
          for (int i = tid; i < kCt * bins; i += kThreads) dst[i] = 0.f;
          continue;
        }
        const RoiGeom g = roi_geometry(rois + (long long)r * 5, spatial_scale, ah, aw, sr);
        const float* src = feat + ((long long)g.batch_ind * channels + c0) * height * width;
        for (int i = tid; i < kCt * bins; i += kThreads) {
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
      __syncthreads();  // tables and entries are reused by the next batch
    }
    if (win_lo + kCandCap >= total) break;
  }
  // a workgroup without a single survivor must not retire while its DMA is still writing LDS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(6);
}

constexpr int kTH = 12, kTW = 28;
long long* g_tiles_timeline = nullptr;

template <int kSR, int kA>
int launch_one(const LevelTable& lv, const float* rois, const int* levels, float* out, int num_rois, int batch,
               int channels, int ah, int aw, int sr, int ntiles, hipStream_t stream) {
  using Cfg = TileCfg<kSR, kA, kTH, kTW>;
  static const bool attr = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&roi_align_fwd_tiles<kSR, kA, kTH, kTW>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::kLdsBytes) == hipSuccess;
  }();
  (void)attr;
  roi_align_fwd_tiles<kSR, kA, kTH, kTW><<<ntiles * (channels / kCt), kThreads, Cfg::kLdsBytes, stream>>>(
      lv, rois, levels, out, num_rois, batch, channels, ah, aw, sr, ntiles, g_tiles_timeline);
  return check_launch("roi_align_fwd_tiles");
}

}  // namespace

void roi_align_fwd_tiles_set_timeline(long long* device_buffer) { g_tiles_timeline = device_buffer; }

bool roi_align_fwd_tiles_supported(int channels, int height, int width, int aligned_height, int aligned_width) {
  return channels > 0 && channels % kCt == 0 && height > 0 && width > 0 && aligned_height > 0 && aligned_width > 0 &&
         aligned_height <= kMaxSamples && aligned_width <= kMaxSamples &&
         (long long)kCt * height * width * 4 < (1LL << 31);
}

int launch_roi_align_fwd_tiles_levels(LevelTable lv, const float* rois, const int* levels, float* output, int batch,
                                      int channels, int num_rois, int aligned_height, int aligned_width,
                                      int sampling_ratio, hipStream_t stream) {
  lv.tile_base[0] = 0;
  for (int l = 0; l < lv.count; l++)
    lv.tile_base[l + 1] =
        lv.tile_base[l] + ((lv.width[l] + kTW - 1) / kTW) * ((lv.height[l] + kTH - 1) / kTH) * batch;
  const int ntiles = lv.tile_base[lv.count];
  if ((long long)ntiles * (channels / kCt) >= (1LL << 31)) {
    set_error("roi_align_fwd_tiles: grid too large");
    return MI_ERR_BAD_ARGUMENT;
  }
  if (sampling_ratio == 2 && aligned_height == 7 && aligned_width == 7)
    return launch_one<2, 7>(lv, rois, levels, output, num_rois, batch, channels, 7, 7, 2, ntiles, stream);
  if (sampling_ratio == 2 && aligned_height == 14 && aligned_width == 14)
    return launch_one<2, 14>(lv, rois, levels, output, num_rois, batch, channels, 14, 14, 2, ntiles, stream);
  return launch_one<0, 0>(lv, rois, levels, output, num_rois, batch, channels, aligned_height, aligned_width,
                          sampling_ratio, ntiles, stream);
}

int launch_roi_align_fwd_tiles(const float* features, const float* rois, float* output, int batch, int channels,
                               int height, int width, int num_rois, int aligned_height, int aligned_width,
                               float spatial_scale, int sampling_ratio, hipStream_t stream) {
  return launch_roi_align_fwd_tiles_levels(single_level(features, nullptr, batch, height, width, spatial_scale), rois,
                                           nullptr, output, batch, channels, num_rois, aligned_height, aligned_width,
                                           sampling_ratio, stream);
}

}  // namespace mi
