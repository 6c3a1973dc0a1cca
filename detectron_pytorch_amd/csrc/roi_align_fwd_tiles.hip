// roi_align_fwd_tiles.hip -- RoIAlign forward (Caffe2 semantics, roi_align_kernel.cu:16-121), NCHW, ONE launch, no
// workspace: the TILE-CENTRIC forward for gfx950.
//
// Why.  A per-RoI gather (roi_align_fwd_records) moves every RoI's window into LDS on its own: on the config-2 input
// the windows overlap 2.2x and a ~70-byte row segment drags in 1.5 cache lines, so 430 MB pass from L2 to the L1s for
// 60 MB of distinct pixels, and a separate launch has to sort the RoIs along a sweep to keep even that local.  Here the
// roles are swapped: a workgroup owns a TH x TW tile of ONE feature map for 16 channels, loads it (plus a 4-pixel halo
// to the right and below) ONCE with coalesced LDS-DMA, and computes every output bin whose first sample's lower tap
// lies in the tile -- whichever RoI it belongs to.  Every feature byte is fetched once per channel group (+ halo),
// neighbouring tiles run side by side, and nothing has to be sorted, ranked or prepared.
//
//   workgroup = (tile, 16-channel group), 512 lanes.  blockIdx % ncg = channel group: with round-robin dispatch an
//       XCD's L2 only ever sees "its" channel slabs.
//   1. every lane fetches the RoIs it will test (lane = RoI), THEN the tile DMA is issued (the loads are ordered in
//      front of the DMA so that waiting for them does not wait for the image), THEN the RoIs are tested:
//      level / image match and [first, last] bin anchor against the tile rectangle -- exact, because the assignment
//      of bins to tiles is separable.  Hits are compacted into LDS with their five floats.
//   2. per batch of <= 16 (8) hits: axis tables in LDS (lane = (hit, axis, sample): the reference's fp32 operations
//      for coordinate, taps and weights, roi_align_kernel.cu:74-110,16-52), then per (hit, axis, bin): footprint
//      check and the range [pa, pb) of bins that belong to this tile.
//   3. work unit = (hit, bin row).  A unit is computed by 32 lanes = 16 channels x {lower-x tap, upper-x tap}: with
//      channel planes of stride = 2 (mod 32) words the 16 + 16 lanes of a ds_read_b32 group hit 32 distinct banks
//      whatever the bin (the two halves differ in the parity of the column, the channels in the rest); each lane
//      weights its taps (hx or lx), the two halves are added with one v_permlane16_swap.  A bin row leaves as a
//      16-byte + 12-byte store per channel (7 bins) -- every output element is written exactly once, by one tile.
//   RoIs the tables cannot describe (a sample outside the [-1, size] band, more samples per axis than the tables
//   hold, a bin whose taps span more than the halo) are computed by the tile that owns their first anchor with the
//   reference's operation order straight from global memory (bit-exact); RoIs of a non-existent image get zeros from
//   tile (index mod tiles).
//
// Arithmetic: taps and weights exactly the reference's; the sum is evaluated as
//   sum_iy (hy/count) * (sum_ix w * F[ylo][.]) + (ly/count) * (sum_ix w * F[yhi][.])  with FMAs, per tap half,
// i.e. fp32 rounding differences only (measured ~5e-7 on unit-variance data; contract 1e-4).  A clamped border sample
// reads (size-1, size-1) with weights (1, 0) as the reference does, so non-finite features behave identically.
#include "common.h"
#include "lds_dma.h"
#include "roi_align_device.h"

namespace mi {
namespace {

constexpr int kCt = 16;         // channels per workgroup
constexpr int kThreads = 512;
constexpr int kNWaves = kThreads / 64;
constexpr int kGroups = kThreads / 32;  // units in flight
constexpr int kHalo = 4;        // rows below / columns right of the tile that a bin's taps may reach
constexpr int kCandCap = 64;    // hits held in LDS per scan pass
constexpr int kMaxSamples = 32; // samples per axis the generic tables hold
constexpr int kPreload = 2;     // RoIs per lane fetched in front of the DMA

constexpr int plane_words(int px) { return ((px - 2 + 31) / 32) * 32 + 2; }  // smallest >= px that is 2 (mod 32)

struct Cand {  // 32 bytes
  float b, x1, y1, x2, y2;
  int id, pad0, pad1;
};
struct YEnt {  // byte offsets of the two tap rows relative to the tile origin, weights already divided by count
  int off_lo, off_hi;
  float hw, lw;
};
struct XEnt {  // one tap: byte offset of the column relative to the tile origin, weight
  int off;
  float w;
};
struct Ent {
  int r, flags, g[2], pa[2], pb[2], unit_base, nunits, slow, pad;
};
enum : int { kEntNotFast = 1, kEntZero = 2 };

struct Axis {
  float start, bin;
  int g;
};
// roi_align_kernel.cu:79-98
__device__ __forceinline__ Axis axis_geo(float c_lo, float c_hi, float scale, int aligned, int sampling_ratio) {
  Axis a;
  a.start = c_lo * scale;
  const float len = fmaxf(c_hi * scale - a.start, 1.f);
  a.bin = len / (float)aligned;
  a.g = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(len / (float)aligned);
  return a;
}
// roi_align_kernel.cu:106-110
__device__ __forceinline__ float sample_coord(const Axis& a, int p, int i) {
  return a.start + (float)p * a.bin + ((float)i + .5f) * a.bin / (float)a.g;
}
// lower tap of an in-band sample (roi_align_kernel.cu:27-44); also the anchor that assigns a bin to a tile
__device__ __forceinline__ int tap_low(float v, int size) {
  if (v <= 0) v = 0;
  int low = (int)v;
  if (low >= size - 1) low = size - 1;
  return low;
}

template <int kSR, int kA, int TH, int TW>
struct TileCfg {
  static constexpr int kRows = TH + kHalo, kPitch = TW + kHalo;
  static constexpr int kPx = kRows * kPitch;
  static constexpr int kPlane = plane_words(kPx);
  static constexpr int kPieces = (kPx + 63) / 64;
  static constexpr int S = (kA > 0 && kSR > 0) ? kA * kSR : kMaxSamples;  // table entries per axis and hit
  static constexpr int EB = S <= 16 ? 16 : 8;                            // hits per batch
  static constexpr size_t kImgBytes = (size_t)kCt * kPlane * 4;
  static constexpr size_t kTabBytes = (size_t)EB * S * (sizeof(YEnt) + 2 * sizeof(XEnt));
  static constexpr size_t kLdsBytes = kImgBytes + kTabBytes + kCandCap * sizeof(Cand) + EB * sizeof(Ent) + 64 * 4;
  static_assert(kPitch % 2 == 0, "the bank scheme needs an even pitch");
  static_assert(EB * 2 * S <= kThreads, "one table pass");
};

__device__ __forceinline__ float lds_f32(unsigned byte_addr) {
  return *reinterpret_cast<lds_cfloat_t>((uintptr_t)byte_addr);
}
// sum of the two 16-lane halves of every 32-lane group, in all lanes
__device__ __forceinline__ float add_halves(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

// kSR > 0 and kA > 0: sampling_ratio == kSR, aligned_height == aligned_width == kA at compile time.
template <int kSR, int kA, int TH, int TW>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))  // two workgroups per CU
roi_align_fwd_tiles(const LevelTable lv, const float* __restrict__ rois, const int* __restrict__ levels,
                    float* __restrict__ out, int num_rois, int batch, int channels, int ah_arg, int aw_arg,
                    int sr_arg, int ntiles) {
  using Cfg = TileCfg<kSR, kA, TH, TW>;
  constexpr int kPitch = Cfg::kPitch, kPlane = Cfg::kPlane, S = Cfg::S, EB = Cfg::EB;
  const int ah = kA > 0 ? kA : ah_arg, aw = kA > 0 ? kA : aw_arg;
  const int sr = kSR > 0 ? kSR : sr_arg;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* img = reinterpret_cast<float*>(smem);
  YEnt* ytab = reinterpret_cast<YEnt*>(smem + Cfg::kImgBytes);
  XEnt* xtab = reinterpret_cast<XEnt*>(ytab + EB * S);                  // [EB][S][2]
  Cand* cand = reinterpret_cast<Cand*>(xtab + EB * S * 2);
  Ent* ents = reinterpret_cast<Ent*>(cand + kCandCap);
  int* misc = reinterpret_cast<int*>(ents + EB);  // [0..7] wave counts, [8] units of the batch

  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  // ---- this workgroup's tile ----
  const int ncg = channels / kCt;
  const int cg = blockIdx.x % ncg;
  const int tile_global = blockIdx.x / ncg;
  int tile = tile_global, lvl = 0;
  while (lvl + 1 < lv.count && tile >= lv.tile_base[lvl + 1]) lvl++;
  tile -= lv.tile_base[lvl];
  const float* __restrict__ feat = lv.feat[lvl];
  const int height = lv.height[lvl], width = lv.width[lvl];
  const float spatial_scale = lv.scale[lvl];
  const int tiles_x = (width + TW - 1) / TW, tiles_y = (height + TH - 1) / TH;
  const int n = tile / (tiles_x * tiles_y);
  const int trem = tile - n * tiles_x * tiles_y;
  const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
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
  for (int k = 0; k < kPreload; k++) {  // the values must have arrived before anything below is issued
#pragma unroll
    for (int j = 0; j < 5; j++) asm volatile("" ::"v"(pre[k][j]));
    asm volatile("" ::"v"(prel[k]));
  }

  // ---- tile image: [channel][row][kPitch] by LDS-DMA, lanes flattened over (row, column) ----
  {
    const float* slab = feat + ((long long)n * channels + c0) * height * width;
    const srd_t srd = make_srd(slab, (unsigned)kCt * plane_bytes);  // the range check includes the scalar offset
    const unsigned img_lds = lds_addr_uniform(img);
    for (int k = wave; k < Cfg::kPieces; k += kNWaves) {
      const int p = k * 64 + lane;
      const int row = p / kPitch, col = p - row * kPitch;
      const bool ok = p < Cfg::kPx && y0 + row < height && x0 + col < width;
      const unsigned voff = (unsigned)((y0 + row) * width + x0 + col) * 4u;
      if (ok) {
#pragma unroll
        for (int c = 0; c < kCt; c++)
          dma_dword(srd, img_lds + (unsigned)(c * kPlane + k * 64) * 4u, voff, (unsigned)c * plane_bytes);
      }
    }
  }

  const int grp = tid >> 5, half = (tid >> 4) & 1, cl = tid & 15;
  const unsigned img_c = (unsigned)(uintptr_t)(lds_cfloat_t)(img + cl * kPlane);  // LDS byte address of this lane's plane
  bool image_ready = false;

  // one RoI against this tile: a fast candidate (a bin anchor inside the tile) or a RoI this tile owns
  const auto test = [&](int i, float rb, float rx1, float ry1, float rx2, float ry2, int rl) -> bool {
    const int b = (int)rb;
    if (b < 0 || b >= batch) return (i % ntiles) == tile_global;  // zeros, written by tile (i mod tiles)
    const int l = min(max(rl, 0), lv.count - 1);
    if (l != lvl || b != n) return false;
    const Axis ay = axis_geo(ry1, ry2, spatial_scale, ah, sr), ax = axis_geo(rx1, rx2, spatial_scale, aw, sr);
    const int ay0 = tap_low(sample_coord(ay, 0, 0), height), ay1 = tap_low(sample_coord(ay, ah - 1, 0), height);
    const int ax0 = tap_low(sample_coord(ax, 0, 0), width), ax1 = tap_low(sample_coord(ax, aw - 1, 0), width);
    return ay0 < y0 + TH && ay1 >= y0 && ax0 < x0 + TW && ax1 >= x0;
  };

  for (int win_lo = 0;; win_lo += kCandCap) {
    // ---- scan: hits with ordinal in [win_lo, win_lo + kCandCap) go to LDS (ordinals follow the RoI index) ----
    int total = 0;
    for (int base = 0; base < num_rois; base += kThreads) {
      const int i = base + tid;
      float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f, r4 = 0.f;
      int rl = 0;
      if (base == 0) {
        r0 = pre[0][0], r1 = pre[0][1], r2 = pre[0][2], r3 = pre[0][3], r4 = pre[0][4], rl = prel[0];
      } else if (kPreload > 1 && base == kThreads) {
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
    const int ncand = min(total - win_lo, kCandCap);

    for (int b0 = 0; b0 < ncand; b0 += EB) {
      const int ne = min(EB, ncand - b0);
      if (tid < EB) {
        Ent e0;
        e0.r = 0, e0.flags = 0, e0.g[0] = 1, e0.g[1] = 1, e0.pa[0] = 0x7fff, e0.pa[1] = 0x7fff, e0.pb[0] = 0,
        e0.pb[1] = 0, e0.unit_base = 0, e0.nunits = 0, e0.slow = 0, e0.pad = 0;
        ents[tid] = e0;
      }
      __syncthreads();
      // ---- tables: lane = (hit, axis, sample); axis 0 = y ----
      {
        const int e = tid / (2 * S), rem = tid - e * (2 * S);
        const int axis = rem / S, s = rem - axis * S;
        if (e < ne) {
          const Cand cr = cand[b0 + e];
          const int b = (int)cr.b;
          const Axis ay = axis_geo(cr.y1, cr.y2, spatial_scale, ah, sr), ax = axis_geo(cr.x1, cr.x2, spatial_scale, aw, sr);
          const Axis a = axis == 0 ? ay : ax;
          const int aligned = axis == 0 ? ah : aw, size = axis == 0 ? height : width;
          // sampling grids the tables cannot hold (only possible with an adaptive grid) make the RoI a slow one; its
          // first table entry still exists, because the owner test reads it
          const bool g_ok = a.g >= 1 && a.g <= S && aligned * a.g <= S;
          const int gs = g_ok ? a.g : 1;
          const int ns = g_ok ? aligned * a.g : 1;
          if (s == 0) {
            ents[e].g[axis] = gs;
            if (axis == 0) ents[e].r = cr.id;
            int f = 0;
            if (b < 0 || b >= batch) f |= kEntZero;
            if (!g_ok) f |= kEntNotFast;
            if (f) atomicOr(&ents[e].flags, f);
          }
          if (s < ns) {
            const int p = s / gs;
            float v = sample_coord(a, p, s - p * gs);
            if (v < -1.0f || v > (float)size) atomicOr(&ents[e].flags, (int)kEntNotFast);
            // roi_align_kernel.cu:27-52
            if (v <= 0) v = 0;
            int lo = (int)v, hi;
            float lw, hw;
            if (lo >= size - 1) {
              hi = lo = size - 1;
              lw = 0.f;
              hw = 1.f;
            } else {
              hi = lo + 1;
              lw = v - (float)lo;
              hw = 1.f - lw;
            }
            if (axis == 0) {
              const float count = (float)ay.g * (float)ax.g;
              YEnt t;
              t.off_lo = (lo - y0) * kPitch * 4;
              t.off_hi = (hi - y0) * kPitch * 4;
              t.hw = hw / count;
              t.lw = lw / count;
              ytab[e * S + s] = t;
            } else {
              XEnt t0, t1;
              t0.off = (lo - x0) * 4, t0.w = hw;
              t1.off = (hi - x0) * 4, t1.w = lw;
              xtab[(e * S + s) * 2 + 0] = t0;
              xtab[(e * S + s) * 2 + 1] = t1;
            }
          }
        }
      }
      __syncthreads();
      // ---- bins: lane = (hit, axis, bin): footprint within the halo?  which bins belong to this tile? ----
      {
        constexpr int PB = kA > 0 ? kA : kMaxSamples;  // bins per axis the mapping provides for
        const int e = tid / (2 * PB), rem = tid - e * (2 * PB);
        const int axis = rem / PB, p = rem - axis * PB;
        if (e < ne && p < (axis == 0 ? ah : aw)) {
          const int g = ents[e].g[axis];
          const int first = p * g, last = first + g - 1;
          if (g >= 1 && last < S) {
            int lo_rel, hi_rel;
            if (axis == 0) {
              lo_rel = ytab[e * S + first].off_lo / (kPitch * 4);
              hi_rel = ytab[e * S + last].off_hi / (kPitch * 4);
            } else {
              lo_rel = xtab[(e * S + first) * 2].off / 4;
              hi_rel = xtab[(e * S + last) * 2 + 1].off / 4;
            }
            if (hi_rel - lo_rel > kHalo) atomicOr(&ents[e].flags, (int)kEntNotFast);
            if (lo_rel >= 0 && lo_rel < (axis == 0 ? TH : TW)) {
              atomicMin(&ents[e].pa[axis], p);
              atomicMax(&ents[e].pb[axis], p + 1);
            }
          }
        }
      }
      __syncthreads();
      // ---- units per hit, prefix ----
      if (tid < 64) {
        int nun = 0;
        if (tid < ne) {
          const Ent en = ents[tid];
          if (!(en.flags & (kEntNotFast | kEntZero))) {
            if (en.pb[0] > en.pa[0] && en.pb[1] > en.pa[1]) nun = en.pb[0] - en.pa[0];
          } else if (en.flags & kEntZero) {
            ents[tid].slow = 1;
          } else {
            // the tile that holds the first anchor computes the whole RoI
            const int ly = ytab[tid * S].off_lo / (kPitch * 4), lx = xtab[tid * S * 2].off / 4;
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
      if (!image_ready) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        image_ready = true;
      }
      __syncthreads();

      // ---- units: 32 lanes = 16 channels x {lower-x tap, upper-x tap} per (hit, bin row) ----
      const int nunits = misc[8];
      for (int u = grp; u < nunits; u += kGroups) {
        int e = 0;
        while (e + 1 < ne && ents[e + 1].unit_base <= u) e++;
        const Ent en = ents[e];
        const int ph = en.pa[0] + (u - en.unit_base);
        const int pwa = en.pa[1], pwb = en.pb[1];
        float* __restrict__ dst = out + (((long long)en.r * channels + c0 + cl) * ah + ph) * aw;
        if constexpr (kA > 0) {
          const YEnt ya = ytab[e * S + ph * kSR], yb = ytab[e * S + ph * kSR + (kSR > 1 ? 1 : 0)];
#pragma unroll
          for (int pw0 = 0; pw0 < kA; pw0 += 4) {
            constexpr int kNBmax = 4;
            const int nb = kA - pw0 < kNBmax ? kA - pw0 : kNBmax;
            if (pw0 < pwb && pw0 + nb > pwa) {
              float v[kNBmax][2][kSR][2];
              float w[kNBmax][kSR];
              float res[kNBmax];
#pragma unroll
              for (int j = 0; j < kNBmax; j++) {
                if (j < nb) {
                  const int pw = min(max(pw0 + j, pwa), pwb - 1);  // bins of other tiles: computed on a safe address, dropped
#pragma unroll
                  for (int ix = 0; ix < kSR; ix++) {
                    const XEnt xe = xtab[(e * S + pw * kSR + ix) * 2 + half];
                    w[j][ix] = xe.w;
                    const unsigned a = img_c + (unsigned)xe.off;
                    v[j][0][ix][0] = lds_f32(a + (unsigned)ya.off_lo);
                    v[j][0][ix][1] = lds_f32(a + (unsigned)ya.off_hi);
                    if (kSR > 1) {
                      v[j][1][ix][0] = lds_f32(a + (unsigned)yb.off_lo);
                      v[j][1][ix][1] = lds_f32(a + (unsigned)yb.off_hi);
                    }
                  }
                }
              }
#pragma unroll
              for (int j = 0; j < kNBmax; j++) {
                res[j] = 0.f;
                if (j < nb) {
                  float acc = 0.f;
#pragma unroll
                  for (int iy = 0; iy < (kSR > 1 ? 2 : 1); iy++) {
                    const YEnt& y = iy == 0 ? ya : yb;
                    float slo = w[j][0] * v[j][iy][0][0], shi = w[j][0] * v[j][iy][0][1];
#pragma unroll
                    for (int ix = 1; ix < kSR; ix++) {
                      slo = __builtin_fmaf(w[j][ix], v[j][iy][ix][0], slo);
                      shi = __builtin_fmaf(w[j][ix], v[j][iy][ix][1], shi);
                    }
                    acc = __builtin_fmaf(y.hw, slo, acc);
                    acc = __builtin_fmaf(y.lw, shi, acc);
                  }
                  res[j] = add_halves(acc);
                }
              }
              // both halves hold the sums: the batches of a row alternate between them
              if (((pw0 >> 2) & 1) == half) {
                if (pw0 >= pwa && pw0 + nb <= pwb) {
                  if (nb == 4)
                    *reinterpret_cast<f4u*>(dst + pw0) = f4u{res[0], res[1], res[2], res[3]};
                  else if (nb == 3)
                    *reinterpret_cast<f3u*>(dst + pw0) = f3u{res[0], res[1], res[2]};
                  else if (nb == 2)
                    *reinterpret_cast<f2u*>(dst + pw0) = f2u{res[0], res[1]};
                  else
                    dst[pw0] = res[0];
                } else {
#pragma unroll
                  for (int j = 0; j < kNBmax; j++)
                    if (j < nb && pw0 + j >= pwa && pw0 + j < pwb) dst[pw0 + j] = res[j];
                }
              }
            }
          }
        } else {
          // any pooled size / sampling grid: bin by bin
          const int gh = en.g[0], gw = en.g[1];
          for (int pw = pwa; pw < pwb; pw++) {
            float acc = 0.f;
            for (int iy = 0; iy < gh; iy++) {
              const YEnt y = ytab[e * S + ph * gh + iy];
              float slo = 0.f, shi = 0.f;
              for (int ix = 0; ix < gw; ix++) {
                const XEnt xe = xtab[(e * S + pw * gw + ix) * 2 + half];
                const unsigned a = img_c + (unsigned)xe.off;
                slo = __builtin_fmaf(xe.w, lds_f32(a + (unsigned)y.off_lo), slo);
                shi = __builtin_fmaf(xe.w, lds_f32(a + (unsigned)y.off_hi), shi);
              }
              acc = __builtin_fmaf(y.hw, slo, acc);
              acc = __builtin_fmaf(y.lw, shi, acc);
            }
            const float t = add_halves(acc);
            if (half == 0) dst[pw] = t;
          }
        }
      }

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
  // a workgroup without a single hit must not retire while its DMA is still writing LDS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

constexpr int kTH = 24, kTW = 32;

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
      lv, rois, levels, out, num_rois, batch, channels, ah, aw, sr, ntiles);
  return check_launch("roi_align_fwd_tiles");
}

}  // namespace

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
