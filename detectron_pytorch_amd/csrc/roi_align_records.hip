// roi_align_records.hip -- RoIAlign forward and backward (Caffe2 semantics), NCHW, the record-driven fast path for
// gfx950 (caller workspace, two launches):
//
//   roi_align_prepare   one wavefront per RoI.  Computes, with exactly the reference's fp32 operations
//       (lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu:74-110 and :16-52), everything that is identical
//       for all channels of the RoI: the window of feature rows/columns its samples touch, the tap row/column and
//       the two interpolation weights of every axis sample, and the split of its bin rows into "stages" whose
//       windows fit half of the LDS image of the main kernel.  It also RANKS the RoIs along a boustrophedon sweep
//       of the image (16-feature-row bands, alternating x direction) and stores each record at its rank, so that
//       the main kernel walks the feature map coherently: rocprofv3 TCC counters on the config-2 input (512 RoIs in
//       random order) show 235 MB of fabric reads per call in arrival order and 67 MB in sweep order, against
//       60 MB of distinct bytes -- in arrival order the gather is bound by the Infinity-Cache -> L2 rate.
//   roi_align_fwd_records   one 256-lane workgroup per (rank, 32-channel tile); tile index = blockIdx % 8, i.e. the
//       slab of the feature map a tile reads is served by one XCD's L2.  Record by scalar load -> window by LDS-DMA ->
//       vmcnt + barrier -> bins (packed FMAs) -> LDS tile -> contiguous 16-byte stores; the stages of an item are pipelined
//       (next window issued before the tile is stored).  (Removed after measurement: a variant that prefetched the next
//       window under the current arithmetic, 16-channel workgroups, roles on waves, a resident ticketed grid, and -- round 6 --
//       static pairs of ranks with the next window landing in VGPRs: a wave spends its time ISSUING the window pieces, not
//       waiting for them, and register loads stall it at the same place: +28 %, profiles/r06_forward_pair_ab.txt -- see
//       DESIGN.md section 5 and docs/history.md.)
//   roi_align_fwd_slab   the forward WITHOUT records, one launch: one wave per (RoI, 8 channels), an XCD reads one 8-channel
//       slab at a time (it fits its L2: no sweep order needed), geometry / tables / stages computed by the wave.  What every
//       caller gets whose workspace has no room for a backward (see its comment below).
//   roi_align_bwd_plan / roi_align_bwd_tiles   the backward over the same records (see below).
//
// Data movement and arithmetic of both forwards:
//   * arithmetic contract: sample coordinates, tap rows / columns and interpolation weights are computed with exactly the
//     reference's fp32 operations (roi_align_kernel.cu:74-110, :16-52), so every output reads the same feature pixels with the
//     same weights; the weighted sum is evaluated separably with fused multiply-adds -- per bin
//     0.25 * sum_iy (hy * R(y_lo) + ly * R(y_lo + 1)), R(row) = sum_ix (hx * F[row][x_lo] + lx * F[row][x_lo + 1]) -- instead of
//     the reference's 16 products added left to right: fp32 rounding only (~5e-7 on unit-variance features; contract 1e-4);
//   * window = the feature rows / columns any sample of the RoI touches, copied L2 -> LDS once by LDS-DMA
//     (buffer_load_dwordx4 ... lds: no VGPR staging, no ds_write pass), lanes flattened over (row, 16-byte group), one plane per
//     channel; tap pair (F[x], F[x + 1]) = one ds_read2_b32 (record-driven kernel: odd plane stride, lane & 31 = channel, no
//     bank conflicts; records-free kernel: see slab_plane);
//   * a clamped border sample is the pixel pair (size - 1, size) with weights (1, 0), the reference's (size - 1, size - 1) with
//     (1, 0): the window ends one row / column past the map and the DMA reads that one from the clamped address, so
//     "high = low + 1" holds for every sample (fixed +4 / +pitch tap addressing) and a non-finite border pixel propagates
//     exactly as in the reference (1 * f + 0 * f);
//   * results are staged in LDS as [channel][bin] and leave as contiguous 16-byte stores; windows larger than the LDS image
//     are processed in stages of bin rows.
// RoIs the LDS image cannot serve (a sample outside the [-1, size] band, one bin row larger than the image, > 32 samples per
// axis, H or W < 2) take the in-kernel direct path (reference operation order, bit-exact).
// (roi_align_fwd_tile.hip -- rounds 1-5's forward without a workspace: this structure per (RoI, 32 channels) in arrival
// order -- is gone: the records-free kernel serves its callers.)
#include "common.h"
#include "roi_align_device.h"
#include "lds_dma.h"
#include "roi_align_record_layout.h"
// Tuning builds (MI_TUNING_BUILD=1) only: wave 0 of every forward workgroup stamps clock64() at the phase boundaries of its
// life (tools/timeline_records.py); the release kernel has neither the parameter nor the stamps.
#if MI_TUNING
#define MI_TL_PARAM , long long* __restrict__ timeline
#define MI_TL_ARG , g_records_timeline
#define MI_STAMP(k)                                                                                                    \
  do {                                                                                                                \
    if (timeline != nullptr && threadIdx.x == 0) timeline[(long long)blockIdx.x * 8 + (k)] = (long long)clock64();     \
  } while (0)
#else
#define MI_TL_PARAM
#define MI_TL_ARG
#define MI_STAMP(k)                                                                                                    \
  do {                                                                                                                \
  } while (0)
#endif

#include <algorithm>
#include <type_traits>

namespace mi {
namespace {

constexpr int kCT = 32;                    // channels per workgroup
constexpr int kThreads = 256;
constexpr int kSlots = kThreads / 32;      // half-waves; each owns output columns pw = slot, slot + 8, ...
constexpr int kTileBins = 56;              // output bins per channel staged in LDS between two stores
constexpr int kMaxRois = 8192;             // the rank pass keeps one key per RoI in LDS
constexpr int kBandRows = 16;              // feature rows per sweep band

__device__ __forceinline__ void axis_taps(float v, int size, int& lo, float& hw, float& lw) {
  if (v <= 0) v = 0;
  int low = (int)v;
  if (low >= size - 1) {
    // reference: low = high = size - 1, l = 0, h = 1.  Encoded as the pair (size - 1, size): the consumers read "row /
    // column `size`" from the clamped address size - 1, so that the border pixel enters the sum as 1 * f + 0 * f exactly
    // as in the reference (a non-finite border pixel gives NaN there, and nothing else does)
    lo = size - 1;
    hw = 1.f;
    lw = 0.f;
  } else {
    lo = low;
    lw = v - (float)low;
    hw = 1.f - lw;
  }
}

// roi_align_kernel.cu:106-110
__device__ __forceinline__ float coord(float start, float bin, int p, int i, int grid) {
  return start + (float)p * bin + ((float)i + .5f) * bin / (float)grid;
}

// sweep key of a RoI: image, band of its centre row, x of its centre (reversed in odd bands)
__device__ __forceinline__ int level_of(const int* __restrict__ levels, int i, const LevelTable& lv) {
  return levels != nullptr ? min(max(levels[i], 0), lv.count - 1) : 0;
}

// level (2 bits) | image (6 bits) | band (8) | cost class (2) | x (14): RoIs of one level and image are contiguous in the
// sweep.  Inside a 16-row band the RoIs whose window needs several LDS stages come first (class from the RoI's size: an
// estimate, any order is correct): the forward's workgroups are handed out in rank order, a three-stage item lives three
// times as long as a small one, and what the launch ends with -- the last band -- should be its cheap items.  A band is
// ~2.3x smaller than what is resident per XCD, so the order inside it does not change what meets in L2.  Measured (one box,
// alternating): config 2 34.6 -> 33.4 us, 1024 RoIs on two images 57.6 -> 56.7; but the channels-last tap kernel, whose
// workgroups are all resident at once and live off their x-neighbours' lines, loses 0.65 us, and the backward over a step's
// clustered pyramid RoIs 1.3 us (its tile lists follow the rank order): `cost_in_band` is set by the planar forward of a
// single map only, everybody else sweeps plainly.
// (Round 5 first put the class in FRONT of the whole key: each class then sweeps the map on its own, the windows of
// neighbouring RoIs no longer meet in L2 -- config 2 35.9 -> 40.4 us, channels-last 30.5 -> 44.6.)
__device__ __forceinline__ unsigned sweep_key(const float* __restrict__ roi, int lvl, float spatial_scale, int height,
                                              int cap_px, int cost_in_band) {
  const float cy = (roi[2] + roi[4]) * 0.5f * spatial_scale, cx = (roi[1] + roi[3]) * 0.5f * spatial_scale;
  const int b = (lvl << 6) | min(max((int)roi[0], 0), 63);
  const int y = min(max((int)cy, 0), max(height - 1, 0)), band = min(y / kBandRows, 255);
  int x = min(max((int)(cx * 4.f), 0), 16383);
  if (band & 1) x = 16383 - x;
  const float rw = fmaxf((roi[3] - roi[1]) * spatial_scale, 1.f), rh = fmaxf((roi[4] - roi[2]) * spatial_scale, 1.f);
  const int px = ((min((int)rw, 4096) + 3 + 3) & ~3) * (min((int)rh, 4096) + 3);  // padded window pitch x window rows
  const unsigned cls = !cost_in_band ? 0u : px <= cap_px ? 2u : (px <= 2 * cap_px ? 1u : 0u);
  return ((unsigned)b << 24) | ((unsigned)band << 16) | (cls << 14) | (unsigned)x;
}

// Where roi_align_prepare takes its RoIs from.
//   PlainRois      the caller's [R, 5] blob and (pyramid calls) the level-table index of every RoI;
//   CollectedRois  the LAST step of the device-side proposal stage folded in (modeling/collect_and_distribute_fpn_rpn_
//                  proposals.py:101-119; mi_rpn_collect_finish without a launch of its own): row r of the RoI blob is candidate
//                  top_idx[r], its FPN level comes from utils/fpn.py:11-28 (fp32, the reference's operation order), and the
//                  owner wave of a RoI also writes the blob row, the validity byte and the level the model consumes -- the
//                  producer of the RoIs leaves their records behind, the forward over them starts with its gather kernel.
struct PlainRois {
  const float* __restrict__ rois;
  const int* __restrict__ levels;
  __device__ __forceinline__ int get(int i, int nlevels, float (&v)[5]) const {
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = rois[(long long)i * 5 + k];
    return levels != nullptr ? min(max(levels[i], 0), nlevels - 1) : 0;
  }
  __device__ __forceinline__ void emit(int, const float (&)[5]) const {}
};
struct CollectedRois {
  const float* __restrict__ top_scores;
  const long long* __restrict__ top_idx;
  const float* __restrict__ cand_rois;
  int mark_invalid, k_min, k_max;
  float s0, lvl0;
  float* __restrict__ out_rois;
  unsigned char* __restrict__ out_valid;
  int* __restrict__ out_levels;
  __device__ __forceinline__ float fpn_level(const float (&v)[5]) const {
    const float w = v[3] - v[1] + 1.f, h = v[4] - v[2] + 1.f;
    float area = w * h;
    area = area < 0.f ? 0.f : area;             // areas[neg_idx] = 0 (utils/boxes.py:113-121 via fpn.py:18)
    const float lvl = floorf(lvl0 + log2f(sqrtf(area) / s0 + 1e-6f));
    return fminf(fmaxf(lvl, (float)k_min), (float)k_max);
  }
  __device__ __forceinline__ int get(int i, int nlevels, float (&v)[5]) const {
    const float* c = cand_rois + top_idx[i] * 5;
    const bool ok = top_scores[i] > -__builtin_inff();
    v[0] = (ok || !mark_invalid) ? c[0] : -1.f;
#pragma unroll
    for (int k = 1; k < 5; k++) v[k] = c[k];
    return min(max(k_max - (int)fpn_level(v), 0), nlevels - 1);  // the level table lists the coarsest map first
  }
  __device__ __forceinline__ void emit(int i, const float (&v)[5]) const {
#pragma unroll
    for (int k = 0; k < 5; k++) out_rois[(long long)i * 5 + k] = v[k];
    out_valid[i] = top_scores[i] > -__builtin_inff() ? 1 : 0;
    out_levels[i] = (int)fpn_level(v);
  }
};

template <class Src>
__global__ void __launch_bounds__(256)
roi_align_prepare(int* __restrict__ ws, int num_rois, int batch, int aligned_height, int aligned_width, int sampling_ratio,
                  int cap_px, int stage_px, int max_rows_tile, int bwd_tables, int channels, int ablate_arg, int cost_in_band,
                  const Src src, const LevelTable lv) {  // scalars first: they arrive preloaded in SGPRs (build.py)
  const int ablate = MI_ABLATE(ablate_arg);  // tuning builds: 32 = no sweep keys / rank, 64 = no tables, 128 = no stage loop
  extern __shared__ unsigned keys[];  // [num_rois]
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < kCounterDwords; i += 256) ws[i] = 0;
  // This wave's RoI: its five floats and its level are fetched FIRST (one address for the whole wave), so that their
  // latency passes under the key phase below instead of after the barrier.
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int r_safe = __builtin_amdgcn_readfirstlane(min(r, num_rois - 1));
  float own[5];
  const int lvl = __builtin_amdgcn_readfirstlane(src.get(r_safe, lv.count, own));
  for (int i = threadIdx.x; i < num_rois && !(ablate & 32); i += 256) {
    float v[5];
    const int l = src.get(i, lv.count, v);
    keys[i] = sweep_key(v, l, lv.scale[l], lv.height[l], cap_px, cost_in_band);
  }
  const float roi_b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, own[0])));
  const float roi_x1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, own[1])));
  const float roi_y1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, own[2])));
  const float roi_x2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, own[3])));
  const float roi_y2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, own[4])));
  if (r < num_rois && lane == 0) src.emit(r, own);
  __syncthreads();
  if (r >= num_rois) return;
  // rank of this RoI in the sweep (ties by index): the record index
  int rank = 0;
  if (ablate & 32) rank = r;
  else {
    const unsigned mine = keys[r];
    for (int j = lane; j < num_rois; j += 64) {
      const unsigned k = keys[j];
      rank += (k < mine || (k == mine && j < r)) ? 1 : 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) rank += __shfl_xor(rank, d);
  }
  int* __restrict__ rec = ws + kCounterDwords + (long long)rank * kRecDwords;
  const int height = lv.height[lvl], width = lv.width[lvl];
  const float spatial_scale = lv.scale[lvl];
  const int batch_ind = (int)roi_b;
  const float start_w = roi_x1 * spatial_scale, start_h = roi_y1 * spatial_scale;
  const float roi_width = fmaxf(roi_x2 * spatial_scale - start_w, 1.f);
  const float roi_height = fmaxf(roi_y2 * spatial_scale - start_h, 1.f);
  const float bin_h = roi_height / (float)aligned_height, bin_w = roi_width / (float)aligned_width;
  const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)aligned_height);
  const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)aligned_width);
  const float count = (float)(gh * gw);
  const int nsy = aligned_height * gh, nsx = aligned_width * gw;
  int flags = 0;
  if (batch_ind < 0 || batch_ind >= batch) flags = kFlagZero;
  const float yf = coord(start_h, bin_h, 0, 0, gh), yl = coord(start_h, bin_h, aligned_height - 1, gh - 1, gh);
  const float xf = coord(start_w, bin_w, 0, 0, gw), xl = coord(start_w, bin_w, aligned_width - 1, gw - 1, gw);
  bool fast = flags == 0 && nsy <= kMaxS && nsx <= kMaxS && aligned_height <= kMaxStages && height >= 2 &&
              width >= 2 && !(yf < -1.0f || yl > (float)height || xf < -1.0f || xl > (float)width) && yl >= yf &&
              xl >= xf;
  int wy0 = 0, wy1 = 1, wx0 = 0, wx1 = 1;
  if (fast) {
    float a, b;
    axis_taps(yf, height, wy0, a, b);
    axis_taps(yl, height, wy1, a, b);
    axis_taps(xf, width, wx0, a, b);
    axis_taps(xl, width, wx1, a, b);
    wy1 += 1;
    wx1 += 1;
  }
  const int ww = wx1 - wx0 + 1;
  const bool tabs = fast;  // the axis tables below are valid whatever the LDS stages decide
  // this lane's y sample (lane < nsy) and x sample (lane < nsx)
  int ylo = wy0, xlo = wx0;
  float yhw = 0.f, ylw = 0.f, xhw = 0.f, xlw = 0.f;  // this lane's sample: weights of (lo, lo + 1); y: divided by count
  if (fast && !(ablate & 64)) {
    float hw = 0.f, lw = 0.f;
    if (lane < nsy) {
      const int ph = lane / gh;
      axis_taps(coord(start_h, bin_h, ph, lane - ph * gh, gh), height, ylo, hw, lw);
      ylo = min(max(ylo, wy0), wy1 - 1);
      yhw = hw / count;
      ylw = lw / count;
      int4 e;
      e.x = ylo * ((ww + 3) & ~3) * 4;  // the NCHW forward lays a window row out on a pitch of whole 16-byte groups
      e.y = __float_as_int(yhw);
      e.z = __float_as_int(ylw);
      e.w = ylo;
      reinterpret_cast<int4*>(rec + kRecY)[lane] = e;
    }
    if (lane < nsx) {
      const int pw = lane / gw;
      axis_taps(coord(start_w, bin_w, pw, lane - pw * gw, gw), width, xlo, hw, lw);
      xlo = min(max(xlo, wx0), wx1 - 1);
      xhw = hw;
      xlw = lw;
      int4 e;
      e.x = (xlo - wx0) * 4;
      e.y = __float_as_int(hw);
      e.z = __float_as_int(lw);
      e.w = xlo;
      reinterpret_cast<int4*>(rec + kRecX)[lane] = e;
    }
  }
  // backward block (roi_align_record_layout.h): per window column / row (lane = column / row) the output columns /
  // rows that reach it and the summed weight of their samples -- what the tile kernel's two passes multiply with.
  // A sample whose lower tap is c taps c with hw and c + 1 with lw; samples are sorted by their lower tap, so the
  // samples that tap column c are the contiguous range [first(c - 1), first(c + 1)) with first(c) = #samples with
  // lower tap < c.  The weights of one bin are summed in sample order (fp32, the order the kernel used to sum them in).
  const int nrows_win = wy1 - wy0 + 1;
  const bool bwd_ok = fast && ww <= kMaxWin && nrows_win <= kMaxWin;
  if (bwd_ok && bwd_tables) {  // forward-only callers (a workspace without room for a backward) skip these 2 us
    char* blk = reinterpret_cast<char*>(rec + kRecB);
    auto merge_axis = [&](int lo, float hwv, float lwv, int ns, int g, int w0, int nwin, int off_w, int off_p, int off_f) {
      int first = 0;
      for (int i = 0; i < ns; i++) first += (__builtin_amdgcn_readlane(lo, i) < w0 + lane) ? 1 : 0;
      const int sa = first;
      int sb = __shfl_down(first, 1);
      if (lane == 63) sb = ns;
      int sp = __shfl_up(first, 1);
      if (lane == 0) sp = 0;
      const bool inside = lane < nwin;
      const int nb = (inside && sb > sp) ? (sb - 1) / g - sp / g + 1 : 0;
      int incl = nb;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
      }
      const int at = incl - nb;  // lanes past the window hold the total
      reinterpret_cast<unsigned char*>(blk + off_f)[lane] = (unsigned char)at;
      int most = nb;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) most = max(most, __shfl_xor(most, d));
      float* wdst = reinterpret_cast<float*>(blk + off_w);
      unsigned char* pdst = reinterpret_cast<unsigned char*>(blk + off_p);
      for (int b = 0; b < most; b++) {  // wave-uniform trip count: every lane takes part in the shuffles
        const int p = sp / g + b;
        float wgt = 0.f;
        for (int i = 0; i < g; i++) {
          const int sidx = p * g + i;
          const int src = min(max(sidx, 0), 63);
          const float l = __shfl(lwv, src), h = __shfl(hwv, src);
          if (b < nb && sidx >= sp && sidx < sb) wgt += sidx < sa ? l : h;
        }
        if (b < nb) {
          wdst[at + b] = wgt;
          pdst[at + b] = (unsigned char)p;
        }
      }
    };
    merge_axis(xlo, xhw, xlw, nsx, gw, wx0, ww, kBwdWx, kBwdPx, kBwdCf);
    merge_axis(ylo, yhw, ylw, nsy, gh, wy0, nrows_win, kBwdWy, kBwdPy, kBwdRf);
  }
  if (lane == 0) {
    // A RoI the backward tables cannot describe is found by every tile its samples can tap -- rows / columns
    // [(int)max(first, 0), (int)max(last, 0) + 1] clamped to the map, a superset of roi_align_kernel.cu:150-190's taps --
    // marked kBoundsNoTables, and whoever finds it adds its share of the tile with the reference's per-sample arithmetic
    // (bwd_slow_in_tile).  A RoI of no image keeps the interval that overlaps no tile.
    int bx0 = wx0, bx1 = wx1, by0 = wy0, by1 = wy1;
    if (!fast) {
      by0 = min((int)fminf(fmaxf(yf, 0.f), (float)height), max(height - 1, 0));
      by1 = min((int)fminf(fmaxf(yl, 0.f), (float)height) + 1, max(height - 1, 0));
      bx0 = min((int)fminf(fmaxf(xf, 0.f), (float)width), max(width - 1, 0));
      bx1 = min((int)fminf(fmaxf(xl, 0.f), (float)width) + 1, max(width - 1, 0));
      by0 = min(by0, by1);
      bx0 = min(bx0, bx1);
    }
    const bool listed = !(flags & kFlagZero);
    int4 bnd;
    bnd.x = listed ? bx0 : 0x3fffffff;  // an interval that overlaps no tile
    bnd.y = listed ? (bx1 | (bwd_ok ? 0 : kBoundsNoTables)) : 0;
    bnd.z = lv.row_base[lvl] + batch_ind * height + by0;  // "global rows": levels and images stacked
    bnd.w = lv.row_base[lvl] + batch_ind * height + by1;
    reinterpret_cast<int4*>(ws + kCounterDwords + (long long)num_rois * kRecDwords)[rank] = bnd;
  }
  // stages: consecutive bin rows whose window fits half the LDS image (so that the next stage can be prefetched while
  // this one is computed), the whole image if a single bin row needs it; at most max_rows_tile output rows
  int nstages = 0;
  if (fast && !(ablate & 128)) {
    const int half = stage_px;
    int ph0 = 0;
    while (ph0 < aligned_height) {
      const int row0 = __builtin_amdgcn_readlane(ylo, ph0 * gh);
      int e = ph0;
      int row1 = row0;
      while (e < aligned_height && e < ph0 + max_rows_tile) {
        const int hi = __builtin_amdgcn_readlane(ylo, e * gh + gh - 1) + 1;
        const int px = (hi - row0 + 1) * ((ww + 3) & ~3);
        if (px > (e == ph0 ? cap_px : half)) break;
        row1 = hi;
        e++;
      }
      if (e == ph0) {  // a single bin row does not fit the LDS image
        fast = false;
        break;
      }
      if (lane == 0) {
        int4 st;
        st.x = ph0 | (e << 16);
        st.y = row0;
        st.z = row1 - row0 + 1;
        st.w = 0;
        reinterpret_cast<int4*>(rec + kRecStages)[nstages] = st;
        if (nstages == 0) reinterpret_cast<int4*>(rec)[3] = st;  // stage 0 rides in the header (one scalar load)
      }
      nstages++;
      ph0 = e;
    }
  }
  if (fast) flags |= kFlagFast;
  if (bwd_ok) flags |= kFlagBwd;
  if (tabs) flags |= kFlagTabs;
  if (lane == 0) {
    int4 h0, h1, h2;
    h0.x = flags;
    h0.y = batch_ind;
    h0.z = wx0;
    h0.w = ww;
    h1.x = (int)((1u << 20) / (((unsigned)ww + 3u) >> 2) + 1u);  // divides a lane's group index by the groups per row
    h1.y = nstages;
    h1.z = gh;
    h1.w = gw;
    h2.x = r;
    h2.y = wy0;
    h2.z = wy1;
    h2.w = lvl;
    reinterpret_cast<int4*>(rec)[0] = h0;
    reinterpret_cast<int4*>(rec)[1] = h1;
    reinterpret_cast<int4*>(rec)[2] = h2;
    // the pipelined forward takes the image's address and the map's size from the record (one LDS read per item instead
    // of an indexed walk of the level table); null maps (backward-only calls) leave an address nobody uses
    const uintptr_t img = reinterpret_cast<uintptr_t>(lv.feat[lvl] + (long long)batch_ind * channels * height * width);
    int4 h4;
    h4.x = (int)(unsigned)(img & 0xffffffffu);
    h4.y = (int)(unsigned)(img >> 32);
    h4.z = height;
    h4.w = width;
    reinterpret_cast<int4*>(rec)[4] = h4;
  }
}

// ---- LDS layout of the main kernel ------------------------------------------------------------------------------
struct TabEntry {
  int off;
  float hw, lw;
  int lo;
};


__device__ __forceinline__ unsigned lds_addr_opaque(const void* p) {
  unsigned a = (unsigned)(uintptr_t)(lds_cfloat_t)p;
  asm volatile("" : "+v"(a));
  return a;
}
typedef float v2f_t __attribute__((ext_vector_type(2)));
// the tap pair (F[x], F[x + 1]) of one row: one ds_read2_b32 into a register pair
__device__ __forceinline__ v2f_t lds_pair2(unsigned a) {
  const lds_cfloat_t q = (lds_cfloat_t)(uintptr_t)a;
  return v2f_t{q[0], q[1]};
}
__device__ __forceinline__ const float* lds_at(const float* base, int byte_off) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

// -------------------------------------------------------------------------------------------------------------------
// Forward over the records: roi_align_fwd_records, one workgroup per (rank, channel tile), dispatched in sweep order.  The
// stage pieces (window DMA, axis tables, border patch, bins, tile store) are functions of their own.  (Round 5 also built a
// RESIDENT form -- 768 workgroups walking the items by ticket, record fronts and next windows in flight across items --
// whose unit chain was 11 % shorter and whose 2-item commitment cost a tail of 8-10 us: profiles/r05_persist_timeline.txt.
// Its stage pipelining lives on below; the kernel itself lost on every shape once the bins used packed FMAs, and is gone.)
// -------------------------------------------------------------------------------------------------------------------
// What a workgroup needs of a record's header -- wave-uniform, fetched with scalar loads.
struct FwdRec {
  int flags, wx0, ww, nstages, gh, gw, r, lvl, height, width;
  unsigned gmagic;        // 2^20 / (groups of 4 pixels per window row) + 1
  uintptr_t img;          // address of channel 0 of the RoI's image in its level's map
  int st_pp, st_row0, st_nrows;  // stage 0: ph0 | ph1 << 16, first window row, rows
};
__device__ __forceinline__ FwdRec fwd_load_rec(const int* __restrict__ records, int pos) {
  const const_int_ptr rec = (const_int_ptr)(uintptr_t)(records + (long long)pos * kRecDwords);
  FwdRec h;
  h.flags = rec[0];
  h.wx0 = rec[2];
  h.ww = rec[3];
  h.gmagic = (unsigned)rec[4];
  h.nstages = rec[5];
  h.gh = rec[6];
  h.gw = rec[7];
  h.r = rec[8];
  h.lvl = rec[11];
  h.st_pp = rec[12];
  h.st_row0 = rec[13];
  h.st_nrows = rec[14];
  h.img = ((uintptr_t)(unsigned)rec[17] << 32) | (unsigned)rec[16];
  h.height = rec[18];
  h.width = rec[19];
  return h;
}

// The window rows [row0, row0 + nrows) of this wave's kChPerWave channels -> LDS planes, lanes flattened over the
// window's (row, group of 4 pixels): buffer_load_dwordx4 ... lds, a quarter of the DMA instructions of a pixel-per-lane
// copy (neither side needs more than dword alignment) -- what the compute unit's address path charges for is
// instructions (profiles/r04_records_timeline.txt).  NO branch around the pieces: a group that runs past the end of a map
// row drags in the first pixels of the next row (or zeros behind the wave's channels: the descriptor ends with them),
// which land in padding columns nobody reads -- except the column one past the map of a window that touches the right
// edge, which fwd_patch_edge rewrites.  Rows past the map read the last row again through the clamp.
// Config 2, alternating runs: 37.9 -> 37.1 us per call (two-image box head 69.3 -> 68.4, pyramid 67.0 -> 65.8): the
// pieces still touch the same cache lines, so the gain is the instruction count only.
template <int kChPerWave, int kPlane>
__device__ __forceinline__ void fwd_issue_window(const FwdRec& h, int first_channel, unsigned plane0, int lane, int row0,
                                                 int nrows) {
  const unsigned plane_bytes = (unsigned)h.height * (unsigned)h.width * 4u;
  const srd_t srd = make_srd(reinterpret_cast<const char*>(h.img) + (size_t)first_channel * plane_bytes,
                             (unsigned)kChPerWave * plane_bytes);
  const unsigned pitch_px = ((unsigned)h.ww + 3u) & ~3u;
  const unsigned gpr = pitch_px >> 2, groups = (unsigned)nrows * gpr;
  for (int kk = 0; kk * 64 < (int)groups; kk++) {
    const unsigned g = (unsigned)(kk * 64 + lane);
    const unsigned q = __umul24(g, h.gmagic) >> 20;  // g / gpr
    const unsigned gc = g - __umul24(q, gpr);
    const unsigned voff = (__umul24(min((unsigned)row0 + q, (unsigned)h.height - 1u), (unsigned)h.width) +
                           (unsigned)h.wx0 + gc * 4u) * 4u;
    if (g < groups) {
#pragma unroll
      for (int c = 0; c < kChPerWave; c++)
        dma_dwordx4(srd, plane0 + (unsigned)(c * kPlane + kk * 256) * 4u, voff, (unsigned)c * plane_bytes);
    }
  }
}

// axis tables: record -> LDS (wave 0: y, wave 1: x)
__device__ __forceinline__ void fwd_issue_tables(const int* __restrict__ records, int pos, int wave, int lane,
                                                 TabEntry* tab, int nsy, int nsx) {
  if (wave >= 2) return;
  const int n = (wave == 0 ? nsy : nsx) * 4;
  const srd_t tsrd = make_srd(records + (long long)pos * kRecDwords + (wave == 0 ? kRecY : kRecX), 4 * kMaxS * 4);
  const unsigned dstl = lds_addr_uniform(tab + wave * kMaxS);
  for (int kk = 0; kk * 64 < n; kk++) dma_dword(tsrd, dstl + (unsigned)kk * 256u, (unsigned)(kk * 64 + lane) * 4u, 0u);
}

// A window that ends one column past the map (a border sample's upper tap, weight 0: axis_taps): that column has to hold
// the border pixel again -- the reference reads it twice -- not what followed it in memory.  Wave-uniform condition; the
// caller puts a barrier behind it.
template <int kCT, int kThreads, int kPlane>
__device__ __forceinline__ bool fwd_patch_edge(const FwdRec& h, float* img, int tid, int nrows) {
  if (h.wx0 + h.ww <= h.width) return false;
  const int edge = h.width - 1 - h.wx0;  // window column of the map's last pixel
  const int pitch_px = (h.ww + 3) & ~3;
  for (int i = tid; i < kCT * nrows; i += kThreads) {
    const int c = i / nrows, rr = i - c * nrows;
    float* rowp = img + c * kPlane + rr * pitch_px;
    rowp[edge + 1] = rowp[edge];
  }
  return true;
}

// Bins [pa, pb) x aligned_width of this lane's channel -> the LDS tile.  Half-wave = output column; taps of 4 bin rows in
// flight before the first use; 0.25 * sum_iy (hy * R(y) + ly * R(y + 1)), R(row) = sum_ix (hx * F[x] + lx * F[x + 1]).
template <int kSR, int kNSlots>
__device__ __forceinline__ void fwd_bins(const TabEntry* ty, const TabEntry* tx, const float* img_c, float* tile_c,
                                         int slot, int pa, int pb, int ph0, int base_off, int pitch, int aligned_width,
                                         int gh, int gw) {
  if (kSR > 0) {
    constexpr int kS = kSR > 0 ? kSR : 1;
    for (int pw = slot; pw < aligned_width; pw += kNSlots) {
      v2f_t hxl[kS];
      unsigned xa[kS];
#pragma unroll
      for (int i = 0; i < kS; i++) {
        const TabEntry ex = tx[pw * kS + i];
        hxl[i] = v2f_t{ex.hw, ex.lw};
        xa[i] = lds_addr_opaque(lds_at(img_c, ex.off - base_off));
      }
      auto rows = [&](int ph, auto kn) {
        constexpr int kN = decltype(kn)::value;
        // A tap pair (F[x], F[x + 1]) arrives in an even-aligned register pair and so do its weights (hx, lx): the row sums
        // run as v_pk_fma_f32 on (even, odd) partial sums -- S = sum_ix (hx, lx) * (F[x], F[x + 1]), acc += (wy, wy) * S --
        // and the two halves meet once per bin: 13 instead of 20 FMA-class instructions per bin (the bins are bound by
        // VALU issue: ~170 instructions per batch of four bin rows before this).  Same taps and weights; the summation
        // order differs from the scalar chain in the last bits (contract 1e-4, asserted 1e-5).
        v2f_t v[kN][kS][2][kS];
        float wy[kN][kS][2];
#pragma unroll
        for (int b = 0; b < kN; b++) {
#pragma unroll
          for (int iy = 0; iy < kS; iy++) {
            const TabEntry ey = ty[(ph + b) * kS + iy];
            wy[b][iy][0] = ey.hw;
            wy[b][iy][1] = ey.lw;
#pragma unroll
            for (int ix = 0; ix < kS; ix++) {
              const unsigned a = xa[ix] + (unsigned)ey.off;
              v[b][iy][0][ix] = lds_pair2(a);
              v[b][iy][1][ix] = lds_pair2(a + (unsigned)pitch);
            }
          }
        }
#pragma unroll
        for (int b = 0; b < kN; b++) {
          v2f_t acc = {0.f, 0.f};
#pragma unroll
          for (int iy = 0; iy < kS; iy++) {
#pragma unroll
            for (int kx = 0; kx < 2; kx++) {
              v2f_t rsum = hxl[0] * v[b][iy][kx][0];
#pragma unroll
              for (int ix = 1; ix < kS; ix++) rsum = __builtin_elementwise_fma(hxl[ix], v[b][iy][kx][ix], rsum);
              acc = __builtin_elementwise_fma(v2f_t{wy[b][iy][kx], wy[b][iy][kx]}, rsum, acc);
            }
          }
          tile_c[(ph + b - ph0) * aligned_width + pw] = acc.x + acc.y;
        }
      };
      int ph = pa;
      for (; ph + 4 <= pb; ph += 4) rows(ph, std::integral_constant<int, 4>());
      switch (pb - ph) {
        case 3: rows(ph, std::integral_constant<int, 3>()); break;
        case 2: rows(ph, std::integral_constant<int, 2>()); break;
        case 1: rows(ph, std::integral_constant<int, 1>()); break;
        default: break;
      }
    }
  } else {
    for (int pw = slot; pw < aligned_width; pw += kNSlots) {
      for (int ph = pa; ph < pb; ph++) {
        float acc = 0.f;
        for (int iy = 0; iy < gh; iy++) {
          const TabEntry ey = ty[ph * gh + iy];
          float r0s = 0.f, r1s = 0.f;
          for (int ix = 0; ix < gw; ix++) {
            const TabEntry ex = tx[pw * gw + ix];
            const float* a = lds_at(img_c, ey.off + ex.off - base_off);
            const float* b = lds_at(a, pitch);
            r0s = __builtin_fmaf(ex.hw, a[0], r0s);
            r0s = __builtin_fmaf(ex.lw, a[1], r0s);
            r1s = __builtin_fmaf(ex.hw, b[0], r1s);
            r1s = __builtin_fmaf(ex.lw, b[1], r1s);
          }
          acc = __builtin_fmaf(ey.hw, r0s, acc);
          acc = __builtin_fmaf(ey.lw, r1s, acc);
        }
        tile_c[(ph - ph0) * aligned_width + pw] = acc;
      }
    }
  }
}

// LDS tile (kCT channels x nb bins, channel stride ts) -> out[r][c0 ..][ph0 * aligned_width ..]: contiguous 16-byte pieces
// when the stage covers the RoI's whole [kCT][bins] block.
template <int kCT, int kThreads>
__device__ __forceinline__ void fwd_store(const float* tile, float* __restrict__ dst, int tid, int ph0, int nb, int ts,
                                          int bins, int aligned_width, bool plain = false) {
  if (nb == bins && ts == nb && ((kCT * nb) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    // non-temporal: the pooled features are read next by another kernel, and a streaming store lets the workgroup's LDS go
    // ~1 % earlier (38.55 -> 38.25 us per config-2 call, three alternating runs)
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    if (plain) {
      for (int i = tid; i < kCT * nb / 4; i += kThreads) reinterpret_cast<v4f_t*>(dst)[i] = reinterpret_cast<const v4f_t*>(tile)[i];
      return;
    }
    for (int i = tid; i < kCT * nb / 4; i += kThreads)
      __builtin_nontemporal_store(reinterpret_cast<const v4f_t*>(tile)[i], reinterpret_cast<v4f_t*>(dst) + i);
  } else {
    float* gdst = dst + ph0 * aligned_width;
    const unsigned nb_magic = (1u << 20) / (unsigned)nb + 1u;
    for (int i = tid; i < kCT * nb; i += kThreads) {
      const int c = (int)(((unsigned)i * nb_magic) >> 20), b = i - c * nb;
      gdst[(long long)c * bins + b] = tile[c * ts + b];
    }
  }
}

// store instructions fwd_store issues in one wave (its first lane runs the most trips of the loop)
template <int CT, int NT>
__device__ __forceinline__ int fwd_store_count(const float* dst, int wave, int nb, int ts, int bins) {
  const bool whole = nb == bins && ts == nb && ((CT * nb) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
  const int n = whole ? CT * nb / 4 : CT * nb;
  return n > wave * 64 ? (n - wave * 64 + NT - 1) / NT : 0;
}

__device__ __forceinline__ void wait_vmcnt_at_most(int n) {
  // s_waitcnt takes an immediate; n is wave-uniform.  Waiting for MORE than asked (a smaller immediate) is always safe.
  // What the partial wait assumes: (1) on the gfx9 family (gfx90a / gfx942 / gfx950) loads -- LDS-DMA included -- and stores
  // share ONE counter, vmcnt, which retires in issue order, so "at most n outstanding" with n = the stores issued after the
  // window pieces means the pieces have landed; (2) fwd_store_count() is an upper bound of the store instructions a wave
  // issues per fwd_store() -- if the compiler merged stores the wait would only be stricter, if it split one it would be too
  // weak.  (2) is pinned by test_roi_align_forward_partial_wait_equals_the_full_wait (MI_ROI_ALIGN_FWD_FULL_WAIT=1 against
  // the default, bit for bit, on multi-stage shapes in the release build); any other target takes vmcnt(0).
#if !(defined(__gfx950__) || defined(__gfx942__) || defined(__gfx940__) || defined(__gfx90a__))
  n = 0;
#endif
  if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// A RoI the LDS image cannot serve (flagged by roi_align_prepare): zeros for a RoI of no image, otherwise the reference's
// operation order straight from memory (bit-exact).
template <int kCT, int kThreads>
__device__ __forceinline__ void fwd_direct_item(const FwdRec& h, const LevelTable& lv, const float* __restrict__ rois,
                                                float* __restrict__ dst, int tid, int c0, int channels,
                                                int aligned_height, int aligned_width, int sampling_ratio) {
  const int bins = aligned_height * aligned_width;
  if (h.flags & kFlagZero) {
    for (int i = tid; i < kCT * bins; i += kThreads) dst[i] = 0.f;
    return;
  }
  const int height = h.height, width = h.width;
  const RoiGeom g = roi_geometry(rois + (long long)h.r * 5, lv.scale[h.lvl], aligned_height, aligned_width, sampling_ratio);
  const float* src = lv.feat[h.lvl] + ((long long)g.batch_ind * channels + c0) * height * width;
  for (int i = tid; i < kCT * bins; i += kThreads) {
    const int c = i / bins, bin = i - c * bins;
    const int ph = bin / aligned_width, pw = bin - ph * aligned_width;
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

// kA > 0: aligned_height == aligned_width == kA at compile time (7: box head, 14: mask / keypoint heads).
template <int kSR, int kCap, int kA = 0>
__global__ void __launch_bounds__(kCT * 8)
roi_align_fwd_records(const float* __restrict__ rois, float* __restrict__ out, const int* __restrict__ ws, int num_rois,
                      int batch, int channels, int aligned_height_arg, int aligned_width_arg, int sampling_ratio, int split,
                      int ablate_arg, int full_wait, const LevelTable lv MI_TL_PARAM) {
  MI_STAMP(0);
  const int aligned_height = kA > 0 ? kA : aligned_height_arg, aligned_width = kA > 0 ? kA : aligned_width_arg;
  const int ablate = MI_ABLATE(ablate_arg);
  constexpr int kChPerWave = kCT / (kThreads / 64);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kPlane = kCap | 1;
  constexpr int kTileWords = kCT * (kTileBins + 1);
  TabEntry* tab = reinterpret_cast<TabEntry*>(smem);
  float* tile = reinterpret_cast<float*>(tab + 2 * kMaxS);
  float* img = tile + kTileWords;
  const int tid = threadIdx.x;
  const int bins = aligned_height * aligned_width;
  const int tiles = channels / kCT;
  // split > 1: an item's stages are dealt to `split` workgroups (stage k to part k % split) -- launches of few, long
  // items (the 14x14 heads: 4 stages per item) otherwise run one and a third rounds over the chip's 768 slots
  const int pos_part = blockIdx.x / tiles;
  const int pos = split > 1 ? pos_part / split : pos_part;
  const int part = split > 1 ? pos_part - pos * split : 0;
  const int c0 = (blockIdx.x - pos_part * tiles) * kCT;
  const int wave = uniform(tid >> 6), lane = tid & 63;
  const int cl = tid % kCT, slot = (tid / kCT) & 7;
  const int* __restrict__ records = ws + kCounterDwords;
  const FwdRec h = fwd_load_rec(records, pos);
  float* __restrict__ dst = out + ((long long)h.r * channels + c0) * bins;
  if (h.flags != 0x7fffffff) MI_STAMP(1);  // the record header has arrived
#if MI_TUNING
  if (timeline != nullptr && threadIdx.x == 0)  // where it ran (XCC_ID, HW_ID) and how many stages
    timeline[(long long)blockIdx.x * 8 + 7] = ((long long)h.nstages << 48) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                              (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
  if (!(h.flags & kFlagFast)) {
    if (part == 0)
      fwd_direct_item<kCT, kThreads>(h, lv, rois, dst, tid, c0, channels, aligned_height, aligned_width, sampling_ratio);
    return;
  }
  if (part >= h.nstages) return;
  {
    // Warm this XCD's L2 with the record a later workgroup of this XCD starts from: XCD x runs the ranks in order (block
    // 8 j + x pools rank j), ~96 at a time, and a record written by roi_align_prepare sits in the Infinity Cache at best.
    // 13 lines (header, y table, x table), fetched as an LDS-DMA piece into the still unused output tile -- a load with a
    // register destination would write that register whenever it lands, long after the compiler has given it away.
    // Config 2, three alternating runs: 38.41 -> 37.97 us per call (distances 32 / 64 / 128 were within 0.2 us of each other).
    constexpr int kAhead = 64;
    const int ahead = pos + kAhead;
    if (wave == kThreads / 64 - 1 && ahead < num_rois && lane < 13 && part == 0)
      dma_dword(make_srd(records + (long long)ahead * kRecDwords, 13 * 128), lds_addr_uniform(tile), (unsigned)lane * 128u, 0u);
  }
  const unsigned plane0 = lds_addr_uniform(img + wave * kChPerWave * kPlane);
  const int pitch = ((h.ww + 3) & ~3) * 4;  // a window row lies in LDS on a pitch of whole 16-byte groups
  const int gh = kSR > 0 ? kSR : h.gh, gw = kSR > 0 ? kSR : h.gw;
  const const_int_ptr rec = (const_int_ptr)(uintptr_t)(records + (long long)pos * kRecDwords);
  // Stages are pipelined inside the item (round 5): the window pieces of stage k + 1 are issued
  // BEFORE the tile of stage k is stored, and the landing wait is vmcnt(stores of stage k) -- window pieces are older than
  // those stores and vmcnt retires in order, so the stores drain under the next stage's bins instead of in front of its copy.
  int pp = h.st_pp, row0 = h.st_row0, nrows = h.st_nrows;
  if (part > 0) {
    const const_int_ptr st = rec + kRecStages + 4 * part;
    pp = st[0];
    row0 = st[1];
    nrows = st[2];
  }
  const int step = split > 1 ? split : 1;
  if (!(ablate & 1)) fwd_issue_window<kChPerWave, kPlane>(h, c0 + wave * kChPerWave, plane0, lane, row0, nrows);
  fwd_issue_tables(records, pos, wave, lane, tab, aligned_height * gh, aligned_width * gw);
  MI_STAMP(2);  // window and table pieces issued
  int stores_out = 0;
  for (int k = part; k < h.nstages; k += step) {
    int n_pp = 0, n_row0 = 0, n_nrows = 0;
    if (k + step < h.nstages) {  // the next stage's descriptor arrives under the bins
      const const_int_ptr st = rec + kRecStages + 4 * (k + step);
      n_pp = st[0];
      n_row0 = st[1];
      n_nrows = st[2];
    }
    const int ph0 = pp & 0xffff, ph1 = pp >> 16;
    wait_vmcnt_at_most(((ablate & 16) || full_wait) ? 0 : stores_out);
    __syncthreads();  // this stage's window has landed; the previous tile is out of LDS
    if (k == part) MI_STAMP(3);  // landed, published
    if (fwd_patch_edge<kCT, kThreads, kPlane>(h, img, tid, nrows)) __syncthreads();
    const int nb = (ph1 - ph0) * aligned_width;
    const int ts = nb | 1;
    if (!(ablate & 2))
      fwd_bins<kSR, kSlots>(tab, tab + kMaxS, img + cl * kPlane, tile + cl * ts, slot, ph0, ph1, ph0, row0 * pitch, pitch,
                            aligned_width, gh, gw);
    MI_STAMP(4);  // this wave's bins are in the tile
    __syncthreads();  // the tile is complete, the image is free
    MI_STAMP(5);
    if (k + step < h.nstages && !(ablate & 1))
      fwd_issue_window<kChPerWave, kPlane>(h, c0 + wave * kChPerWave, plane0, lane, n_row0, n_nrows);
    stores_out = 0;
    if (!(ablate & 4)) {
      fwd_store<kCT, kThreads>(tile, dst, tid, ph0, nb, ts, bins, aligned_width);
      stores_out = fwd_store_count<kCT, kThreads>(dst, wave, nb, ts, bins);
    }
    MI_STAMP(6);  // stores issued
    pp = n_pp;
    row0 = n_row0;
    nrows = n_nrows;
  }
}

// -------------------------------------------------------------------------------------------------------------------
// roi_align_fwd_slab: the forward WITHOUT records, one launch (round 6, late).  What the records launch buys the kernel
// above is the sweep order -- a 32-channel slab of a 200x336 map is 8.6 MB, twice an XCD's L2, so its RoIs have to walk
// the map coherently -- and what it costs is a launch of ~5 us plus a kernel boundary in front of a ~30 us gather.  Here
// the slab is cut to fit instead: one WAVE per (RoI, 8 channels), grid (8 * R, phases); workgroup x of a phase works on
// channel tile 8 * phase + (x & 7), i.e. (workgroups go round-robin over the XCDs) at any time one XCD reads ONE
// 8-channel slab (2.15 MB of its 4 MB L2) for all RoIs in arrival order: every line comes from the fabric once whatever
// the order of the RoIs, and nobody has to rank them.  The wave computes the RoI's geometry and axis tables itself (the
// arithmetic of roi_align_prepare, lane = sample), cuts the stages on the fly, and runs the same window DMA and bins as
// the record-driven kernel on its own LDS image; its bins leave straight from the lanes (8 channels x 8 columns per wave: a
// 28-byte run per channel and store instruction -- no LDS output tile) -- no barrier anywhere, 18 independent waves per CU.
// Bit-equal to roi_align_fwd_records (same tables, same bins).  Callers whose workspace announces a backward keep the
// records path: the backward reads the records (at their sweep rank: profiles/r06_slab_forward.txt section 8).
// -------------------------------------------------------------------------------------------------------------------
constexpr int kSlabCT = 8;
#if MI_TUNING
#define MI_SLAB_STAMP(k)                                                                                               \
  do {                                                                                                                \
    if (timeline != nullptr && threadIdx.x == 0)                                                                      \
      timeline[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = (long long)clock64();                    \
  } while (0)
#else
#define MI_SLAB_STAMP(k)                                                                                               \
  do {                                                                                                                \
  } while (0)
#endif
// LDS of a wave: axis tables | image (no output tile: a lane's bins go straight to memory, see the kernel).  The LDS is handed
// out in granules of 1280 bytes (tools/micro/launch_rate.hip: residency of one-wave workgroups steps at 6400 / 7680 / 8960 /
// 10240 / 11520 bytes), and 8 granules is what lets a CU hold 18 waves instead of 14 -- the tables are sized for the instance
// (14 samples per axis at 7x7, sampling ratio 2) and the image takes what is left of 10240 bytes.
template <int kSR, int kA>
struct SlabLds {
  static constexpr int kNS = (kSR > 0 && kA > 0) ? kSR * kA : kMaxS;                          // table entries per axis
  // the largest image capacity = 4 (mod 32) pixels that keeps the wave inside 8 granules
  static constexpr int kCapMax = (((10240 - 2 * kNS * 16) / (kSlabCT * 4) - 4) & ~31) + 4;
};
// Plane stride = 4 (mod 32) dwords with lanes of a 32-lane LDS group = 8 channels x 4 output columns: the eight planes start
// on banks 0, 4, .., 28, and two lanes of a tap read meet in a bank only when two of the group's four columns are congruent
// mod 4 -- 1.73 LDS passes per tap read on config 2's RoIs against 2.72 for an odd stride with 4 channels x 8 columns
// (simulated over the RoI set; all-distinct banks measured 3 us faster than the odd stride, this takes about half of it).
template <int kCap>
constexpr int slab_plane() { return kCap % 32 == 4 ? kCap : (kCap | 1); }
template <int kSR, int kCap, int kA>
constexpr size_t slab_lds_bytes() {
  return 2 * SlabLds<kSR, kA>::kNS * sizeof(TabEntry) + (size_t)(kSlabCT * slab_plane<kCap>()) * 4;
}
// kLevels: the RoIs carry a level index (pyramid calls); otherwise level 0 and its kernel arguments, no indexed fetch
template <int kSR, int kCap, int kA, bool kLevels>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kA > 0 ? 5 : 4)))
roi_align_fwd_slab(const float* __restrict__ rois, float* __restrict__ out, const int* __restrict__ levels, int num_rois,
                   int batch, int channels, int aligned_height_arg, int aligned_width_arg, int sampling_ratio, int full_wait,
                   const LevelTable lv MI_TL_PARAM) {
  MI_SLAB_STAMP(0);
  // tuning builds only (MI_ROI_ALIGN_FWD_FULL_WAIT bits): 4 = no bins (and no stores), 16 = conflict-free taps
  const int ablate = MI_ABLATE(full_wait);
  const int aligned_height = kA > 0 ? kA : aligned_height_arg, aligned_width = kA > 0 ? kA : aligned_width_arg;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kPlane = slab_plane<kCap>();
  constexpr int kNS = SlabLds<kSR, kA>::kNS;
  TabEntry* tab = reinterpret_cast<TabEntry*>(smem);
  float* img = reinterpret_cast<float*>(tab + 2 * kNS);
  const int lane = threadIdx.x;
  const int bins = aligned_height * aligned_width;
  const int c0 = (blockIdx.y * 8 + (blockIdx.x & 7)) * kSlabCT;
  // lanes of a 32-lane LDS group = 8 channels x 4 output columns (see slab_plane)
  const int cl = lane & 7, slot = lane >> 3;
  // The arguments in front of the level table are preloaded into SGPRs (-amdgpu-kernarg-preload-count, build.py): the RoI's
  // five floats are fetched with the first instructions, beside -- not behind -- the rest of the kernel arguments.
  const int r = blockIdx.x >> 3;
  const const_int_ptr rp = (const_int_ptr)(uintptr_t)(rois + (long long)r * 5);
  const float roi_b = __int_as_float(rp[0]), roi_x1 = __int_as_float(rp[1]), roi_y1 = __int_as_float(rp[2]);
  const float roi_x2 = __int_as_float(rp[3]), roi_y2 = __int_as_float(rp[4]);
  const int lvl = (kLevels && levels != nullptr) ? min(max(((const_int_ptr)(uintptr_t)levels)[r], 0), lv.count - 1) : 0;
  if (c0 >= channels) return;
  // ---- geometry: roi_align_prepare's arithmetic (roi_align_kernel.cu:74-110, :16-52), lane = y sample AND x sample ----
  const int height = lv.height[lvl], width = lv.width[lvl];
  const float spatial_scale = lv.scale[lvl];
  const int batch_ind = (int)roi_b;
  const float start_w = roi_x1 * spatial_scale, start_h = roi_y1 * spatial_scale;
  const float roi_width = fmaxf(roi_x2 * spatial_scale - start_w, 1.f);
  const float roi_height = fmaxf(roi_y2 * spatial_scale - start_h, 1.f);
  const float bin_h = roi_height / (float)aligned_height, bin_w = roi_width / (float)aligned_width;
  const int gh = kSR > 0 ? kSR : (sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)aligned_height));
  const int gw = kSR > 0 ? kSR : (sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)aligned_width));
  const float count = (float)(gh * gw);
  const int nsy = aligned_height * gh, nsx = aligned_width * gw;
  FwdRec h;
  h.flags = (batch_ind < 0 || batch_ind >= batch) ? kFlagZero : 0;
  h.r = r;
  h.lvl = lvl;
  h.height = height;
  h.width = width;
  h.gh = gh;
  h.gw = gw;
  float* __restrict__ dst = out + ((long long)r * channels + c0) * bins;
  // The first and last sample of an axis ARE the window's ends (samples do not decrease along the lanes), so the window,
  // the band test and the clamp of roi_align_prepare fall out of the per-lane taps: four v_readlane instead of four scalar
  // axis_taps in front of everything else.
  const int lane_y = min(lane, max(min(nsy, 64), 1) - 1), ph_of = lane_y / gh;
  const int lane_x = min(lane, max(min(nsx, 64), 1) - 1), pw_of = lane_x / gw;
  const float yc = coord(start_h, bin_h, ph_of, lane_y - ph_of * gh, gh);
  const float xc = coord(start_w, bin_w, pw_of, lane_x - pw_of * gw, gw);
  int ylo, xlo;
  float yhw, ylw, xhw, xlw;
  axis_taps(yc, height, ylo, yhw, ylw);
  axis_taps(xc, width, xlo, xhw, xlw);
  const float yf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yc), 0));
  const float yl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yc), 63));
  const float xf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xc), 0));
  const float xl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xc), 63));
  bool fast = h.flags == 0 && nsy <= kMaxS && nsx <= kMaxS && aligned_height <= kMaxStages && height >= 2 && width >= 2 &&
              !(yf < -1.0f || yl > (float)height || xf < -1.0f || xl > (float)width) && yl >= yf && xl >= xf;
  const int wx0 = __builtin_amdgcn_readlane(xlo, 0);
  const int wx1 = __builtin_amdgcn_readlane(xlo, 63) + 1;
  const int ww = wx1 - wx0 + 1;
  const int pitch_px = (ww + 3) & ~3;
  // The stages are cut from the lower taps of the LAST sample of every bin row (lanes `is_last`): "the bin rows from ph0 on
  // whose window still fits" is a prefix of them -- one ballot.
  const bool is_last = lane < nsy && lane - ph_of * gh == gh - 1;
  const int max_rows_tile = aligned_height;  // no output tile: a stage takes as many bin rows as its window fits the image
  auto cut = [&](int ph0, int& e, int& row0, int& nrows) {
    row0 = __builtin_amdgcn_readlane(ylo, ph0 * gh);
    const bool ok = is_last && ph_of >= ph0 && ph_of < ph0 + max_rows_tile && (ylo + 1 - row0 + 1) * pitch_px <= kCap;
    e = ph0 + __popcll(__ballot(ok));
    const int row1 = e > ph0 ? __builtin_amdgcn_readlane(ylo, e * gh - 1) + 1 : row0;
    nrows = row1 - row0 + 1;
  };
  if (fast) {
    // a bin row whose window does not fit the LDS image sends the whole RoI down the direct path (as the records do)
    const int first = __shfl_up(ylo, gh - 1);
    if (__ballot(is_last && (ylo + 1 - first + 1) * pitch_px > kCap) != 0ull) fast = false;
  }
  if (!fast) {
    fwd_direct_item<kSlabCT, 64>(h, lv, rois, dst, lane, c0, channels, aligned_height, aligned_width, sampling_ratio);
    return;
  }
  h.wx0 = wx0;
  h.ww = ww;
  h.gmagic = (1u << 20) / (((unsigned)ww + 3u) >> 2) + 1u;
  h.img = reinterpret_cast<uintptr_t>(lv.feat[lvl] + (long long)batch_ind * channels * height * width);
  const unsigned plane0 = lds_addr_uniform(img);
  const int pitch = pitch_px * 4;
  int ph0 = 0, ph1, row0, nrows;
  cut(0, ph1, row0, nrows);
  MI_SLAB_STAMP(1);  // geometry and the first stage are known
  fwd_issue_window<kSlabCT, kPlane>(h, c0, plane0, lane, row0, nrows);
  MI_SLAB_STAMP(2);  // window pieces issued
  // ---- axis tables, under the window's flight (roi_align_prepare's entries: lane = sample) ----
  if (lane < nsy) {
    TabEntry e;
    e.off = ylo * pitch_px * 4;
    e.hw = yhw / count;
    e.lw = ylw / count;
    e.lo = ylo;
    tab[lane] = e;
  }
  if (lane < nsx) {
    TabEntry e;
    e.off = (ablate & 16) ? (pw_of * 4 + (lane - pw_of * gw)) * 4 : (xlo - wx0) * 4;  // 16: conflict-free taps (wrong pixels)
    e.hw = xhw;
    e.lw = xlw;
    e.lo = xlo;
    tab[kNS + lane] = e;
  }
  int nst = 0;
  while (true) {
    int n_ph1 = 0, n_row0 = 0, n_nrows = 0;
    if (ph1 < aligned_height) cut(ph1, n_ph1, n_row0, n_nrows);
    wait_vmcnt_at_most(0);  // this stage's window has landed (a single wave: no barrier)
    if (nst == 0) MI_SLAB_STAMP(3);
    nst++;
    if (h.wx0 + h.ww > h.width) fwd_patch_edge<kSlabCT, 64, kPlane>(h, img, lane, nrows);
    // A lane's bins go STRAIGHT to memory (dword stores in runs of aligned_width floats per channel) instead of through an LDS
    // tile and 16-byte stores: seven store instructions per lane instead of two, but no tile write / read-back round trip in
    // front of them and 1.5 KB less LDS per wave -- config 2 28.9 -> 27.7 us, a step's pyramid (forward only) 53.5 -> 48.9 and
    // 25.1 -> 22.1, 128 x 14x14 22.8 -> 22.0, 1024 RoIs on two images 54.6 -> 55.5 (A/B, bit-equal).
    if (!(ablate & 4))
      fwd_bins<kSR, 8>(tab, tab + kNS, img + cl * kPlane, dst + (long long)cl * bins, slot, ph0, ph1, 0, row0 * pitch, pitch,
                       aligned_width, gh, gw);
    const bool more = ph1 < aligned_height;
#if MI_TUNING
    if (!more) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MI_SLAB_STAMP(4);
      MI_SLAB_STAMP(5);
    }
#endif
    if (more) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the bins' reads of the image are done
      fwd_issue_window<kSlabCT, kPlane>(h, c0, plane0, lane, n_row0, n_nrows);
    }
    // this stage's stores were issued BEFORE the next window's pieces: only vmcnt(0) says that those have landed
    if (!more) {
      MI_SLAB_STAMP(6);
#if MI_TUNING
      if (timeline != nullptr && threadIdx.x == 0)
        timeline[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] =
            ((long long)nst << 48) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
            (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
      break;
    }
    ph0 = ph1;
    ph1 = n_ph1;
    row0 = n_row0;
    nrows = n_nrows;
  }
}

// -------------------------------------------------------------------------------------------------------------------
// Backward (roi_align_kernel.cu:150-270) as a GATHER over tiles of the feature-map gradient -- no zero fill, and no
// atomics except where a planned launch cuts a tile's RoI list into slices (their sums are added atomically; lists of
// up to 32 RoIs, and every list of the unplanned launch, are summed in a fixed order).
//
// Measured first on MI355X (512 RoIs x 256 ch x 7x7 on 200x336): scatter formulations are bound by the atomic units,
// not by bandwidth -- 37 M global_atomic_add_f32 (one per window pixel and channel, already coalesced) cost ~190 us,
// and accumulating a RoI's window in LDS with ds_add_f32 is slower still (~500 us: the LDS executes float atomics
// lane by lane).  The reference's 16 atomics per output element are worse on both counts.
//
// The bilinear scatter is separable:  dF[row][col] = sum_ph wy(ph,row) * ( sum_pw wx(pw,col) * g[ph][pw] ).
// One 256-lane workgroup owns an 8-row x 32-column tile of dF for KC channels and walks the RoIs whose window touches
// the tile (found by scanning the int4 window of every rank; kept in sweep order, so the summation order is fixed):
//   pass 1  lanes = (channel, column): T[c][ph][col] = sum over the output columns pw that reach `col` of
//           Wx(col, pw) * g[c][ph][pw]   (7 register accumulators)
//   pass 2  lanes = (row, column), a half-wave per row: acc[c] += sum over the bin rows ph that reach `row` of
//           Wy(row, ph) * T[c][ph][col], for the KC channels held in registers
// Wx / Wy -- per window column / row the list of bins and the summed weight of their samples -- come merged from
// roi_align_prepare (the record's backward block; the passes used to rebuild them per lane from the sample tables:
// PMC showed them bound by that integer work).  g and the block of the NEXT RoI arrive by LDS-DMA while the current
// one is processed.
// At the end each lane stores its KC sums as 128-byte rows (or adds them to what the caller supplied).
// RoIs the tables cannot describe (a sample outside the [-1, size] band, window > 63 rows or columns, > 32 samples
// per axis) are listed like the others and added by the same workgroup behind its store, with the reference's
// arithmetic and atomics restricted to the tile (bwd_slow_in_tile; a launch of its own until round 5).
// Weights: g * (hy / count) * hx instead of the reference's g * (hy * hx) / count -- fp32 rounding only (the
// accumulation order of the reference's atomics is unspecified; contract 1e-4).
// -------------------------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kTW = 32;  // columns of the dF tile owned by a workgroup (rows: template parameter, 32 lanes each)

template <int KC>
struct BwdLds {
  static constexpr int kTabDw = kBwdTabDw;  // the record's backward block: merged weights of both passes
  static constexpr int kTStride = kMaxStages * kTW + 1;            // words per channel of T (odd): up to 32 bin rows
  static constexpr int kGWords = KC * kTileBins * 4;               // g block: up to 224 bins per channel
};

// roi_align_bwd_plan: one 256-lane workgroup per tile, in front of roi_align_bwd_tiles when the caller's workspace has
// room for the plan (roi_align_bwd_workspace_bytes).  It finds the RoIs whose window touches the tile ONCE (the unplanned
// tile kernel repeats the scan in each of its channel groups), leaves their ranks and their number in the workspace, and
// files the tile's work as ENTRIES {tile, first, length, slices} in one of six cost classes:
//   class 0   the slices of a list longer than `slice_len` (even slices of <= slice_len RoIs) -- while the budget of extra
//             entries lasts (sum of slices - 1 <= `extra`: the tile kernel's grid is tiles + extra rows); then whole
//   class 1-4 whole lists of >= 20, 10-19, 5-9, 1-4 RoIs
//   class 5   (OVERWRITE contract only) tiles without RoIs: their item stores zeros
// A class is an array filled through one atomic counter (ws[kBwdBucket + 32 c]: one 128-byte line each -- atomics on one line
// serialise at ~11 ns apiece); the order inside a class is the order of arrival and only decides WHO sums a tile, never
// in which order (a list is summed in rank order by one workgroup, or slice-wise with atomics as before).  The resident
// tile kernel walks class 0, 1, ... 5: longest processing time first, so that what is left for the end of the launch are
// the one-visit items and the zero stores.  (Rounds 3-4 built the same table with a single sorting workgroup in a launch
// of its own -- roi_align_bwd_items, 5.6 us + a launch boundary per call -- and dispatched one workgroup per table row.)
// Why slices: the RoIs of a training step cluster on the ground-truth boxes (a few 16 x 32 tiles of P4 see 100-200 of the
// 1024 RoIs, most tiles none), and one workgroup per tile then works for 300 us while the rest of the chip is idle.
// A sliced tile is zero-filled here (OVERWRITE contract: its slices ADD): under that contract every tile has kZeroParts
// more workgroups in this launch, which repeat the count and, if the list is long, zero their share of the channels (one
// workgroup zero-filling a whole 512 KB tile made this launch 35 us long on a step's clustered RoIs).
constexpr int kPlanThreads = 256, kPlanWaves = kPlanThreads / 64;  // (1024 lanes: 9 roles x 402 tiles of 16-wave workgroups took 20 us to dispatch)
constexpr int kMaxPlanTiles = 8192;
constexpr int kZeroParts = 8;
__host__ __device__ inline int bwd_plan_extra(int num_rois) {  // entries beyond one per tile
  return num_rois / 4 < 64 ? 64 : (num_rois / 4 > 2048 ? 2048 : num_rois / 4);
}
__host__ __device__ inline int bwd_class_cap0(int tiles, int num_rois) { return (tiles + bwd_plan_extra(num_rois) + 3) & ~3; }
__host__ __device__ inline int bwd_class_base(int c, int tiles, int num_rois) {  // first entry of class c
  return c == 0 ? 0 : bwd_class_cap0(tiles, num_rois) + (c - 1) * tiles;
}
__device__ __forceinline__ int bwd_class_of(int len) { return len >= 20 ? 1 : len >= 10 ? 2 : len >= 5 ? 3 : len >= 1 ? 4 : 5; }
// One RoI without backward tables, its contribution to `nch` channels of ONE tile of the gradient map:
// roi_align_kernel.cu:195-270 -- the reference's lane mapping (one lane per output element), sample coordinates, weights and
// atomics -- restricted to the taps that lie in rows [y0, y0 + th) x columns [x0, x0 + tw); the tiles of the map partition
// its pixels, so every tap is added exactly once.  Bins and sample rows that cannot reach the tile are skipped before
// their inner loops (a RoI of 100 x 100 window pixels is found by 28 tiles; without the test each of them would walk all
// of its samples).  Rare: windows wider than kMaxWin, more than kMaxS samples per axis, samples outside the [-1, size] band.
__device__ __noinline__ void bwd_slow_in_tile(const float* __restrict__ top_grad, const float* __restrict__ roi,
                                              float* __restrict__ bottom_grad, float spatial_scale, int height, int width,
                                              int channels, int c0, int nch, int aligned_height, int aligned_width,
                                              int sampling_ratio, bool nhwc, int x0, int y0, int th, int tw, int r, int tid,
                                              int nthreads) {
  const int bins = aligned_height * aligned_width;
  const RoiGeom g = roi_geometry(roi, spatial_scale, aligned_height, aligned_width, sampling_ratio);
  const float* __restrict__ gsrc = top_grad + ((long long)r * channels + c0) * bins;
  // element strides of (channel, pixel) in the gradient map: NCHW or channels-last
  const int cs = nhwc ? 1 : height * width, ps = nhwc ? channels : 1;
  float* gdst = bottom_grad + (long long)g.batch_ind * channels * height * width + (long long)c0 * cs;
  // rows / columns a sample at coordinate v can tap: (int)max(v, 0) and the next one, clamped to the map
  auto first_tap = [](float v, int size) { return min((int)fminf(fmaxf(v, 0.f), (float)size), size - 1); };
#pragma nounroll
  for (int i = tid; i < nch * bins; i += nthreads) {
    const int c = i / bins, bin = i - c * bins;
    const int ph = bin / aligned_width, pw = bin - ph * aligned_width;
    // the bin's samples lie in [start + p * bin, start + (p + 1) * bin] (one more pixel of margin for fp32 rounding)
    const float by = g.start_h + (float)ph * g.bin_h, bx = g.start_w + (float)pw * g.bin_w;
    if (first_tap(by + g.bin_h, height) + 2 < y0 || first_tap(by, height) - 1 >= y0 + th) continue;
    if (first_tap(bx + g.bin_w, width) + 2 < x0 || first_tap(bx, width) - 1 >= x0 + tw) continue;
    float* plane = gdst + (long long)c * cs;
    const float top_diff_this_bin = gsrc[i];
#pragma nounroll
    for (int iy = 0; iy < g.grid_h; iy++) {
      const float y = sample_y(g, ph, iy);
      const int ty = first_tap(y, height);
      if (ty + 1 < y0 || ty >= y0 + th) continue;
#pragma nounroll
      for (int ix = 0; ix < g.grid_w; ix++) {
        const float x = sample_x(g, pw, ix);
        const Taps t = sample_taps(height, width, y, x);
        if (t.y_low < 0) continue;
        const bool yl_in = t.y_low >= y0 && t.y_low < y0 + th, yh_in = t.y_high >= y0 && t.y_high < y0 + th;
        const bool xl_in = t.x_low >= x0 && t.x_low < x0 + tw, xh_in = t.x_high >= x0 && t.x_high < x0 + tw;
        if (yl_in && xl_in) atomicAdd(plane + (t.y_low * width + t.x_low) * ps, top_diff_this_bin * t.w1 / g.count);
        if (yl_in && xh_in) atomicAdd(plane + (t.y_low * width + t.x_high) * ps, top_diff_this_bin * t.w2 / g.count);
        if (yh_in && xl_in) atomicAdd(plane + (t.y_high * width + t.x_low) * ps, top_diff_this_bin * t.w3 / g.count);
        if (yh_in && xh_in) atomicAdd(plane + (t.y_high * width + t.x_high) * ps, top_diff_this_bin * t.w4 / g.count);
      }
    }
  }
}

// RoIs WITHOUT backward tables (kBoundsNoTables) never enter a list: this kernel adds their share of the tile.  Planned
// launches: BEFORE the tile kernel runs, and the tile's entry is filed in "add" mode (slices = 2: the tile kernel then adds
// its sums atomically instead of storing them); under the OVERWRITE contract the tile is zero-filled first by the role
// workgroups, each of which then adds the RoIs' share of the channels it zeroed; without it the planner adds into the
// caller's values.  (The unplanned tile kernel, which has no launch in front of it, is followed by roi_align_bwd_untabled.)
// (Until round 5 a launch of its own behind the tile kernel -- roi_align_bwd_slow, 4.7 us + a launch boundary per call,
// nearly always to find nothing -- did this and reset the counters.)
__global__ void __launch_bounds__(kPlanThreads)
roi_align_bwd_plan(int* __restrict__ ws, const float* __restrict__ top_grad, const float* __restrict__ rois, int num_rois,
                   int batch, int channels, int th, int plan_tiles, int plan_cap, int slice_len, int overwrite,
                   int aligned_height, int aligned_width, int sampling_ratio, const LevelTable lv) {
  __shared__ int wave_hits[2 * kPlanWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  // role 0 plans the tile, roles 1.. zero a share of a sliced tile; a tile's roles are neighbours in the grid (planners
  // first and 4 larger shares measured 4-10 us slower on a step's RoIs)
  const int roles = (overwrite & 1) ? 1 + kZeroParts : 1;
  const int tile_global = blockIdx.x / roles, role = blockIdx.x - tile_global * roles;
  int tile_lin = tile_global, lvl = 0;
  while (lvl + 1 < lv.count && tile_lin >= lv.tile_base[lvl + 1]) lvl++;
  tile_lin -= lv.tile_base[lvl];
  const int height = lv.height[lvl], width = lv.width[lvl];
  const int tiles_x = (width + kTW - 1) / kTW, tiles_y = (height + th - 1) / th;
  const int n = tile_lin / (tiles_x * tiles_y);
  const int trem = tile_lin - n * tiles_x * tiles_y;
  const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
  const int x0 = txi * kTW, y0 = tyi * th;
  const int img_row0 = lv.row_base[lvl] + n * height;
  const int gy0 = img_row0 + y0;
  const int4* __restrict__ bounds = reinterpret_cast<const int4*>(ws + kCounterDwords + (long long)num_rois * kRecDwords);
  int* __restrict__ counts = ws + kCounterDwords + (long long)num_rois * (kRecDwords + 4);
  int4* __restrict__ entries = reinterpret_cast<int4*>(counts + ((plan_tiles + 3) & ~3));
  const int entries_total = bwd_class_base(kBwdClasses - 1, plan_tiles, num_rois) + plan_tiles;
  unsigned short* __restrict__ list = reinterpret_cast<unsigned short*>(entries + entries_total) + (long long)tile_global * plan_cap;
  // Counter set of this call (roi_align_record_layout.h): the set the parity word names; the first workgroup zeroes the
  // other one for the next call and tells the tile kernel which set to read -- no launch exists only to reset counters.
  const int set = ((const_int_ptr)(uintptr_t)(ws + kBwdParity))[0] & 1;
  int* __restrict__ bucket = ws + kBwdBucket + set * kBwdSetStride;
  if (blockIdx.x == 0 && tid <= kBwdClasses) ws[kBwdBucket + (set ^ 1) * kBwdSetStride + tid * kBwdCounterStride] = 0;
  if (blockIdx.x == 0 && tid == 0) ws[kBwdInUse] = set;
  // each wave owns a contiguous share of the ranks: count, one barrier, then write at the wave's offset (rank order)
  const int share = (((num_rois + kPlanWaves - 1) / kPlanWaves) + 63) & ~63;
  const int r0 = wave * share, r1 = min(num_rois, r0 + share);
  // 0: the RoI at rank i does not touch the tile, 1: it does (listed), 2: it does and has no tables (added here)
  auto hit_of = [&](int i) {
    if (i >= r1) return 0;
    const int4 b = bounds[i];
    const int bx1 = b.y & (kBoundsNoTables - 1);
    const bool hit = bx1 >= x0 && b.x < x0 + kTW && b.w >= gy0 && b.z < gy0 + th && b.z >= img_row0 && b.z < img_row0 + height;
    return !hit ? 0 : (b.y & kBoundsNoTables) ? 2 : 1;
  };
  int mine = 0, mine_slow = 0;
  for (int base = r0; base < r1; base += 64) {
    const int h = hit_of(base + lane);
    mine += __popcll(__ballot(h == 1));
    mine_slow += __popcll(__ballot(h == 2));
  }
  if (lane == 0) {
    wave_hits[wave] = mine;
    wave_hits[kPlanWaves + wave] = mine_slow;
  }
  __syncthreads();
  int off = 0, total = 0, nslow = 0;
  for (int w = 0; w < kPlanWaves; w++) {
    off += w < wave ? wave_hits[w] : 0;
    total += wave_hits[w];
    nslow += wave_hits[kPlanWaves + w];
  }
  float* __restrict__ grad = lv.grad[lvl];
  // the RoIs without tables, for channels [c_lo, c_lo + nch): every wave walks its own share of the ranks
  auto add_slow = [&](int c_lo, int nch) {
    if (mine_slow == 0) return;
    for (int base = r0; base < r1; base += 64) {
      unsigned long long m = __ballot(hit_of(base + lane) == 2);
      while (m != 0ull) {
        const int pos = base + (int)__builtin_ctzll(m);
        m &= m - 1ull;
        const int r = ((const_int_ptr)(uintptr_t)(ws + kCounterDwords + (long long)pos * kRecDwords))[8];
        bwd_slow_in_tile(top_grad, rois + (long long)r * 5, grad, lv.scale[lvl], height, width, channels, c_lo, nch,
                         aligned_height, aligned_width, sampling_ratio, (overwrite & 2) != 0, x0, y0, th, kTW, r, lane, 64);
      }
    }
  };
  // zeros into channels [c_lo, c_lo + cz) of the tile
  auto zero_share = [&](int c_lo, int cz) {
    const int rows = min(th, height - y0), cols = min(kTW, width - x0);
    if (overwrite & 2) {
      // channels-last: `channels` contiguous floats per pixel
      const int c4 = cz / 4;
      for (int i = tid; i < rows * cols * c4; i += kPlanThreads) {
        const int px = i / c4, q = i - px * c4;
        const int rr = px / cols, cc = px - rr * cols;
        reinterpret_cast<float4*>(grad + (((long long)n * height + y0 + rr) * width + x0 + cc) * channels + c_lo)[q] =
            make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else if (cols == kTW && (width & 3) == 0) {
      // 128-byte row pieces, 16-byte aligned: eight lanes per piece
      for (int i = tid; i < cz * rows * 8; i += kPlanThreads) {
        const int seg = i >> 3, q = i & 7;
        const int c = seg / rows, rr = seg - c * rows;
        reinterpret_cast<float4*>(grad + (((long long)n * channels + c_lo + c) * height + y0 + rr) * width + x0)[q] =
            make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      const int col = tid & 31;
      if (col < cols)
        for (int i = tid >> 5; i < cz * rows; i += kPlanThreads / 32) {
          const int c = i / rows, rr = i - c * rows;
          grad[(((long long)n * channels + c_lo + c) * height + y0 + rr) * width + x0 + col] = 0.f;
        }
    }
  };
  if (role > 0) {
    // ---- zero this workgroup's share of the channels if the tile's sums will be ADDED: its list is cut into slices (a
    // list that does not get its slices any more -- budget used up -- is summed by one workgroup that overwrites: the
    // zeros are then wasted, not wrong), or RoIs without tables add to it ----
    if (total <= slice_len && nslow == 0) return;
    const int cz = channels / kZeroParts, c_lo = (role - 1) * cz;  // channels % 32 == 0
    zero_share(c_lo, cz);
    if (nslow > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // the zeros have reached the L2 the atomics execute in
      add_slow(c_lo, cz);
    }
    return;
  }
  if (mine > 0)
    for (int base = r0; base < r1; base += 64) {
      const bool hit = hit_of(base + lane) == 1;
      const unsigned long long m = __ballot(hit);
      if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(base + lane);
      off += __popcll(m);
    }
  if (tid == 0) {
    counts[tile_global] = total;
    const int add_mode = nslow > 0 ? 2 : 1;  // slices = 2 on a whole list: the tile kernel adds atomically
    bool whole = true;
    if (total > slice_len) {
      const int v = (total + slice_len - 1) / slice_len, per = (total + v - 1) / v;  // even slices: (v - 1) * per < total
      if (atomicAdd(bucket + kBwdClasses * kBwdCounterStride, v - 1) + v - 1 <= bwd_plan_extra(num_rois)) {
        whole = false;
        const int slot = atomicAdd(bucket, v);
        for (int sl = 0; sl < v; sl++) entries[slot + sl] = make_int4(tile_global, sl * per, min(per, total - sl * per), v);
      }
    }
    // a tile without listed RoIs is stored as zeros under the OVERWRITE contract -- unless RoIs without tables add to it:
    // then the role workgroups have zeroed it already
    if (whole && (total > 0 || ((overwrite & 1) && nslow == 0))) {
      const int c = bwd_class_of(total);
      const int slot = atomicAdd(bucket + c * kBwdCounterStride, 1);
      entries[bwd_class_base(c, plan_tiles, num_rois) + slot] = make_int4(tile_global, 0, total, add_mode);
    }
  }
  if (!(overwrite & 1) && nslow > 0) add_slow(0, channels);  // accumulate contract: into the caller's values, all channels
}

// Behind the UNPLANNED tile kernel (MI_ROI_ALIGN_BWD_SLICE=0, forward-sized workspaces, deterministic mode), whose lists
// skip the RoIs without backward tables: those RoIs, whole, with the reference's mapping and atomics -- bwd_slow_in_tile with
// the map as the tile.  One workgroup per (64 ranks, channel tile) looks at the 64 window entries and leaves at once when
// none is marked, which is nearly always.  The default (planned) path has no such launch: roi_align_bwd_plan serves them.
__global__ void __launch_bounds__(256)
roi_align_bwd_untabled(const float* __restrict__ top_grad, const float* __restrict__ rois, const LevelTable lv,
                       const int* __restrict__ ws, int num_rois, int channels, int aligned_height, int aligned_width,
                       int sampling_ratio, int nhwc) {
  const int tiles = channels / kCT;
  const int group = blockIdx.x / tiles, c0 = (blockIdx.x - group * tiles) * kCT;
  const int p = group * 64 + (threadIdx.x & 63);
  const int4* __restrict__ bounds = reinterpret_cast<const int4*>(ws + kCounterDwords + (long long)num_rois * kRecDwords);
  unsigned long long todo = __ballot(p < num_rois && (bounds[min(p, num_rois - 1)].y & kBoundsNoTables) != 0);
  while (todo != 0ull) {
    const int pos = group * 64 + (int)__builtin_ctzll(todo);
    todo &= todo - 1ull;
    const const_int_ptr rec = (const_int_ptr)(uintptr_t)(ws + kCounterDwords + (long long)pos * kRecDwords);
    const int r = rec[8], lvl = rec[11];
    bwd_slow_in_tile(top_grad, rois + (long long)r * 5, lv.grad[lvl], lv.scale[lvl], lv.height[lvl], lv.width[lvl], channels,
                     c0, kCT, aligned_height, aligned_width, sampling_ratio, nhwc != 0, 0, 0, lv.height[lvl], lv.width[lvl], r,
                     threadIdx.x, 256);
  }
}

// 16-row tiles with 32 channels: 84 VGPRs would cap a SIMD at 5 waves = two 8-wave workgroups per CU; pinning the
// kernel to 6 waves per SIMD (<= 80 VGPRs) lets the third workgroup the LDS budget allows become resident.
// kA > 0: aligned_height == aligned_width == kA at compile time (7 and 14, the sizes of the box / mask heads): the
// per-lane index arithmetic of the g block (divisions by the channel stride and the padded height, once per workgroup
// for eight DMA pieces per lane) and the loop bounds of pass 1 fold to constants.
// kNHWC: the gradient maps are stored channels-last (its store path costs the planar instance registers it does not have)
template <int kNHWC, int KC, int kTH, int kA = 0>
__global__ void __launch_bounds__(kTH * 32)
    __attribute__((amdgpu_waves_per_eu(kTH == 16 && KC == 32 ? 6 : 1, kTH == 16 && KC == 32 ? 6 : 8)))
roi_align_bwd_tiles(const float* __restrict__ top_grad, int* __restrict__ ws, int num_rois, int batch, int channels,
                    int aligned_height_arg, int aligned_width_arg, int overwrite, int ablate_arg, int plan_tiles, int plan_cap,
                    const LevelTable lv MI_TL_PARAM) {
  const int aligned_height = kA > 0 ? kA : aligned_height_arg, aligned_width = kA > 0 ? kA : aligned_width_arg;
  // the block of top gradients of one RoI and KC channels, as it lies in memory: [c][ph][pw], KC * bins floats
  const int g_words = (KC * aligned_height * aligned_width + 3) & ~3;
  const int ablate = MI_ABLATE(ablate_arg);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kTabDw = BwdLds<KC>::kTabDw;
  constexpr int kThreads = kTH * kTW, kNWaves = kThreads / 64;
  constexpr int kCS = KC + 4;  // words per (bin row, column) of T
  // LDS (all of it in the dynamic region, 16-byte aligned pieces):
  //   ctl[32] | list[num_rois] (16-bit ranks) | tab[2][kTabDw] | g[2][g_words] | T[aligned_height][kTW][KC + 4]
  int* wave_count = reinterpret_cast<int*>(smem);
  int& list_len = wave_count[kNWaves];
  unsigned short* list = reinterpret_cast<unsigned short*>(wave_count + 32);  // ranks < kMaxRois = 8192
  const int list_words = ((num_rois + 1) / 2 + 3) & ~3;
  int* tab0 = wave_count + 32 + list_words;
  float* g0 = reinterpret_cast<float*>(tab0 + 2 * kTabDw);
  float* T = g0 + 2 * g_words;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int bins = aligned_height * aligned_width;
  const int ncg = channels / KC;
  const int* __restrict__ records = ws + kCounterDwords;
  const int4* __restrict__ bounds = reinterpret_cast<const int4*>(ws + kCounterDwords + (long long)num_rois * kRecDwords);
  // Planned launch (roi_align_bwd_plan ran before): blockIdx / ncg is the e-th ENTRY in class order (longest lists first:
  // the hardware hands out workgroups in blockIdx order); the grid is an upper bound of the entry count.
  const int cg = blockIdx.x % ncg;
  int tile_lin = blockIdx.x / ncg;
  int nslices = 1, planned_len = 0, planned_first = 0;
  const int4* __restrict__ entries =
      reinterpret_cast<const int4*>(ws + kCounterDwords + (long long)num_rois * (kRecDwords + 4) + ((plan_tiles + 3) & ~3));
  if (plan_tiles > 0) {
    // both counter sets and the word that says which one this call filed into arrive together (no dependent scalar load);
    // the first workgroup hands the OTHER set to the next backward over this workspace
    const const_int_ptr bc = (const_int_ptr)(uintptr_t)(ws + kBwdBucket);
    const int set = bc[kBwdInUse - kBwdBucket] & 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) ws[kBwdParity] = set ^ 1;
    const int e = tile_lin;
    int c = 0, first = 0, run = 0;
#pragma unroll
    for (int k = 0; k < kBwdClasses; k++) {
      const int cnt0 = bc[k * kBwdCounterStride], cnt1 = bc[kBwdSetStride + k * kBwdCounterStride];
      const int cnt = set ? cnt1 : cnt0;
      run += cnt;
      if (k + 1 < kBwdClasses && e >= run) {
        c = k + 1;
        first = run;
      }
    }
    if (e >= run) return;
    const const_int_ptr it = (const_int_ptr)(uintptr_t)(entries + bwd_class_base(c, plan_tiles, num_rois) + (e - first));
    tile_lin = it[0];
    planned_first = it[1];
    planned_len = it[2];
    nslices = it[3];
  }
#if MI_TUNING
  // tuning builds: wave 0 sums clock64() differences per phase over the visits of this workgroup (tools/timeline_bwd.py)
  long long tl_t = clock64(), tl_acc[5] = {0, 0, 0, 0, 0};
  const long long tl_start = tl_t;
#define MI_TL_LAP(k)                          \
  do {                                        \
    const long long tl_now = clock64();       \
    tl_acc[k] += tl_now - tl_t;               \
    tl_t = tl_now;                            \
  } while (0)
#else
#define MI_TL_LAP(k) \
  do {               \
  } while (0)
#endif
  const int tile_global = tile_lin;
  int lvl = 0;  // the level this tile belongs to (tiles of all levels share the grid in an FPN-fused call)
  while (lvl + 1 < lv.count && tile_lin >= lv.tile_base[lvl + 1]) lvl++;
  tile_lin -= lv.tile_base[lvl];
  float* __restrict__ bottom_grad = lv.grad[lvl];
  const int height = lv.height[lvl], width = lv.width[lvl];
  const int tiles_x = (width + kTW - 1) / kTW, tiles_y = (height + kTH - 1) / kTH;
  const int n = tile_lin / (tiles_x * tiles_y);
  const int trem = tile_lin - n * tiles_x * tiles_y;
  const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
  const int x0 = txi * kTW, y0 = tyi * kTH;            // tile origin in the feature map
  const int img_row0 = lv.row_base[lvl] + n * height;  // "global rows": levels and images stacked
  const int gy0 = img_row0 + y0;
  const int c0 = cg * KC;

  // ---- RoIs whose window touches this tile, in rank order ----
  if (tid == 0) list_len = planned_len;
  if (plan_tiles > 0) {
    const unsigned short* __restrict__ lists =
        reinterpret_cast<const unsigned short*>(entries + bwd_class_base(kBwdClasses - 1, plan_tiles, num_rois) + plan_tiles);
    for (int i = tid; i < planned_len; i += kThreads) list[i] = lists[(long long)tile_global * plan_cap + planned_first + i];
  }
  __syncthreads();
  for (int base = 0; plan_tiles == 0 && base < num_rois; base += kThreads) {
    const int i = base + tid;
    bool hit = false;
    if (i < num_rois) {
      const int4 b = bounds[i];
      hit = (b.y & (kBoundsNoTables - 1)) >= x0 && b.x < x0 + kTW && b.w >= gy0 && b.z < gy0 + kTH && b.z >= img_row0 &&
            b.z < img_row0 + height;
      // a RoI without tables is not listed: the launch behind this one (roi_align_bwd_untabled) adds it to what the tiles store
      if (b.y & kBoundsNoTables) hit = false;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_count[wave] = __popcll(m);
    __syncthreads();
    int off = list_len;
    for (int w = 0; w < wave; w++) off += wave_count[w];
    if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
    __syncthreads();
    if (tid == 0) {
      int add = 0;
      for (int w = 0; w < kNWaves; w++) add += wave_count[w];
      list_len += add;
    }
    __syncthreads();
  }
  const int nlist = uniform(list_len);

  // pass-2 identity of this lane: one pixel of the tile
  const int prow = tid >> 5, pcol = tid & 31;
  v2f acc2[KC / 2];  // channel pairs: pass 2 runs on v_pk_fma_f32
#pragma unroll
  for (int c = 0; c < KC / 2; c++) acc2[c] = v2f{0.f, 0.f};

  // LDS-DMA of RoI `pos`: tables + xfirst/yfirst (contiguous in the record) and the [KC][bins] block of top gradients
  auto issue_loads = [&](int pos, int buf) {
    const const_int_ptr rec = (const_int_ptr)(uintptr_t)(records + (long long)uniform(pos) * kRecDwords);
    const int r = rec[8];
    const srd_t tsrd = make_srd(records + (long long)uniform(pos) * kRecDwords + kRecB, kTabDw * 4);
    const unsigned tdst = lds_addr_uniform(tab0 + buf * kTabDw);
    // 16-byte pieces (the launcher checks the alignment): the per-visit cost is the number of DMA instructions, not bytes
    for (int k = wave; k * 64 < kTabDw / 4; k += kNWaves)
      if (k * 64 + lane < kTabDw / 4) dma_dwordx4(tsrd, tdst + (unsigned)k * 1024u, (unsigned)(k * 64 + lane) * 16u, 0u);
    // top gradients: the [KC][bins] block is one contiguous run of KC * bins floats and lands in LDS as it lies in memory
    // ([c][ph][pw], channel stride bins: odd for 7 x 7, 4 mod 64 for 14 x 14 -- the KC channel lanes of pass 1 hit distinct
    // banks).  16-byte LDS-DMA pieces (the run is 16-byte aligned whenever top_grad is; the launcher checks): 7 per visit
    // for the box head instead of the 30 dword pieces of the transposing gather used until round 4 -- the visits are bound
    // by the DMA INSTRUCTIONS the compute unit's address path takes (profiles/r04_records_timeline.txt), and this took
    // 5-8 us off every backward (config 2: 60.3 -> 55.3 us planned, 52.5 -> 47.1 unplanned; step RoIs 108.8 -> 103.3 / 76.1
    // -> 67.8 us, alternating same-box runs).
    const srd_t gsrd = make_srd(top_grad + ((long long)r * channels + c0) * bins, (unsigned)(KC * bins) * 4u);
    const unsigned gdst = lds_addr_uniform(g0 + buf * g_words);
    const int total16 = KC * bins / 4;
    for (int k = wave; k * 64 < total16; k += kNWaves)
      if (k * 64 + lane < total16) dma_dwordx4(gsrd, gdst + (unsigned)k * 1024u, (unsigned)(k * 64 + lane) * 16u, 0u);
  };

  if (nlist > 0) issue_loads(list[0], 0);
  MI_TL_LAP(4);  // list + first issue (not per visit)
  for (int li = 0; li < nlist; li++) {
    const int buf = li & 1;
    const int pos = uniform(list[li]);
    const const_int_ptr rec = (const_int_ptr)(uintptr_t)(records + (long long)pos * kRecDwords);
    const int wx0 = rec[2], ww = rec[3], wy0 = rec[9], wy1 = rec[10];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // B1: buffers of this RoI have landed; everybody is done with the previous RoI (T, other buffer)
    MI_TL_LAP(0);  // wait for the landing + barrier B1
    if (li + 1 < nlist) issue_loads(list[li + 1], buf ^ 1);
    MI_TL_LAP(1);  // issue of the next RoI's pieces
    const char* blk = reinterpret_cast<const char*>(tab0 + buf * kTabDw);
    const float* wxt = reinterpret_cast<const float*>(blk + kBwdWx);
    const float* wyt = reinterpret_cast<const float*>(blk + kBwdWy);
    const unsigned char* pxt = reinterpret_cast<const unsigned char*>(blk + kBwdPx);
    const unsigned char* pyt = reinterpret_cast<const unsigned char*>(blk + kBwdPy);
    const unsigned char* cft = reinterpret_cast<const unsigned char*>(blk + kBwdCf);
    const unsigned char* rft = reinterpret_cast<const unsigned char*>(blk + kBwdRf);
    const float* g = g0 + buf * g_words;

    // ---- pass 1: T[ph][col][c] for the tile columns inside the window ----
    if (!(ablate & 1)) {
      const int c = tid % KC, slot = tid / KC;
      constexpr int kColStep = kThreads / KC;
      for (int col = slot; col < kTW; col += kColStep) {
        const int lc = x0 + col - wx0;
        if (lc < 0 || lc >= ww) continue;
        const int k0 = cft[lc], k1 = cft[lc + 1];  // the output columns that reach this column, with their weights
        const float* gc = g + c * bins;
        for (int ph0 = 0; ph0 < aligned_height; ph0 += 8) {
          v2f t[4];  // bin rows ph0 .. ph0 + 7, two per register pair: v_pk_fma_f32
#pragma unroll
          for (int j = 0; j < 4; j++) t[j] = v2f{0.f, 0.f};
          for (int k = k0; k < k1; k++) {
            const v2f wgt = {wxt[k], wxt[k]};
            const int pw = pxt[k];
            const float* gq = gc + ph0 * aligned_width + pw;  // bin rows of output column pw: stride aligned_width
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float a = (ph0 + 2 * j < aligned_height) ? gq[(2 * j) * aligned_width] : 0.f;
              const float b = (ph0 + 2 * j + 1 < aligned_height) ? gq[(2 * j + 1) * aligned_width] : 0.f;
              t[j] = __builtin_elementwise_fma(wgt, v2f{a, b}, t[j]);
            }
          }
#pragma unroll
          for (int j = 0; j < 8; j++)
            if (ph0 + j < aligned_height) T[((ph0 + j) * kTW + col) * kCS + c] = t[j >> 1][j & 1];
        }
      }
    }
    __syncthreads();  // B2: T complete
    MI_TL_LAP(2);  // pass 1 + barrier B2

    // ---- pass 2: this lane's pixel, all KC channels ----
    if (!(ablate & 2)) {
      const int lr = y0 + prow - wy0, lc = x0 + pcol - wx0;
      if (lr >= 0 && lr <= wy1 - wy0 && lc >= 0 && lc < ww) {
        const int k0 = rft[lr], k1 = rft[lr + 1];  // the bin rows that reach this feature row, with their weights
        for (int k = k0; k < k1; k++) {
          const v2f wgt = {wyt[k], wyt[k]};
          const int ph = pyt[k];
          // T is [ph][col][channel] with a column stride of KC + 4 words: the lane's KC channels are KC / 4
          // conflict-free ds_read_b128 (16-lane groups land on 16 distinct 4-bank slots)
          const float4* tp = reinterpret_cast<const float4*>(T + (ph * kTW + pcol) * kCS);
#pragma unroll
          for (int c4 = 0; c4 < KC / 4; c4++) {
            const float4 tv = tp[c4];
            acc2[2 * c4 + 0] = __builtin_elementwise_fma(wgt, v2f{tv.x, tv.y}, acc2[2 * c4 + 0]);
            acc2[2 * c4 + 1] = __builtin_elementwise_fma(wgt, v2f{tv.z, tv.w}, acc2[2 * c4 + 1]);
          }
        }
      }
    }
    MI_TL_LAP(3);  // pass 2
  }

#if MI_TUNING
  if (timeline != nullptr && threadIdx.x == 0) {
    long long* o = timeline + (long long)blockIdx.x * 8;
    o[0] = tl_acc[0]; o[1] = tl_acc[1]; o[2] = tl_acc[2]; o[3] = tl_acc[3]; o[4] = tl_acc[4];
    o[5] = clock64() - tl_start; o[6] = nlist; o[7] = nslices;
  }
#endif
  // ---- the tile leaves as 128-byte rows (NCHW) or as one 4*KC-byte run per pixel (channels-last) ----
  float acc[KC];
#pragma unroll
  for (int c = 0; c < KC; c++) acc[c] = acc2[c >> 1][c & 1];
  const int row = y0 + prow, col = x0 + pcol;
  // Channels-last, whole list (no atomics): a lane holds ITS pixel's KC channels, and storing them from there is KC / 4
  // 16-byte stores per lane, each instruction touching 64 cache lines (the pixels lie channels * 4 bytes apart).  The sums go
  // through LDS instead -- [pixel][KC + 4] over the g / T buffers nobody reads any more, as many tile rows per pass as fit --
  // and leave with KC / 4 neighbouring lanes on one pixel's 4 * KC-byte run: whole lines per store instruction.
  const int xpose_rows = min((int)kTH, (2 * g_words + aligned_height * kTW * kCS) / (kTW * kCS));
  if (kNHWC && nslices == 1 && xpose_rows > 0 && !(ablate & 4)) {
    float* X = g0;
    constexpr int kQ = KC / 4;
    for (int r0 = 0; r0 < kTH; r0 += xpose_rows) {
      __syncthreads();  // the visit loop (first pass) / the previous pass is done with the buffer
      if (prow >= r0 && prow < r0 + xpose_rows) {
        float4* xp = reinterpret_cast<float4*>(X + ((prow - r0) * kTW + pcol) * kCS);
#pragma unroll
        for (int c4 = 0; c4 < kQ; c4++) xp[c4] = make_float4(acc[4 * c4 + 0], acc[4 * c4 + 1], acc[4 * c4 + 2], acc[4 * c4 + 3]);
      }
      __syncthreads();
      const int rows_here = min(xpose_rows, (int)kTH - r0);
      for (int i = tid; i < rows_here * kTW * kQ; i += kThreads) {
        const int px = i / kQ, q = i - px * kQ;
        const int rr = y0 + r0 + px / kTW, cc = x0 + (px & (kTW - 1));
        if (rr >= height || cc >= width) continue;
        float4 v = reinterpret_cast<const float4*>(X + px * kCS)[q];
        float4* dst = reinterpret_cast<float4*>(bottom_grad + (((long long)n * height + rr) * width + cc) * channels + c0) + q;
        if (!(overwrite & 1)) {
          const float4 o = *dst;
          v.x += o.x;
          v.y += o.y;
          v.z += o.z;
          v.w += o.w;
        }
        *dst = v;
      }
    }
  } else if (row < height && col < width && nslices > 1) {
    // the slices of a long list add into the tile the plan kernel zero-filled (or the caller's values): hardware fp32
    // atomics, the order of the slices' sums is not fixed (the reference's backward is atomic throughout)
    const long long cs = kNHWC ? 1 : (long long)height * width;
    float* dst = kNHWC ? bottom_grad + (((long long)n * height + row) * width + col) * channels + c0
                                 : bottom_grad + (((long long)n * channels + c0) * height + row) * width + col;
#pragma unroll
    for (int c = 0; c < KC; c++)
      if (acc[c] != 0.f) atomicAdd(dst + c * cs, acc[c]);
  } else if (row < height && col < width) {
    if (kNHWC) {
      float4* dst = reinterpret_cast<float4*>(bottom_grad + (((long long)n * height + row) * width + col) * channels + c0);
#pragma unroll
      for (int c4 = 0; c4 < KC / 4; c4++) {
        float4 v = make_float4(acc[4 * c4 + 0], acc[4 * c4 + 1], acc[4 * c4 + 2], acc[4 * c4 + 3]);
        if (!(overwrite & 1)) {
          const float4 o = dst[c4];
          v.x += o.x;
          v.y += o.y;
          v.z += o.z;
          v.w += o.w;
        }
        dst[c4] = v;
      }
    } else {
      float* dst = bottom_grad + (((long long)n * channels + c0) * height + row) * width + col;
      const long long plane = (long long)height * width;
#pragma unroll
      for (int c = 0; c < KC; c++) {
        if (overwrite & 1)
          dst[c * plane] = acc[c];  // non-temporal stores measured equal here (107.9 / 61.0 us either way)
        else
          dst[c * plane] += acc[c];
      }
    }
  }
#undef MI_TL_LAP
}

size_t records_lds_bytes(int cap, int ct) {
  return 2 * kMaxS * sizeof(TabEntry) + (size_t)(ct * (kTileBins + 1) + ct * (cap | 1)) * 4;
}

template <class Src>
int launch_prepare_from(const Src& src, int* ws, int batch, const LevelTable& lv, int num_rois, int aligned_height,
                        int aligned_width, int sampling_ratio, int cap_px, bool bwd_tables, hipStream_t stream, int channels,
                        bool cost_in_band = false) {
  const int max_rows_tile = kTileBins / aligned_width;
  roi_align_prepare<Src><<<(num_rois + 3) / 4, 256, (size_t)num_rois * sizeof(unsigned), stream>>>(
      ws, num_rois, batch, aligned_height, aligned_width, sampling_ratio, cap_px, cap_px, max_rows_tile, bwd_tables ? 1 : 0,
      channels, tuning().ablate, cost_in_band ? 1 : 0, src, lv);
  return check_launch("roi_align_prepare");
}
int launch_prepare(const float* rois, const int* levels, int* ws, int batch, const LevelTable& lv, int num_rois,
                   int aligned_height, int aligned_width, int sampling_ratio, int cap_px, bool bwd_tables,
                   hipStream_t stream, int channels = 0, bool cost_in_band = false) {
  return launch_prepare_from(PlainRois{rois, levels}, ws, batch, lv, num_rois, aligned_height, aligned_width, sampling_ratio,
                             cap_px, bwd_tables, stream, channels, cost_in_band);
}

#if MI_TUNING
long long* g_records_timeline = nullptr;
#endif

template <int kCap>
int launch_cap(const LevelTable& lv, const float* rois, const int* levels, float* output, int* ws, int batch,
               int channels, int num_rois, int aligned_height, int aligned_width, int sampling_ratio, bool bwd_tables,
               hipStream_t stream, bool records_ready) {
  // records_ready: the producer of the RoIs wrote the records (launch_roi_align_prepare_collected, same LDS capacity)
  int rc = records_ready ? MI_OK
                         : launch_prepare(rois, levels, ws, batch, lv, num_rois, aligned_height, aligned_width, sampling_ratio,
                                          kCap, bwd_tables, stream, channels, /*cost_in_band=*/levels == nullptr);
  if (rc != MI_OK) return rc;
  const size_t lds = records_lds_bytes(kCap, kCT);
  // An item has at least ceil(aligned_height / rows per output tile) stages (four at 14x14).  A launch of few such items
  // -- the mask / keypoint heads: 128 RoIs = 1024 items over the chip's 768 slots -- deals every item's stages to two
  // workgroups: 128 x 14x14 30.5 -> 28.7 us, a step's mask RoIs 37.4 -> 37.0 (four: 29.8 / 39.8).  Bit-equal: a stage is
  // computed by the same code either way.  MI_ROI_ALIGN_FWD_SPLIT = 1 / 2 / 4 overrides.
  const int min_stages = (aligned_height + kTileBins / aligned_width - 1) / (kTileBins / aligned_width);
  int split = tuning().fwd_split;
  if (split <= 0) split = (min_stages >= 2 && num_rois * (channels / kCT) <= 2048) ? 2 : 1;
  if (split > min_stages) split = min_stages;
  const int items = num_rois * (channels / kCT) * split;
  // the LDS images of MI_ROI_ALIGN_CAP >= 448 exceed the 64 KB a kernel may ask for without opting in
#define MI_LAUNCH_REC(SR, A)                                                                                          \
  do {                                                                                                                \
    if (lds > 64 * 1024)                                                                                              \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&roi_align_fwd_records<SR, kCap, A>),                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
    roi_align_fwd_records<SR, kCap, A><<<items, kThreads, lds, stream>>>(                                             \
        rois, output, ws, num_rois, batch, channels, aligned_height, aligned_width, sampling_ratio, split,            \
        tuning().ablate, tuning().fwd_full_wait, lv MI_TL_ARG);                                                       \
  } while (0)
  const int a = aligned_height == aligned_width ? aligned_height : 0;
  if (sampling_ratio == 2 && kCap == 336 && a == 7)
    MI_LAUNCH_REC(2, (kCap == 336 ? 7 : 0));
  else if (sampling_ratio == 2 && kCap == 336 && a == 14)
    MI_LAUNCH_REC(2, (kCap == 336 ? 14 : 0));
  else if (sampling_ratio == 2)
    MI_LAUNCH_REC(2, 0);
  else
    MI_LAUNCH_REC(0, 0);
#undef MI_LAUNCH_REC
  return check_launch("roi_align_fwd_records");
}

}  // namespace

// Backward over the records.  `lv` carries the gradient map of every level (grad[], height[], width[], scale[],
// row_base[]); tile_base[] is filled here.  nhwc: the gradient maps are stored channels-last.
namespace {
int bwd_tile_count(LevelTable& lv, int batch, int th) {
  lv.tile_base[0] = 0;
  for (int l = 0; l < lv.count; l++)
    lv.tile_base[l + 1] = lv.tile_base[l] + ((lv.width[l] + kTW - 1) / kTW) * ((lv.height[l] + th - 1) / th) * batch;
  return lv.tile_base[lv.count];
}
// the plan region behind the records and bounds: one count per tile, the entry arrays of the six cost classes (int4 {tile,
// first, length, slices of the tile}: roi_align_bwd_plan), one list of 16-bit ranks per tile
int bwd_plan_entries(int tiles, int num_rois) { return bwd_class_base(kBwdClasses - 1, tiles, num_rois) + tiles; }
size_t bwd_plan_bytes(int tiles, int num_rois) {
  return (size_t)((tiles + 3) & ~3) * 4 + (size_t)bwd_plan_entries(tiles, num_rois) * 16 +
         (size_t)tiles * ((num_rois + 7) & ~7) * 2;
}
}  // namespace

size_t roi_align_bwd_workspace_bytes(LevelTable lv, int batch, int num_rois) {
  const size_t rec = roi_align_records_workspace_bytes(num_rois);
  if (num_rois <= 0 || batch <= 0 || tuning().bwd_slice <= 0) return rec;
  const int tiles = bwd_tile_count(lv, batch, tuning().bwd_tile_rows);
  return tiles <= kMaxPlanTiles ? rec + bwd_plan_bytes(tiles, num_rois) : rec;
}

int launch_roi_align_bwd_records_levels(const float* top_grad, const float* rois, const int* levels, LevelTable lv,
                                        void* workspace, size_t workspace_bytes, bool records_ready, bool overwrite,
                                        bool nhwc, int batch,
                                        int channels, int num_rois, int aligned_height, int aligned_width,
                                        int sampling_ratio, int cap_px, hipStream_t stream) {
  int* ws = static_cast<int*>(workspace);
  // the forward writes the record's backward block only into a workspace with room for a backward (it costs the
  // records launch 2 us): RECORDS_READY is honoured for such a workspace, otherwise the records are rewritten here
  if (records_ready && workspace_bytes < roi_align_bwd_workspace_bytes(lv, batch, num_rois)) records_ready = false;
  if (!records_ready) {
    // backward tables do not depend on the LDS capacity the forward stages were cut for
    int rc = launch_prepare(rois, levels, ws, batch, lv, num_rois, aligned_height, aligned_width, sampling_ratio,
                            cap_px >= 336 ? 336 : 192, true, stream);
    if (rc != MI_OK) return rc;
  }
  const int bins = aligned_height * aligned_width;
  const int th = tuning().bwd_tile_rows;  // rows per tile (16; 8 and 32 measured slower): 32 * th lanes per workgroup
  const int g_ablate_p = tuning().ablate;
  // channels per workgroup: 32 accumulators per lane while the g block and T fit LDS comfortably, else 16
  const int kc = (bins <= 64) ? 32 : 16;
  const int g_words = (kc * bins + 3) & ~3;  // the gradient block of one RoI and kc channels, natural layout
  const int tab_dw = kBwdTabDw;
  const size_t lds = (32 + (size_t)(((num_rois + 1) / 2 + 3) & ~3) + 2 * tab_dw + 2 * g_words + (size_t)aligned_height * kTW * (kc + 4)) * 4;
  const int tiles = bwd_tile_count(lv, batch, th);
  // planned launch (see roi_align_bwd_plan) when the workspace has room for the plan: the grid is an upper bound of the
  // number of list slices (every tile once + room for extra slices of the long lists)
  const int slice_min = tuning().bwd_slice;
  const bool planned = slice_min > 0 && tiles <= kMaxPlanTiles && tiles < 65536 &&
                       workspace_bytes >= roi_align_records_workspace_bytes(num_rois) + bwd_plan_bytes(tiles, num_rois);
  const int plan_cap = (num_rois + 7) & ~7;
  int grid = tiles * (channels / kc);
  if (planned) {
    // files every tile's list AND adds the RoIs without backward tables (rare, but only the device knows whether there are
    // any): the trailing roi_align_bwd_slow launch of rounds 1-5 is gone from this, the default, path
    roi_align_bwd_plan<<<tiles * (overwrite ? 1 + kZeroParts : 1), kPlanThreads, 0, stream>>>(
        ws, top_grad, rois, num_rois, batch, channels, th, tiles, plan_cap, slice_min, (overwrite ? 1 : 0) | (nhwc ? 2 : 0),
        aligned_height, aligned_width, sampling_ratio, lv);
    int rc = check_launch("roi_align_bwd_plan");
    if (rc != MI_OK) return rc;
    grid = (tiles + bwd_plan_extra(num_rois)) * (channels / kc);  // upper bound of the entries: every tile once + the budget of extra slices
  }
#define MI_LAUNCH_TILES_A(SR, KC, TH, A) /* SR: 1 = channels-last gradient maps */                                      \
  do {                                                                                                                \
    if (lds > 64 * 1024)                                                                                              \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&roi_align_bwd_tiles<SR, KC, TH, A>),                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
    roi_align_bwd_tiles<SR, KC, TH, A><<<grid, TH * 32, lds, stream>>>(                                              \
        top_grad, ws, num_rois, batch, channels, aligned_height, aligned_width,                                       \
        (overwrite ? 1 : 0) | (nhwc ? 2 : 0), g_ablate_p & 7, planned ? tiles : 0, plan_cap, lv MI_TL_ARG);          \
  } while (0)
#define MI_LAUNCH_TILES_TH(SR, KC, TH)                                                                                \
  do {                                                                                                                \
    if (TH == 16 && KC == 32 && aligned_height == 7 && aligned_width == 7)                                            \
      MI_LAUNCH_TILES_A(SR, KC, TH, (TH == 16 && KC == 32 ? 7 : 0));                                                  \
    else if (TH == 16 && KC == 16 && aligned_height == 14 && aligned_width == 14)                                     \
      MI_LAUNCH_TILES_A(SR, KC, TH, (TH == 16 && KC == 16 ? 14 : 0));                                                 \
    else                                                                                                              \
      MI_LAUNCH_TILES_A(SR, KC, TH, 0);                                                                               \
  } while (0)
#define MI_LAUNCH_TILES(SR, KC)                                                                                       \
  do {                                                                                                                \
    if (th == 32)                                                                                                     \
      MI_LAUNCH_TILES_TH(SR, KC, 32);                                                                                 \
    else if (th == 16)                                                                                                \
      MI_LAUNCH_TILES_TH(SR, KC, 16);                                                                                 \
    else                                                                                                              \
      MI_LAUNCH_TILES_TH(SR, KC, 8);                                                                                  \
  } while (0)
  if (g_ablate_p & 8) {
  } else if (kc == 32 && !nhwc) {
    MI_LAUNCH_TILES(0, 32);
  } else if (kc == 32) {
    MI_LAUNCH_TILES(1, 32);
  } else if (!nhwc) {
    MI_LAUNCH_TILES(0, 16);
  } else {
    MI_LAUNCH_TILES(1, 16);
  }
#undef MI_LAUNCH_TILES
#undef MI_LAUNCH_TILES_TH
#undef MI_LAUNCH_TILES_A
  int rc = check_launch("roi_align_bwd_tiles");
  if (rc != MI_OK || planned) return rc;
  roi_align_bwd_untabled<<<((num_rois + 63) / 64) * (channels / kCT), 256, 0, stream>>>(
      top_grad, rois, lv, ws, num_rois, channels, aligned_height, aligned_width, sampling_ratio, nhwc ? 1 : 0);
  return check_launch("roi_align_bwd_untabled");
}

int launch_roi_align_bwd_records(const float* top_grad, const float* rois, float* bottom_grad, void* workspace,
                                 size_t workspace_bytes, bool records_ready, bool overwrite, bool nhwc, int batch,
                                 int channels, int height,
                                 int width, int num_rois, int aligned_height, int aligned_width, float spatial_scale,
                                 int sampling_ratio, int cap_px, hipStream_t stream) {
  return launch_roi_align_bwd_records_levels(top_grad, rois, nullptr,
                                             single_level(nullptr, bottom_grad, batch, height, width, spatial_scale),
                                             workspace, workspace_bytes, records_ready, overwrite, nhwc, batch, channels,
                                             num_rois,
                                             aligned_height, aligned_width, sampling_ratio, cap_px, stream);
}

bool roi_align_bwd_records_supported(int channels, int height, int width, int num_rois, int aligned_height,
                                     int aligned_width) {
  const int bins = aligned_height * aligned_width;
  const int kc = (bins <= 64) ? 32 : 16;
  const int tab_dw = kBwdTabDw;
  const size_t lds = (32 + (size_t)(((num_rois + 1) / 2 + 3) & ~3) + 2 * tab_dw + 2 * (size_t)((kc * bins + 3) & ~3) +
                      (size_t)aligned_height * kTW * (kc + 4)) * 4;
  return channels > 0 && channels % 32 == 0 && aligned_height > 0 && aligned_width > 0 &&
         aligned_height <= kMaxStages && num_rois <= kMaxRois && lds <= 160 * 1024 - 4096;
}

int launch_roi_align_prepare(const float* rois, void* workspace, int batch, int height, int width, int num_rois,
                             int aligned_height, int aligned_width, float spatial_scale, int sampling_ratio,
                             bool bwd_tables, hipStream_t stream, int channels, const float* features) {
  return launch_prepare(rois, nullptr, static_cast<int*>(workspace), batch,
                        single_level(features, nullptr, batch, height, width, spatial_scale), num_rois, aligned_height,
                        aligned_width, sampling_ratio, 336, bwd_tables, stream, channels);
}

int launch_roi_align_prepare_levels(const LevelTable& lv, const float* rois, const int* levels, void* workspace,
                                    int batch, int num_rois, int aligned_height, int aligned_width, int sampling_ratio,
                                    bool bwd_tables, hipStream_t stream, int channels) {
  return launch_prepare(rois, levels, static_cast<int*>(workspace), batch, lv, num_rois, aligned_height, aligned_width,
                        sampling_ratio, 336, bwd_tables, stream, channels);
}

int launch_roi_align_prepare_collected(const LevelTable& lv, const float* top_scores, const long long* top_idx,
                                       const float* cand_rois, int mark_invalid, int k_min, int k_max, float s0, float lvl0,
                                       float* rois, unsigned char* valid, int* levels, void* workspace, int batch,
                                       int num_rois, int aligned_height, int aligned_width, int sampling_ratio, int cap_px,
                                       bool bwd_tables, hipStream_t stream, int channels) {
  const CollectedRois src = {top_scores, top_idx, cand_rois, mark_invalid, k_min, k_max, s0, lvl0, rois, valid, levels};
  // the forward cuts its stages for one of five LDS capacities: the same rounding as launch_roi_align_fwd_records_levels
  const int cap = cap_px >= 640 ? 640 : cap_px >= 448 ? 448 : cap_px >= 336 ? 336 : cap_px >= 256 ? 256 : 192;
  return launch_prepare_from(src, static_cast<int*>(workspace), batch, lv, num_rois, aligned_height, aligned_width,
                             sampling_ratio, cap, bwd_tables, stream, channels);
}

void roi_align_fwd_records_set_timeline(long long* device_buffer) {
#if MI_TUNING
  g_records_timeline = device_buffer;
#else
  (void)device_buffer;
#endif
}

size_t roi_align_records_workspace_bytes(int num_rois) {
  return ((size_t)kCounterDwords + (size_t)(num_rois > 0 ? num_rois : 0) * (kRecDwords + 4)) * sizeof(int);
}

bool roi_align_fwd_records_supported(int channels, int height, int width, int num_rois, int aligned_height,
                                     int aligned_width) {
  return channels > 0 && channels % kCT == 0 && channels / kCT <= kCounterDwords && aligned_width <= kTileBins &&
         aligned_height > 0 && aligned_width > 0 && num_rois <= kMaxRois &&
         (long long)kCT * height * width * 4 < (1LL << 31);
}

int launch_roi_align_fwd_records_levels(const LevelTable& lv, const float* rois, const int* levels, float* output,
                                        void* workspace, int batch, int channels, int num_rois, int aligned_height,
                                        int aligned_width, int sampling_ratio, int cap_px, bool bwd_tables,
                                        hipStream_t stream, bool records_ready) {
  int* ws = static_cast<int*>(workspace);
#define MI_CAP(C)                                                                                                     \
  return launch_cap<C>(lv, rois, levels, output, ws, batch, channels, num_rois, aligned_height, aligned_width,        \
                       sampling_ratio, bwd_tables, stream, records_ready)
  if (cap_px >= 640) MI_CAP(640);
  if (cap_px >= 448) MI_CAP(448);
  if (cap_px >= 336) MI_CAP(336);
  if (cap_px >= 256) MI_CAP(256);
  MI_CAP(192);
#undef MI_CAP
}

bool roi_align_fwd_slab_supported(const LevelTable& lv, int channels, int num_rois, int aligned_height, int aligned_width) {
  if (tuning().slab <= 0) return false;
  for (int l = 0; l < lv.count; l++)
    if ((long long)kSlabCT * lv.height[l] * lv.width[l] * 4 >= (1LL << 31)) return false;
  return channels > 0 && channels % kSlabCT == 0 && aligned_width <= kTileBins && aligned_height > 0 && aligned_width > 0 &&
         num_rois > 0 && num_rois <= (1 << 24) && (channels / kSlabCT + 7) / 8 <= 65535;
}

int launch_roi_align_fwd_slab(const LevelTable& lv, const float* rois, const int* levels, float* output, int batch,
                              int channels, int num_rois, int aligned_height, int aligned_width, int sampling_ratio,
                              hipStream_t stream) {
  const dim3 grid((unsigned)num_rois * 8u, (unsigned)((channels / kSlabCT + 7) / 8));
  const int cap = tuning().slab >= 64 ? tuning().slab : 0;
  const int a = aligned_height == aligned_width ? aligned_height : 0;
#define MI_LAUNCH_SLAB(SR, CAP, A, LV)                                                                                 \
  roi_align_fwd_slab<SR, CAP, A, LV><<<grid, 64, slab_lds_bytes<SR, CAP, A>(), stream>>>(                              \
      rois, output, levels, num_rois, batch, channels, aligned_height, aligned_width, sampling_ratio, tuning().fwd_full_wait, lv MI_TL_ARG)
#define MI_SLAB_PICK(CAP7, CAP14, CAPG)                                                                                \
  do {                                                                                                                \
    if (sampling_ratio == 2 && a == 7 && levels == nullptr)                                                           \
      MI_LAUNCH_SLAB(2, CAP7, 7, false);                                                                              \
    else if (sampling_ratio == 2 && a == 7)                                                                           \
      MI_LAUNCH_SLAB(2, CAP7, 7, true);                                                                               \
    else if (sampling_ratio == 2 && a == 14 && levels == nullptr)                                                     \
      MI_LAUNCH_SLAB(2, CAP14, 14, false);                                                                            \
    else if (sampling_ratio == 2 && a == 14)                                                                          \
      MI_LAUNCH_SLAB(2, CAP14, 14, true);                                                                             \
    else if (sampling_ratio == 2)                                                                                     \
      MI_LAUNCH_SLAB(2, CAPG, 0, true);                                                                               \
    else                                                                                                              \
      MI_LAUNCH_SLAB(0, CAPG, 0, true);                                                                               \
  } while (0)
  // default: the largest image that keeps a wave's LDS inside 8 granules; MI_ROI_ALIGN_SLAB >= 64 picks other capacities
  // default: the largest image that keeps a wave's LDS inside 8 granules (292 pixels at 7x7 and 14x14: 18 waves per CU);
  // MI_ROI_ALIGN_SLAB >= 64 picks other capacities
  if (cap == 0)
    MI_SLAB_PICK((SlabLds<2, 7>::kCapMax), (SlabLds<2, 14>::kCapMax), (SlabLds<0, 0>::kCapMax));
  else if (cap >= 324)
    MI_SLAB_PICK(324, 324, 324);
  else if (cap >= 260)
    MI_SLAB_PICK(260, 260, 260);
  else
    MI_SLAB_PICK(228, 228, 228);
#undef MI_SLAB_PICK
#undef MI_LAUNCH_SLAB
  return check_launch("roi_align_fwd_slab");
}

int launch_roi_align_fwd_records(const float* features, const float* rois, float* output, void* workspace, int batch,
                                 int channels, int height, int width, int num_rois, int aligned_height,
                                 int aligned_width, float spatial_scale, int sampling_ratio, int cap_px, bool bwd_tables,
                                 hipStream_t stream) {
  return launch_roi_align_fwd_records_levels(single_level(features, nullptr, batch, height, width, spatial_scale), rois,
                                             nullptr, output, workspace, batch, channels, num_rois, aligned_height,
                                             aligned_width, sampling_ratio, cap_px, bwd_tables, stream, false);
}

}  // namespace mi
