// roi_align.hip -- RoIAlign forward / backward for gfx950 (MI355X), C-ABI mi_roi_align_*.
//
// Arithmetic contract (kept operation for operation, fp32, compiled with -ffp-contract=off so the
// results equal the CPU oracle bit for bit in the forward pass):
//   CAFFE2 variant: lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu:16-121 (fwd), :150-270 (bwd)
//   LEGACY variant: lib/model/roi_align/src/roi_align_kernel.cu:15-70 (fwd), :94-143 (bwd)
//
// Kernels in this file:
//   roi_align_fwd_direct / roi_align_bwd_direct   workgroup = (RoI, 32 channels), the RoI's geometry once per workgroup,
//       lanes over (channel, bin); any layout, reference operation order (bit-exact): the generic path -- shapes /
//       alignments the fast paths decline, MI_ROI_ALIGN_IMPL=direct.
//   roi_align_legacy<fwd / bwd>   the legacy variant from a per-workgroup point table (see there).
//   the dispatch of mi_roi_align_* to the fast paths, which live in files of their own: roi_align_records.hip (records,
//       the records-free and the record-driven NCHW forward, tile backward), roi_align_nhwc.hip (channels-last forward).
#include "common.h"
#include "roi_align_device.h"

#include <cstdlib>
#include <cstring>

namespace {

using namespace mi;

struct FeatStrides {
  long long n, c, h, w;  // element strides of the logical [N,C,H,W] tensor
};

__host__ FeatStrides make_strides(int layout, int C, int H, int W) {
  if (layout == MI_LAYOUT_NHWC) return {(long long)H * W * C, 1, (long long)W * C, C};
  return {(long long)C * H * W, (long long)H * W, W, 1};
}

// ------------------------------------------------------------------------------------------
// Generic direct kernels: workgroup = (RoI, tile of 32 channels).  What the reference recomputes for every output element
// (roi_align_kernel.cu:74-99: the RoI's box, bin sizes and sampling grid) is identical for the whole workgroup and comes
// out of scalar loads once; lanes run over (channel, bin) with the bin fastest, so a wave's taps lie in few rows of few
// planes and its results leave as one contiguous run (the tile's outputs are contiguous in [R][C][PH][PW]); over
// channels-last storage the channel is the fastest lane index instead (a tap = one line of 32 channels).  The
// arithmetic per element is the reference's, operation for operation.
// ------------------------------------------------------------------------------------------
constexpr int kDirCT = 32;
constexpr int kDirThreads = 256;

struct DirItem {
  int n, c0, cvalid, bins;
};
__device__ __forceinline__ DirItem dir_item(int channels, int aligned_height, int aligned_width) {
  const int tiles = (channels + kDirCT - 1) / kDirCT;
  DirItem it;
  it.n = (int)blockIdx.x / tiles;
  it.c0 = ((int)blockIdx.x - it.n * tiles) * kDirCT;
  it.cvalid = min(kDirCT, channels - it.c0);
  it.bins = aligned_height * aligned_width;
  return it;
}

// kSlabCT > 0 (planar storage): workgroup = (RoI, kSlabCT channels) on a grid (8 R, phases) -- an XCD reads one narrow channel slab
// of the map at a time, which fits its L2 whatever the order of the RoIs (roi_align_fwd_slab's mapping); 0: (RoI, 32 channels)
template <int kSlabCT>
__global__ void __launch_bounds__(kDirThreads)
roi_align_fwd_direct(const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out, int batch,
                     int channels, int height, int width, int aligned_height, int aligned_width, float spatial_scale,
                     int sampling_ratio, FeatStrides st) {
  DirItem it = dir_item(channels, aligned_height, aligned_width);
  if (kSlabCT > 0) {
    it.n = (int)(blockIdx.x >> 3);
    it.c0 = (int)(blockIdx.y * 8 + (blockIdx.x & 7)) * kSlabCT;
    if (it.c0 >= channels) return;
    it.cvalid = min(kSlabCT, channels - it.c0);
  }
  const RoiGeom g = roi_geometry(rois + (long long)it.n * 5, spatial_scale, aligned_height, aligned_width, sampling_ratio);
  float* __restrict__ dst = out + ((long long)it.n * channels + it.c0) * it.bins;
  const bool image_ok = g.batch_ind >= 0 && g.batch_ind < batch;  // the reference would read out of bounds
  const float* __restrict__ tile = feat + (image_ok ? g.batch_ind : 0) * st.n + it.c0 * st.c;
  const bool channel_fastest = st.c == 1;  // channels-last storage: a tap of 32 neighbouring lanes is one 128-byte line
  for (int i = threadIdx.x; i < it.cvalid * it.bins; i += kDirThreads) {
    const int c = channel_fastest ? i % it.cvalid : i / it.bins, b = channel_fastest ? i / it.cvalid : i - c * it.bins;
    const int ph = b / aligned_width, pw = b - ph * aligned_width, o = c * it.bins + b;
    if (!image_ok) {
      dst[o] = 0.f;
      continue;
    }
    const float* plane = tile + c * st.c;
    float output_val = 0.f;
    for (int iy = 0; iy < g.grid_h; iy++) {
      const float y = sample_y(g, ph, iy);
      for (int ix = 0; ix < g.grid_w; ix++) {
        const float x = sample_x(g, pw, ix);
        Taps t = sample_taps(height, width, y, x);
        float val = 0.f;
        if (t.y_low >= 0) {
          float v1 = plane[t.y_low * st.h + t.x_low * st.w];
          float v2 = plane[t.y_low * st.h + t.x_high * st.w];
          float v3 = plane[t.y_high * st.h + t.x_low * st.w];
          float v4 = plane[t.y_high * st.h + t.x_high * st.w];
          val = (t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4);  // :60
        }
        output_val += val;
      }
    }
    output_val /= g.count;  // :117
    dst[o] = output_val;
  }
}

__global__ void __launch_bounds__(kDirThreads)
roi_align_bwd_direct(const float* __restrict__ top_diff, const float* __restrict__ rois, float* __restrict__ bottom_diff,
                     int batch, int channels, int height, int width, int aligned_height, int aligned_width,
                     float spatial_scale, int sampling_ratio, FeatStrides st) {
  const DirItem it = dir_item(channels, aligned_height, aligned_width);
  const RoiGeom g = roi_geometry(rois + (long long)it.n * 5, spatial_scale, aligned_height, aligned_width, sampling_ratio);
  if (g.batch_ind < 0 || g.batch_ind >= batch) return;
  const float* __restrict__ src = top_diff + ((long long)it.n * channels + it.c0) * it.bins;
  float* __restrict__ tile = bottom_diff + g.batch_ind * st.n + it.c0 * st.c;
  const bool channel_fastest = st.c == 1;  // (over planar storage this order costs the atomics 2x: 3.4 against 1.6 ms at config 2)
  for (int i = threadIdx.x; i < it.cvalid * it.bins; i += kDirThreads) {
    const int c = channel_fastest ? i % it.cvalid : i / it.bins, b = channel_fastest ? i / it.cvalid : i - c * it.bins;
    const int ph = b / aligned_width, pw = b - ph * aligned_width;
    float* plane = tile + c * st.c;
    const float top_diff_this_bin = src[c * it.bins + b];
    for (int iy = 0; iy < g.grid_h; iy++) {
      const float y = sample_y(g, ph, iy);
      for (int ix = 0; ix < g.grid_w; ix++) {
        const float x = sample_x(g, pw, ix);
        Taps t = sample_taps(height, width, y, x);
        float g1 = top_diff_this_bin * t.w1 / g.count;  // :252-255
        float g2 = top_diff_this_bin * t.w2 / g.count;
        float g3 = top_diff_this_bin * t.w3 / g.count;
        float g4 = top_diff_this_bin * t.w4 / g.count;
        if (t.x_low >= 0 && t.x_high >= 0 && t.y_low >= 0 && t.y_high >= 0) {
          atomicAdd(plane + t.y_low * st.h + t.x_low * st.w, g1);
          atomicAdd(plane + t.y_low * st.h + t.x_high * st.w, g2);
          atomicAdd(plane + t.y_high * st.h + t.x_low * st.w, g3);
          atomicAdd(plane + t.y_high * st.h + t.x_high * st.w, g4);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Legacy variant (lib/model/roi_align/src/roi_align_kernel.cu:15-70, :94-143): ONE bilinear point per bin, at the corners
// of an (aligned - 1) grid over the box.  The reference mixes double literals into the expressions, so parts of the
// arithmetic are fp64; reproduced as written.  Same shape as roi_crop_fwd: workgroup = (RoI, 32 channels); one wave
// tabulates the points of the RoI 64 at a time (top-left pixel, the two ratios, inside or not -- :32-53, identical for
// every channel), then lanes run over (channel, point), the point fastest: neighbouring lanes tap neighbouring pixels of
// one plane and write neighbouring outputs.  The backward walks the same table and adds with global atomics as the
// reference does (its order is undefined there too).
// ------------------------------------------------------------------------------------------
constexpr int kLegPts = 64;
struct LegacyTab {
  int off[kLegPts];                        // hstart * width + wstart, or -1: the point lies outside the map (:53)
  float h_ratio[kLegPts], w_ratio[kLegPts];
};

// wave 0: points [p0, p0 + np) of the RoI
__device__ __forceinline__ void legacy_build_table(LegacyTab* tab, const float* __restrict__ roi, int p0, int np, int lane,
                                                   float spatial_scale, int height, int width, int aligned_height,
                                                   int aligned_width) {
  const int p = p0 + min(lane, np - 1), ph = p / aligned_width, pw = p - ph * aligned_width;
  float roi_start_w = roi[1] * spatial_scale;
  float roi_start_h = roi[2] * spatial_scale;
  float roi_end_w = roi[3] * spatial_scale;
  float roi_end_h = roi[4] * spatial_scale;
  float roi_width = fmaxf((float)((double)(roi_end_w - roi_start_w) + 1.), 0.f);  // :39-40
  float roi_height = fmaxf((float)((double)(roi_end_h - roi_start_h) + 1.), 0.f);
  float bin_size_h = (float)((double)roi_height / ((double)aligned_height - 1.));  // :41-42
  float bin_size_w = (float)((double)roi_width / ((double)aligned_width - 1.));
  float h = (float)(ph)*bin_size_h + roi_start_h;  // :44-45
  float w = (float)(pw)*bin_size_w + roi_start_w;
  const int hstart = (int)fminf(floorf(h), (float)(height - 2));  // :47-48
  const int wstart = (int)fminf(floorf(w), (float)(width - 2));
  const bool inside = !(h < 0 || h >= (float)height || w < 0 || w >= (float)width);  // :53
  tab->off[lane] = inside ? hstart * width + wstart : -1;
  tab->h_ratio[lane] = h - (float)hstart;
  tab->w_ratio[lane] = w - (float)wstart;
}

// The forward's workgroup = (RoI, 8 channels) on a grid (8 R, ceil(C / 64)): an XCD samples ONE 8-channel slab of the map at a
// time (it fits its L2 whatever the order of the RoIs: roi_align_records.hip, roi_align_fwd_slab); the backward keeps
// (RoI, 32 channels).  Config-2 shape: 29.6 -> 21.1 us (8 channels x 64 lanes; x 128: 22.4, x 256: 25.9;
// tools/build_defines.sh MI_LEG_CT / MI_LEG_THREADS).
#ifndef MI_LEG_CT
#define MI_LEG_CT 8
#endif
#ifndef MI_LEG_THREADS
#define MI_LEG_THREADS 64
#endif
constexpr int kLegFwdCT = MI_LEG_CT, kLegFwdThreads = MI_LEG_THREADS;
template <bool kBackward>
__global__ void __launch_bounds__(kBackward ? kDirThreads : kLegFwdThreads)
roi_align_legacy(const float* __restrict__ in, const float* __restrict__ rois, float* __restrict__ outp, int batch,
                 int channels, int height, int width, int aligned_height, int aligned_width, float spatial_scale) {
  __shared__ LegacyTab tab;
  constexpr int kThreadsHere = kBackward ? kDirThreads : kLegFwdThreads;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DirItem it = dir_item(channels, aligned_height, aligned_width);
  if (!kBackward && kLegFwdCT < 32) {
    it.n = (int)(blockIdx.x >> 3);
    it.c0 = (int)(blockIdx.y * 8 + (blockIdx.x & 7)) * kLegFwdCT;
    if (it.c0 >= channels) return;
    it.cvalid = min(kLegFwdCT, channels - it.c0);
  }
  const float* __restrict__ roi = rois + (long long)it.n * 5;
  const long long limit = (long long)batch * channels * height * width;
  // :50 `int img_start = roi_batch_ind * channels * height * width` is a float product
  const long long img = (long long)(int)(roi[0] * (float)channels * (float)height * (float)width);
  const long long plane_px = (long long)height * width;
  const long long tile_px = img + (long long)it.c0 * plane_px;                      // map side: the tile's first plane
  const long long tile_bins = ((long long)it.n * channels + it.c0) * it.bins;       // pooled side: the tile's first bin
  for (int p0 = 0; p0 < it.bins; p0 += kLegPts) {
    const int np = min(kLegPts, it.bins - p0);
    __syncthreads();  // the previous group's table is no longer read
    if (wave == 0) legacy_build_table(&tab, roi, p0, np, lane, spatial_scale, height, width, aligned_height, aligned_width);
    __syncthreads();
    const unsigned np_magic = (1u << 20) / (unsigned)np + 1u;
    for (int i = tid; i < it.cvalid * np; i += kThreadsHere) {
      const int c = (int)(((unsigned)i * np_magic) >> 20), p = i - c * np;  // i / np, exact for i < 32 * 64
      const long long bin = tile_bins + (long long)c * it.bins + p0 + p;
      const int off = tab.off[p];
      const long long upleft = tile_px + (long long)c * plane_px + off;
      const long long upright = upleft + 1, downleft = upleft + width, downright = downleft + 1;
      const bool ok = off >= 0 && upleft >= 0 && downright < limit;  // the second pair: a guard; the reference reads unchecked
      const float h_ratio = tab.h_ratio[p], w_ratio = tab.w_ratio[p];
      if (!kBackward) {
        float result = 0.f;
        if (ok) {
          // C++ promotion rules identical to the reference expression (:63-66): float*double terms are
          // evaluated in fp64, the float*float*float term in fp32, the sum in fp64.
          result = in[upleft] * (1. - h_ratio) * (1. - w_ratio) + in[upright] * (1. - h_ratio) * w_ratio +
                   in[downleft] * h_ratio * (1. - w_ratio) + in[downright] * h_ratio * w_ratio;
        }
        outp[bin] = result;
      } else if (ok) {
        // same literal types as the reference (:135-138): `1.` is double, `1` is int
        const float g = in[bin];
        atomicAdd(outp + upleft, (float)(g * (1. - h_ratio) * (1 - w_ratio)));
        atomicAdd(outp + upright, (float)(g * (1. - h_ratio) * w_ratio));
        atomicAdd(outp + downleft, (float)(g * h_ratio * (1 - w_ratio)));
        atomicAdd(outp + downright, (float)(g * h_ratio * w_ratio));
      }
    }
  }
}

bool force_direct() { return mi::tuning().force_direct; }
bool no_ws() { return mi::tuning().no_ws; }
int ring_words() { return mi::tuning().cap_px; }

int check_common(const void* a, const void* rois, const void* b, int batch, int channels,
                 int height, int width, int num_rois, int ah, int aw, int variant, int layout) {
  MI_REQUIRE(batch >= 0 && channels >= 0 && height >= 0 && width >= 0 && num_rois >= 0,
             "roi_align: negative size");
  MI_REQUIRE(ah > 0 && aw > 0, "roi_align: aligned size must be positive");
  MI_REQUIRE(variant == MI_ROI_ALIGN_CAFFE2 || variant == MI_ROI_ALIGN_LEGACY,
             "roi_align: unknown variant %d", variant);
  MI_REQUIRE(layout == MI_LAYOUT_NCHW || layout == MI_LAYOUT_NHWC, "roi_align: unknown layout %d",
             layout);
  MI_REQUIRE(!(variant == MI_ROI_ALIGN_LEGACY && layout != MI_LAYOUT_NCHW),
             "roi_align: the legacy variant is NCHW only");
  long long out_elems = (long long)num_rois * channels * ah * aw;
  long long in_elems = (long long)batch * channels * height * width;
  if (out_elems > 0)
    MI_REQUIRE(a != nullptr && rois != nullptr && b != nullptr && in_elems > 0,
               "roi_align: null pointer or empty feature map");
  return MI_OK;
}

}  // namespace

// Tuning aid, not part of include/mi_detectron_ops.h: device buffer of 8 int64 stamps per forward workgroup
// (tools/timeline.py); nullptr switches the stamps off.
extern "C" void mi_dbg_roi_align_timeline(long long* device_buffer) {
  mi::roi_align_fwd_nhwc_set_timeline(device_buffer);
  mi::roi_align_fwd_records_set_timeline(device_buffer);
}

namespace {
int roi_align_forward_impl(const float* features, const float* rois, float* output, int batch, int channels,
                           int height, int width, int num_rois, int aligned_height, int aligned_width,
                           float spatial_scale, int sampling_ratio, int variant, int layout, void* workspace,
                           size_t workspace_bytes, mi_stream_t stream) {
  mi::begin_call();
  int rc = check_common(features, rois, output, batch, channels, height, width, num_rois,
                        aligned_height, aligned_width, variant, layout);
  if (rc != MI_OK) return rc;
  const long long total = (long long)num_rois * channels * aligned_height * aligned_width;
  if (total == 0) return MI_OK;
  hipStream_t s = mi::as_stream(stream);
  // the generic kernels: one workgroup per (RoI, 32 channels), whose pooled bins are walked 256 at a time
  const long long dir_grid = (long long)num_rois * mi::ceil_div(channels, kDirCT);
  MI_REQUIRE(dir_grid < (1LL << 31) && (long long)kDirCT * aligned_height * aligned_width < (1LL << 31), "roi_align: too many (RoI, channel tile) items");
  if (variant == MI_ROI_ALIGN_LEGACY) {
    const dim3 lgrid = kLegFwdCT < 32 ? dim3((unsigned)num_rois * 8u, (unsigned)((mi::ceil_div(channels, kLegFwdCT) + 7) / 8))
                                      : dim3((unsigned)dir_grid);
    roi_align_legacy<false><<<lgrid, kLegFwdThreads, 0, s>>>(features, rois, output, batch, channels, height, width,
                                                                  aligned_height, aligned_width, spatial_scale);
    return mi::check_launch("roi_align_legacy<fwd>");
  }
  const int cap = ring_words();
  if (workspace != nullptr) {
    MI_REQUIRE(workspace_bytes >= mi::roi_align_records_workspace_bytes(num_rois),
               "roi_align: workspace of %zu bytes, %zu needed", workspace_bytes,
               mi::roi_align_records_workspace_bytes(num_rois));
    MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "roi_align: workspace must be 16-byte aligned");
    // a workspace with room for the (planned) backward announces one: the records then carry the backward block
    const bool bwd_tables =
        workspace_bytes >= mi::roi_align_bwd_workspace_bytes(mi::single_level(nullptr, nullptr, batch, height, width,
                                                                              spatial_scale), batch, num_rois);
    // nobody will read records (no room for a backward): the records-free forward, one launch
    if (layout == MI_LAYOUT_NCHW && !force_direct() && !no_ws() && !bwd_tables &&
        mi::roi_align_fwd_slab_supported(mi::single_level(features, nullptr, batch, height, width, spatial_scale), channels,
                                         num_rois, aligned_height, aligned_width))
      return mi::launch_roi_align_fwd_slab(mi::single_level(features, nullptr, batch, height, width, spatial_scale), rois,
                                           nullptr, output, batch, channels, num_rois, aligned_height, aligned_width,
                                           sampling_ratio, s);
    if (layout == MI_LAYOUT_NCHW && !force_direct() && !no_ws() &&
        mi::roi_align_fwd_records_supported(channels, height, width, num_rois, aligned_height, aligned_width))
      return mi::launch_roi_align_fwd_records(features, rois, output, workspace, batch, channels, height, width,
                                              num_rois, aligned_height, aligned_width, spatial_scale,
                                              sampling_ratio, cap, bwd_tables, s);
    if (layout == MI_LAYOUT_NHWC && !force_direct() && !no_ws() &&
        num_rois <= 8192 &&
        mi::roi_align_fwd_nhwc_supported(channels, height, width, num_rois, aligned_height, aligned_width)) {
      rc = mi::launch_roi_align_prepare(rois, workspace, batch, height, width, num_rois, aligned_height,
                                        aligned_width, spatial_scale, sampling_ratio, bwd_tables, s, channels, features);
      if (rc != MI_OK) return rc;
      return mi::launch_roi_align_fwd_nhwc(features, rois, output, workspace, batch, channels, height, width,
                                           num_rois, aligned_height, aligned_width, spatial_scale, sampling_ratio, s);
    }
  }
  if (layout == MI_LAYOUT_NCHW && !force_direct() &&
      mi::roi_align_fwd_slab_supported(mi::single_level(features, nullptr, batch, height, width, spatial_scale), channels,
                                       num_rois, aligned_height, aligned_width))
    return mi::launch_roi_align_fwd_slab(mi::single_level(features, nullptr, batch, height, width, spatial_scale), rois,
                                         nullptr, output, batch, channels, num_rois, aligned_height, aligned_width,
                                         sampling_ratio, s);
  FeatStrides st = make_strides(layout, channels, height, width);
  if (layout == MI_LAYOUT_NCHW) {
    constexpr int kCT = 8;
    const dim3 sgrid((unsigned)num_rois * 8u, (unsigned)((mi::ceil_div(channels, kCT) + 7) / 8));
    roi_align_fwd_direct<kCT><<<sgrid, kDirThreads, 0, s>>>(features, rois, output, batch, channels, height, width,
                                                            aligned_height, aligned_width, spatial_scale, sampling_ratio, st);
  } else {
    roi_align_fwd_direct<0><<<(int)dir_grid, kDirThreads, 0, s>>>(features, rois, output, batch, channels, height, width,
                                                                  aligned_height, aligned_width, spatial_scale, sampling_ratio, st);
  }
  return mi::check_launch("roi_align_fwd_direct");
}
}  // namespace

extern "C" int mi_roi_align_forward(const float* features, const float* rois, float* output,
                                    int batch, int channels, int height, int width, int num_rois,
                                    int aligned_height, int aligned_width, float spatial_scale,
                                    int sampling_ratio, int variant, int layout,
                                    mi_stream_t stream) {
  return roi_align_forward_impl(features, rois, output, batch, channels, height, width, num_rois, aligned_height,
                                aligned_width, spatial_scale, sampling_ratio, variant, layout, nullptr, 0, stream);
}

extern "C" size_t mi_roi_align_forward_workspace_bytes(int num_rois) {
  return mi::roi_align_records_workspace_bytes(num_rois);
}

extern "C" int mi_roi_align_forward_ws(const float* features, const float* rois, float* output,
                                       int batch, int channels, int height, int width, int num_rois,
                                       int aligned_height, int aligned_width, float spatial_scale,
                                       int sampling_ratio, int variant, int layout,
                                       void* workspace, size_t workspace_bytes, mi_stream_t stream) {
  return roi_align_forward_impl(features, rois, output, batch, channels, height, width, num_rois, aligned_height,
                                aligned_width, spatial_scale, sampling_ratio, variant, layout, workspace,
                                workspace_bytes, stream);
}

namespace {
int roi_align_backward_impl(const float* top_grad, const float* rois, float* bottom_grad, int batch, int channels,
                            int height, int width, int num_rois, int aligned_height, int aligned_width,
                            float spatial_scale, int sampling_ratio, int variant, int layout, void* workspace,
                            size_t workspace_bytes, int records_ready /* bit0: records ready, bit1: overwrite */, mi_stream_t stream) {
  mi::begin_call();
  int rc = check_common(top_grad, rois, bottom_grad, batch, channels, height, width, num_rois,
                        aligned_height, aligned_width, variant, layout);
  if (rc != MI_OK) return rc;
  const long long total = (long long)num_rois * channels * aligned_height * aligned_width;
  hipStream_t s = mi::as_stream(stream);
  if (total == 0) {
    // no RoI contributes: with the OVERWRITE contract the caller did not zero-fill, so the zeros are ours to write
    const long long in_elems = (long long)batch * channels * height * width;
    if ((records_ready & 2) != 0 && in_elems > 0 && bottom_grad != nullptr &&
        hipMemsetAsync(bottom_grad, 0, (size_t)in_elems * sizeof(float), s) != hipSuccess)
      return mi::check_launch("roi_align_backward: zero fill");
    return MI_OK;
  }
  const long long dir_grid = (long long)num_rois * mi::ceil_div(channels, kDirCT);
  MI_REQUIRE(dir_grid < (1LL << 31) && (long long)kDirCT * aligned_height * aligned_width < (1LL << 31), "roi_align: too many (RoI, channel tile) items");
  if (variant == MI_ROI_ALIGN_LEGACY) {
    roi_align_legacy<true><<<(int)dir_grid, kDirThreads, 0, s>>>(top_grad, rois, bottom_grad, batch, channels, height, width,
                                                                 aligned_height, aligned_width, spatial_scale);
    return mi::check_launch("roi_align_legacy<bwd>");
  }
  if (workspace != nullptr) {
    MI_REQUIRE(workspace_bytes >= mi::roi_align_records_workspace_bytes(num_rois),
               "roi_align: workspace of %zu bytes, %zu needed", workspace_bytes,
               mi::roi_align_records_workspace_bytes(num_rois));
    MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "roi_align: workspace must be 16-byte aligned");
    // the tile kernel takes a RoI's block of top gradients in 16-byte pieces: a dword-aligned gradient view (legal at this
    // boundary) goes to the generic kernel below instead of being refused
    if (!force_direct() && !no_ws() && (reinterpret_cast<uintptr_t>(top_grad) & 15) == 0 &&
        mi::roi_align_bwd_records_supported(channels, height, width, num_rois, aligned_height, aligned_width))
      return mi::launch_roi_align_bwd_records(top_grad, rois, bottom_grad, workspace, workspace_bytes,
                                              (records_ready & 1) != 0,
                                              (records_ready & 2) != 0, layout == MI_LAYOUT_NHWC, batch, channels,
                                              height, width, num_rois, aligned_height, aligned_width, spatial_scale,
                                              sampling_ratio, ring_words(), s);
  }
  // the generic kernel accumulates: under the OVERWRITE contract (only reachable here with a workspace and a gradient
  // view the tile kernel cannot take) the zeros are ours to write
  if ((records_ready & 2) != 0 &&
      hipMemsetAsync(bottom_grad, 0, (size_t)batch * channels * height * width * sizeof(float), s) != hipSuccess)
    return mi::check_launch("roi_align_backward: zero fill");
  FeatStrides st = make_strides(layout, channels, height, width);
  roi_align_bwd_direct<<<(int)dir_grid, kDirThreads, 0, s>>>(top_grad, rois, bottom_grad, batch, channels, height, width,
                                                            aligned_height, aligned_width, spatial_scale, sampling_ratio, st);
  return mi::check_launch("roi_align_bwd_direct");
}
}  // namespace

extern "C" int mi_roi_align_backward(const float* top_grad, const float* rois, float* bottom_grad,
                                     int batch, int channels, int height, int width, int num_rois,
                                     int aligned_height, int aligned_width, float spatial_scale,
                                     int sampling_ratio, int variant, int layout,
                                     mi_stream_t stream) {
  return roi_align_backward_impl(top_grad, rois, bottom_grad, batch, channels, height, width, num_rois,
                                 aligned_height, aligned_width, spatial_scale, sampling_ratio, variant, layout, nullptr,
                                 0, 0, stream);
}

extern "C" int mi_roi_align_backward_ws(const float* top_grad, const float* rois, float* bottom_grad,
                                        int batch, int channels, int height, int width, int num_rois,
                                        int aligned_height, int aligned_width, float spatial_scale,
                                        int sampling_ratio, int variant, int layout, void* workspace,
                                        size_t workspace_bytes, int flags, mi_stream_t stream) {
  return roi_align_backward_impl(top_grad, rois, bottom_grad, batch, channels, height, width, num_rois,
                                 aligned_height, aligned_width, spatial_scale, sampling_ratio, variant, layout,
                                 workspace, workspace_bytes, flags, stream);
}

namespace {
// mi_fpn_levels -> LevelTable; returns false when the table is malformed
bool to_level_table(const mi_fpn_levels* in, int batch, bool forward, mi::LevelTable* out) {
  if (in == nullptr || in->num_levels < 1 || in->num_levels > mi::kMaxLevels) return false;
  mi::LevelTable t = {};
  t.count = in->num_levels;
  for (int l = 0; l < t.count; l++) {
    if (in->height[l] <= 0 || in->width[l] <= 0) return false;
    if (forward ? in->features[l] == nullptr : in->grads[l] == nullptr) return false;
    t.height[l] = in->height[l];
    t.width[l] = in->width[l];
    t.scale[l] = in->spatial_scale[l];
    t.feat[l] = in->features[l];
    t.grad[l] = in->grads[l];
    t.row_base[l + 1] = t.row_base[l] + batch * in->height[l];
  }
  *out = t;
  return true;
}
}  // namespace

extern "C" size_t mi_roi_align_backward_workspace_bytes(const mi_fpn_levels* levels, int batch, int num_rois) {
  if (levels == nullptr || levels->num_levels < 1 || levels->num_levels > mi::kMaxLevels || num_rois <= 0 || batch <= 0)
    return mi::roi_align_records_workspace_bytes(num_rois);
  mi::LevelTable lv = {};
  lv.count = levels->num_levels;
  for (int l = 0; l < lv.count; l++) {
    if (levels->height[l] <= 0 || levels->width[l] <= 0) return mi::roi_align_records_workspace_bytes(num_rois);
    lv.height[l] = levels->height[l];
    lv.width[l] = levels->width[l];
  }
  return mi::roi_align_bwd_workspace_bytes(lv, batch, num_rois);
}

extern "C" int mi_roi_align_fpn_supported(const mi_fpn_levels* levels, int channels, int num_rois, int aligned_height,
                                          int aligned_width, int layout) {
  if (levels == nullptr || levels->num_levels < 1 || levels->num_levels > mi::kMaxLevels || force_direct() ||
      no_ws() || num_rois <= 0 || num_rois > 8192 ||
      (layout != MI_LAYOUT_NCHW && layout != MI_LAYOUT_NHWC))
    return 0;
  for (int l = 0; l < levels->num_levels; l++) {
    const int h = levels->height[l], w = levels->width[l];
    const bool fwd = layout == MI_LAYOUT_NCHW
                         ? mi::roi_align_fwd_records_supported(channels, h, w, num_rois, aligned_height, aligned_width)
                         : mi::roi_align_fwd_nhwc_supported(channels, h, w, num_rois, aligned_height, aligned_width);
    if (!fwd || !mi::roi_align_bwd_records_supported(channels, h, w, num_rois, aligned_height, aligned_width)) return 0;
  }
  return 1;
}

namespace {
int roi_align_forward_fpn_impl(const mi_fpn_levels* levels, const float* rois, const int32_t* roi_levels, float* output,
                               int batch, int channels, int num_rois, int aligned_height, int aligned_width,
                               int sampling_ratio, int layout, void* workspace, size_t workspace_bytes, bool records_ready,
                               mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(batch > 0 && channels > 0 && num_rois >= 0 && aligned_height > 0 && aligned_width > 0,
             "roi_align_fpn: bad size");
  if (num_rois == 0) return MI_OK;
  mi::LevelTable lv;
  MI_REQUIRE(to_level_table(levels, batch, true, &lv), "roi_align_fpn: malformed level table");
  MI_REQUIRE(rois != nullptr && roi_levels != nullptr && output != nullptr, "roi_align_fpn: null pointer");
  MI_REQUIRE(mi_roi_align_fpn_supported(levels, channels, num_rois, aligned_height, aligned_width, layout) == 1,
             "roi_align_fpn: shapes not served by the fused path (mi_roi_align_fpn_supported() == 0)");
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "roi_align_fpn: workspace must be 16-byte aligned");
  const int cap = ring_words();
  MI_REQUIRE(workspace != nullptr, "roi_align_fpn: null pointer");
  MI_REQUIRE(workspace_bytes >= mi::roi_align_records_workspace_bytes(num_rois),
             "roi_align_fpn: workspace of %zu bytes, %zu needed", workspace_bytes,
             mi::roi_align_records_workspace_bytes(num_rois));
  const bool bwd_tables = workspace_bytes >= mi::roi_align_bwd_workspace_bytes(lv, batch, num_rois);
  if (layout == MI_LAYOUT_NHWC) {
    if (!records_ready) {
      int rc = mi::launch_roi_align_prepare_levels(lv, rois, roi_levels, workspace, batch, num_rois, aligned_height,
                                                   aligned_width, sampling_ratio, bwd_tables, mi::as_stream(stream), channels);
      if (rc != MI_OK) return rc;
    }
    return mi::launch_roi_align_fwd_nhwc_levels(lv, rois, output, workspace, batch, channels, num_rois, aligned_height,
                                                aligned_width, sampling_ratio, mi::as_stream(stream));
  }
  // nobody will read records (forward-sized workspace, none written by a producer): the records-free forward, one launch
  if (!bwd_tables && !records_ready && mi::roi_align_fwd_slab_supported(lv, channels, num_rois, aligned_height, aligned_width))
    return mi::launch_roi_align_fwd_slab(lv, rois, roi_levels, output, batch, channels, num_rois, aligned_height,
                                         aligned_width, sampling_ratio, mi::as_stream(stream));
  return mi::launch_roi_align_fwd_records_levels(lv, rois, roi_levels, output, workspace, batch, channels, num_rois,
                                                 aligned_height, aligned_width, sampling_ratio, cap, bwd_tables,
                                                 mi::as_stream(stream), records_ready);
}
}  // namespace

extern "C" int mi_roi_align_forward_fpn(const mi_fpn_levels* levels, const float* rois, const int32_t* roi_levels,
                                        float* output, int batch, int channels, int num_rois, int aligned_height,
                                        int aligned_width, int sampling_ratio, int layout, void* workspace,
                                        size_t workspace_bytes, mi_stream_t stream) {
  return roi_align_forward_fpn_impl(levels, rois, roi_levels, output, batch, channels, num_rois, aligned_height,
                                    aligned_width, sampling_ratio, layout, workspace, workspace_bytes, false, stream);
}

extern "C" int mi_roi_align_forward_fpn_records(const mi_fpn_levels* levels, const float* rois, const int32_t* roi_levels,
                                                float* output, int batch, int channels, int num_rois, int aligned_height,
                                                int aligned_width, int sampling_ratio, int layout, void* workspace,
                                                size_t workspace_bytes, mi_stream_t stream) {
  return roi_align_forward_fpn_impl(levels, rois, roi_levels, output, batch, channels, num_rois, aligned_height,
                                    aligned_width, sampling_ratio, layout, workspace, workspace_bytes, true, stream);
}

extern "C" int mi_rpn_collect_finish_records(const float* top_scores, const int64_t* top_indices, const float* cand_rois,
                                             int rows, int mark_invalid, int k_min, int k_max, float canonical_scale,
                                             float canonical_level, float* rois, uint8_t* valid, int32_t* roi_fpn_levels,
                                             const mi_fpn_levels* levels, int batch, int channels, int aligned_height,
                                             int aligned_width, int sampling_ratio, int layout, void* workspace,
                                             size_t workspace_bytes, mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(rows >= 0 && k_min <= k_max, "rpn_collect_finish_records: bad size");
  if (rows == 0) return MI_OK;
  MI_REQUIRE(top_scores != nullptr && top_indices != nullptr && cand_rois != nullptr && rois != nullptr &&
                 valid != nullptr && roi_fpn_levels != nullptr && workspace != nullptr,
             "rpn_collect_finish_records: null pointer");
  MI_REQUIRE(batch > 0 && channels > 0 && aligned_height > 0 && aligned_width > 0, "rpn_collect_finish_records: bad size");
  mi::LevelTable lv;
  MI_REQUIRE(to_level_table(levels, batch, true, &lv), "rpn_collect_finish_records: malformed level table");
  MI_REQUIRE(lv.count == k_max - k_min + 1,
             "rpn_collect_finish_records: %d maps for FPN levels %d..%d (coarsest first)", lv.count, k_min, k_max);
  MI_REQUIRE(mi_roi_align_fpn_supported(levels, channels, rows, aligned_height, aligned_width, layout) == 1,
             "rpn_collect_finish_records: shapes not served by the fused RoIAlign (mi_roi_align_fpn_supported() == 0)");
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0 &&
                 workspace_bytes >= mi::roi_align_records_workspace_bytes(rows),
             "rpn_collect_finish_records: workspace of %zu bytes (16-byte aligned), %zu needed", workspace_bytes,
             mi::roi_align_records_workspace_bytes(rows));
  const bool bwd_tables = workspace_bytes >= mi::roi_align_bwd_workspace_bytes(lv, batch, rows);
  return mi::launch_roi_align_prepare_collected(lv, top_scores, reinterpret_cast<const long long*>(top_indices), cand_rois,
                                                mark_invalid, k_min, k_max, canonical_scale, canonical_level, rois, valid,
                                                roi_fpn_levels, workspace, batch, rows, aligned_height, aligned_width,
                                                sampling_ratio, ring_words(), bwd_tables, mi::as_stream(stream), channels);
}

extern "C" int mi_roi_align_backward_fpn(const mi_fpn_levels* levels, const float* top_grad, const float* rois,
                                         const int32_t* roi_levels, int batch, int channels, int num_rois,
                                         int aligned_height, int aligned_width, int sampling_ratio, int layout,
                                         void* workspace, size_t workspace_bytes, int flags, mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(batch > 0 && channels > 0 && num_rois >= 0 && aligned_height > 0 && aligned_width > 0,
             "roi_align_fpn: bad size");
  mi::LevelTable lv;
  MI_REQUIRE(to_level_table(levels, batch, false, &lv), "roi_align_fpn: malformed level table");
  if (num_rois == 0) {
    // an empty RoI set (e.g. no foreground RoI for the mask head): every level's gradient is all zeros
    if ((flags & 2) != 0)
      for (int l = 0; l < lv.count; l++)
        if (hipMemsetAsync(lv.grad[l], 0, (size_t)batch * channels * lv.height[l] * lv.width[l] * sizeof(float),
                           mi::as_stream(stream)) != hipSuccess)
          return mi::check_launch("roi_align_backward_fpn: zero fill");
    return MI_OK;
  }
  MI_REQUIRE(rois != nullptr && roi_levels != nullptr && top_grad != nullptr && workspace != nullptr,
             "roi_align_fpn: null pointer");
  MI_REQUIRE(mi_roi_align_fpn_supported(levels, channels, num_rois, aligned_height, aligned_width, layout) == 1,
             "roi_align_fpn: shapes not served by the fused path (mi_roi_align_fpn_supported() == 0)");
  MI_REQUIRE(workspace_bytes >= mi::roi_align_records_workspace_bytes(num_rois),
             "roi_align_fpn: workspace of %zu bytes, %zu needed", workspace_bytes,
             mi::roi_align_records_workspace_bytes(num_rois));
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "roi_align_fpn: workspace must be 16-byte aligned");
  MI_REQUIRE((reinterpret_cast<uintptr_t>(top_grad) & 15) == 0, "roi_align_fpn: top_grad must be 16-byte aligned");
  return mi::launch_roi_align_bwd_records_levels(top_grad, rois, roi_levels, lv, workspace, workspace_bytes,
                                                 (flags & 1) != 0,
                                                 (flags & 2) != 0, layout == MI_LAYOUT_NHWC, batch, channels, num_rois,
                                                 aligned_height,
                                                 aligned_width, sampling_ratio, ring_words(), mi::as_stream(stream));
}

extern "C" int mi_roi_align_forward_writes_records(int channels, int height, int width, int num_rois,
                                                   int aligned_height, int aligned_width, int variant, int layout) {
  if (variant != MI_ROI_ALIGN_CAFFE2 || force_direct() || no_ws() || num_rois <= 0)
    return 0;
  if (layout == MI_LAYOUT_NCHW)
    return mi::roi_align_fwd_records_supported(channels, height, width, num_rois, aligned_height, aligned_width) ? 1 : 0;
  if (layout == MI_LAYOUT_NHWC)
    return num_rois <= 8192 &&
                   mi::roi_align_fwd_nhwc_supported(channels, height, width, num_rois, aligned_height, aligned_width)
               ? 1
               : 0;
  return 0;
}

extern "C" int mi_roi_align_backward_overwrites(int channels, int height, int width, int num_rois, int aligned_height,
                                                int aligned_width, int variant, int layout) {
  return variant == MI_ROI_ALIGN_CAFFE2 && (layout == MI_LAYOUT_NCHW || layout == MI_LAYOUT_NHWC) && !force_direct() &&
                 !no_ws() &&
                 mi::roi_align_bwd_records_supported(channels, height, width, num_rois, aligned_height, aligned_width)
             ? 1
             : 0;
}
