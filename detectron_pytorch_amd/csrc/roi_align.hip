// roi_align.hip -- RoIAlign forward / backward for gfx950 (MI355X), C-ABI mi_roi_align_*.
//
// Arithmetic contract (kept operation for operation, fp32, compiled with -ffp-contract=off so the
// results equal the CPU oracle bit for bit in the forward pass):
//   CAFFE2 variant: lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu:16-121 (fwd), :150-270 (bwd)
//   LEGACY variant: lib/model/roi_align/src/roi_align_kernel.cu:15-70 (fwd), :94-143 (bwd)
//
// Kernels in this file:
//   roi_align_fwd_direct / roi_align_bwd_direct   one lane per output element, any layout, reference operation order
//       (bit-exact): the generic path -- legacy variant, shapes / alignments the fast paths decline, MI_ROI_ALIGN_IMPL=direct.
//   the dispatch of mi_roi_align_* to the fast paths, which live in files of their own: roi_align_records.hip (records,
//       NCHW forward, tile backward), roi_align_nhwc.hip (channels-last forward), roi_align_fwd_tile.hip (NCHW forward
//       without a workspace).
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
// Generic direct kernels (one lane per output element; reference thread mapping)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
roi_align_fwd_direct(long long total, const float* __restrict__ feat, const float* __restrict__ rois,
                     float* __restrict__ out, int batch, int channels, int height, int width,
                     int aligned_height, int aligned_width, float spatial_scale, int sampling_ratio,
                     FeatStrides st) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
       index += (long long)gridDim.x * blockDim.x) {
    int pw = (int)(index % aligned_width);
    int ph = (int)((index / aligned_width) % aligned_height);
    int c = (int)((index / aligned_width / aligned_height) % channels);
    int n = (int)(index / aligned_width / aligned_height / channels);
    RoiGeom g = roi_geometry(rois + (long long)n * 5, spatial_scale, aligned_height, aligned_width,
                             sampling_ratio);
    if (g.batch_ind < 0 || g.batch_ind >= batch) {  // the reference would read out of bounds
      out[index] = 0.f;
      continue;
    }
    const float* plane = feat + g.batch_ind * st.n + c * st.c;
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
    out[index] = output_val;
  }
}

__global__ void __launch_bounds__(256)
roi_align_bwd_direct(long long total, const float* __restrict__ top_diff,
                     const float* __restrict__ rois, float* __restrict__ bottom_diff, int batch,
                     int channels, int height, int width, int aligned_height, int aligned_width,
                     float spatial_scale, int sampling_ratio, FeatStrides st) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
       index += (long long)gridDim.x * blockDim.x) {
    int pw = (int)(index % aligned_width);
    int ph = (int)((index / aligned_width) % aligned_height);
    int c = (int)((index / aligned_width / aligned_height) % channels);
    int n = (int)(index / aligned_width / aligned_height / channels);
    RoiGeom g = roi_geometry(rois + (long long)n * 5, spatial_scale, aligned_height, aligned_width,
                             sampling_ratio);
    if (g.batch_ind < 0 || g.batch_ind >= batch) continue;
    float* plane = bottom_diff + g.batch_ind * st.n + c * st.c;
    const float top_diff_this_bin = top_diff[index];
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
// Legacy variant (lib/model/roi_align/src/roi_align_kernel.cu).  The reference mixes double
// literals into the expressions, so parts of the arithmetic are fp64; reproduced as written.
// ------------------------------------------------------------------------------------------
struct LegacyPoint {
  bool inside;
  int hstart, wstart;
  float h_ratio, w_ratio;
  int img;
};

__device__ __forceinline__ LegacyPoint legacy_point(const float* __restrict__ roi, int ph, int pw,
                                                    float spatial_scale, int channels, int height,
                                                    int width, int aligned_height,
                                                    int aligned_width) {
  LegacyPoint p;
  float roi_batch_ind = roi[0];
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
  p.hstart = (int)fminf(floorf(h), (float)(height - 2));  // :47-48
  p.wstart = (int)fminf(floorf(w), (float)(width - 2));
  // :50 `int img_start = roi_batch_ind * channels * height * width` is a float product
  p.img = (int)(roi_batch_ind * (float)channels * (float)height * (float)width);
  p.inside = !(h < 0 || h >= (float)height || w < 0 || w >= (float)width);  // :53
  p.h_ratio = h - (float)p.hstart;
  p.w_ratio = w - (float)p.wstart;
  return p;
}

__global__ void __launch_bounds__(256)
roi_align_legacy_fwd(long long total, const float* __restrict__ bottom_data,
                     const float* __restrict__ rois, float* __restrict__ top_data, int batch,
                     int channels, int height, int width, int aligned_height, int aligned_width,
                     float spatial_scale) {
  const long long limit = (long long)batch * channels * height * width;
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
       index += (long long)gridDim.x * blockDim.x) {
    int pw = (int)(index % aligned_width);
    int ph = (int)((index / aligned_width) % aligned_height);
    int c = (int)((index / aligned_width / aligned_height) % channels);
    int n = (int)(index / aligned_width / aligned_height / channels);
    LegacyPoint p = legacy_point(rois + (long long)n * 5, ph, pw, spatial_scale, channels, height,
                                 width, aligned_height, aligned_width);
    float result = 0.f;
    if (p.inside) {
      long long upleft = (long long)p.img + ((long long)c * height + p.hstart) * width + p.wstart;
      long long upright = upleft + 1, downleft = upleft + width, downright = downleft + 1;
      if (upleft >= 0 && downright < limit) {  // guard; the reference reads unchecked
        // C++ promotion rules identical to the reference expression (:63-66): float*double terms are
        // evaluated in fp64, the float*float*float term in fp32, the sum in fp64.
        const float h_ratio = p.h_ratio, w_ratio = p.w_ratio;
        result = bottom_data[upleft] * (1. - h_ratio) * (1. - w_ratio) +
                 bottom_data[upright] * (1. - h_ratio) * w_ratio +
                 bottom_data[downleft] * h_ratio * (1. - w_ratio) +
                 bottom_data[downright] * h_ratio * w_ratio;
      }
    }
    top_data[index] = result;
  }
}

__global__ void __launch_bounds__(256)
roi_align_legacy_bwd(long long total, const float* __restrict__ top_diff,
                     const float* __restrict__ rois, float* __restrict__ bottom_diff, int batch,
                     int channels, int height, int width, int aligned_height, int aligned_width,
                     float spatial_scale) {
  const long long limit = (long long)batch * channels * height * width;
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
       index += (long long)gridDim.x * blockDim.x) {
    int pw = (int)(index % aligned_width);
    int ph = (int)((index / aligned_width) % aligned_height);
    int c = (int)((index / aligned_width / aligned_height) % channels);
    int n = (int)(index / aligned_width / aligned_height / channels);
    LegacyPoint p = legacy_point(rois + (long long)n * 5, ph, pw, spatial_scale, channels, height,
                                 width, aligned_height, aligned_width);
    if (!p.inside) continue;
    long long upleft = (long long)p.img + ((long long)c * height + p.hstart) * width + p.wstart;
    long long upright = upleft + 1, downleft = upleft + width, downright = downleft + 1;
    if (upleft < 0 || downright >= limit) continue;
    // same literal types as the reference (:135-138): `1.` is double, `1` is int
    const float h_ratio = p.h_ratio, w_ratio = p.w_ratio, g = top_diff[index];
    atomicAdd(bottom_diff + upleft, (float)(g * (1. - h_ratio) * (1 - w_ratio)));
    atomicAdd(bottom_diff + upright, (float)(g * (1. - h_ratio) * w_ratio));
    atomicAdd(bottom_diff + downleft, (float)(g * h_ratio * (1 - w_ratio)));
    atomicAdd(bottom_diff + downright, (float)(g * h_ratio * w_ratio));
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
  mi::roi_align_fwd_tile_set_timeline(device_buffer);
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
  const int block = 256;
  if (variant == MI_ROI_ALIGN_LEGACY) {
    roi_align_legacy_fwd<<<mi::grid_for(total, block), block, 0, s>>>(
        total, features, rois, output, batch, channels, height, width, aligned_height,
        aligned_width, spatial_scale);
    return mi::check_launch("roi_align_legacy_fwd");
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
      mi::roi_align_fwd_tile_supported(channels, height, width, aligned_height, aligned_width))
    return mi::launch_roi_align_fwd_tile(features, rois, output, batch, channels, height, width, num_rois,
                                         aligned_height, aligned_width, spatial_scale, sampling_ratio, cap, s);
  FeatStrides st = make_strides(layout, channels, height, width);
  roi_align_fwd_direct<<<mi::grid_for(total, block), block, 0, s>>>(
      total, features, rois, output, batch, channels, height, width, aligned_height, aligned_width,
      spatial_scale, sampling_ratio, st);
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
  const int block = 256;
  if (variant == MI_ROI_ALIGN_LEGACY) {
    roi_align_legacy_bwd<<<mi::grid_for(total, block), block, 0, s>>>(
        total, top_grad, rois, bottom_grad, batch, channels, height, width, aligned_height,
        aligned_width, spatial_scale);
    return mi::check_launch("roi_align_legacy_bwd");
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
  roi_align_bwd_direct<<<mi::grid_for(total, block), block, 0, s>>>(
      total, top_grad, rois, bottom_grad, batch, channels, height, width, aligned_height,
      aligned_width, spatial_scale, sampling_ratio, st);
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
