// roi_align_device.h -- device helpers shared by the RoIAlign kernels (roi_align*.hip).
// Each function restates one piece of lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu in fp32,
// operation for operation (the library is compiled with -ffp-contract=off).
#pragma once

#include "common.h"

namespace mi {

// Geometry of one RoI, exactly as roi_align_kernel.cu:74-101 computes it.
struct RoiGeom {
  int batch_ind;
  float start_w, start_h, bin_h, bin_w;
  int grid_h, grid_w;
  float count;
};

__device__ __forceinline__ RoiGeom roi_geometry(const float* __restrict__ roi, float spatial_scale,
                                                int aligned_height, int aligned_width,
                                                int sampling_ratio) {
  RoiGeom g;
  g.batch_ind = (int)roi[0];  // :76 float -> int truncation
  g.start_w = roi[1] * spatial_scale;  // :79-82, no rounding
  g.start_h = roi[2] * spatial_scale;
  float end_w = roi[3] * spatial_scale;
  float end_h = roi[4] * spatial_scale;
  float roi_width = fmaxf(end_w - g.start_w, 1.f);  // :85-86
  float roi_height = fmaxf(end_h - g.start_h, 1.f);
  g.bin_h = roi_height / (float)aligned_height;  // :87-88
  g.bin_w = roi_width / (float)aligned_width;
  g.grid_h = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / (float)aligned_height);
  g.grid_w = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / (float)aligned_width);
  g.count = (float)(g.grid_h * g.grid_w);  // :101
  return g;
}

// One bilinear sample: taps and weights of roi_align_kernel.cu:16-58 / :150-190.
struct Taps {
  int y_low, y_high, x_low, x_high;  // -1 when the sample is outside the [-1, H] x [-1, W] band
  float w1, w2, w3, w4;
};

__device__ __forceinline__ Taps sample_taps(int height, int width, float y, float x) {
  Taps t;
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) {
    t.y_low = t.y_high = t.x_low = t.x_high = -1;
    t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
    return t;
  }
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  t.y_low = (int)y;
  t.x_low = (int)x;
  if (t.y_low >= height - 1) {
    t.y_high = t.y_low = height - 1;
    y = (float)t.y_low;
  } else {
    t.y_high = t.y_low + 1;
  }
  if (t.x_low >= width - 1) {
    t.x_high = t.x_low = width - 1;
    x = (float)t.x_low;
  } else {
    t.x_high = t.x_low + 1;
  }
  float ly = y - (float)t.y_low;
  float lx = x - (float)t.x_low;
  float hy = 1.f - ly, hx = 1.f - lx;
  t.w1 = hy * hx;
  t.w2 = hy * lx;
  t.w3 = ly * hx;
  t.w4 = ly * lx;
  return t;
}

__device__ __forceinline__ float sample_y(const RoiGeom& g, int ph, int iy) {
  return g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;  // :106-107
}
__device__ __forceinline__ float sample_x(const RoiGeom& g, int pw, int ix) {
  return g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;  // :109-110
}


// The feature maps one call works on: one level for the plain entry points, up to four FPN levels for the fused ones
// (mi_roi_align_forward_fpn / _backward_fpn).  Passed to the record kernels by value.
constexpr int kMaxLevels = 4;
struct LevelTable {
  int count;
  int height[kMaxLevels], width[kMaxLevels];
  float scale[kMaxLevels];
  const float* feat[kMaxLevels];  // forward: features of the level
  float* grad[kMaxLevels];        // backward: gradient map of the level
  int row_base[kMaxLevels + 1];   // prefix sum of batch * height: the "global row" space windows and tiles live in
  int tile_base[kMaxLevels + 1];  // backward: prefix sum of the levels' tile counts (filled by the launcher)
};
inline LevelTable single_level(const float* feat, float* grad, int batch, int height, int width, float scale) {
  LevelTable t = {};
  t.count = 1;
  t.height[0] = height;
  t.width[0] = width;
  t.scale[0] = scale;
  t.feat[0] = feat;
  t.grad[0] = grad;
  t.row_base[1] = batch * height;
  return t;
}

// two-launch forward fast path with caller scratch (roi_align_records.hip)
size_t roi_align_records_workspace_bytes(int num_rois);
void roi_align_fwd_records_set_timeline(long long* device_buffer);  // tuning builds only (no-op otherwise)
bool roi_align_fwd_records_supported(int channels, int height, int width, int num_rois, int aligned_height,
                                     int aligned_width);
int launch_roi_align_fwd_records(const float* features, const float* rois, float* output, void* workspace, int batch,
                                 int channels, int height, int width, int num_rois, int aligned_height,
                                 int aligned_width, float spatial_scale, int sampling_ratio, int cap_px, bool bwd_tables,
                                 hipStream_t stream);
// records-free forward, one launch (roi_align_records.hip: roi_align_fwd_slab); `levels` may be nullptr (level 0)
bool roi_align_fwd_slab_supported(const LevelTable& lv, int channels, int num_rois, int aligned_height, int aligned_width);
int launch_roi_align_fwd_slab(const LevelTable& lv, const float* rois, const int* levels, float* output, int batch,
                              int channels, int num_rois, int aligned_height, int aligned_width, int sampling_ratio,
                              hipStream_t stream);
// records_ready: the workspace already holds the records of THESE rois at THIS geometry (written by a forward call)
// overwrite: every element of bottom_grad is written (no zero fill needed) instead of accumulated into
// nhwc: bottom_grad is stored channels-last ([N][H][W][C]); top_grad is always dense [R][C][PH][PW]
int launch_roi_align_bwd_records(const float* top_grad, const float* rois, float* bottom_grad, void* workspace,
                                 size_t workspace_bytes, bool records_ready, bool overwrite, bool nhwc, int batch,
                                 int channels, int height, int width, int num_rois, int aligned_height, int aligned_width, float spatial_scale,
                                 int sampling_ratio, int cap_px, hipStream_t stream);
bool roi_align_bwd_records_supported(int channels, int height, int width, int num_rois, int aligned_height,
                                     int aligned_width);
// the same two paths over up to kMaxLevels feature maps in one call (`levels`: device int32 per RoI, or nullptr)
int launch_roi_align_fwd_records_levels(const LevelTable& lv, const float* rois, const int* levels, float* output,
                                        void* workspace, int batch, int channels, int num_rois, int aligned_height,
                                        int aligned_width, int sampling_ratio, int cap_px, bool bwd_tables,
                                        hipStream_t stream, bool records_ready = false);
// the records of RoIs that are still candidates of the proposal stage: row r = candidate top_idx[r]; also writes the RoI
// blob, its validity bytes and FPN levels (what mi_rpn_collect_finish writes) -- roi_align_records.hip, CollectedRois
int launch_roi_align_prepare_collected(const LevelTable& lv, const float* top_scores, const long long* top_idx,
                                       const float* cand_rois, int mark_invalid, int k_min, int k_max, float s0, float lvl0,
                                       float* rois, unsigned char* valid, int* levels, void* workspace, int batch,
                                       int num_rois, int aligned_height, int aligned_width, int sampling_ratio, int cap_px,
                                       bool bwd_tables, hipStream_t stream, int channels);
// workspace_bytes >= roi_align_bwd_workspace_bytes(): the planned backward (roi_align_bwd_plan + list slices);
// a workspace of roi_align_records_workspace_bytes() only: every tile's workgroups scan the RoIs themselves
size_t roi_align_bwd_workspace_bytes(LevelTable lv, int batch, int num_rois);
int launch_roi_align_bwd_records_levels(const float* top_grad, const float* rois, const int* levels, LevelTable lv,
                                        void* workspace, size_t workspace_bytes, bool records_ready, bool overwrite,
                                        bool nhwc, int batch, int channels, int num_rois, int aligned_height, int aligned_width,
                                        int sampling_ratio, int cap_px, hipStream_t stream);
// records only (the first launch of the two-launch paths); `workspace` as roi_align_records_workspace_bytes
// bwd_tables: also write the record's backward block (merged pass weights; +2 us) -- the caller will run a backward
int launch_roi_align_prepare(const float* rois, void* workspace, int batch, int height, int width, int num_rois,
                             int aligned_height, int aligned_width, float spatial_scale, int sampling_ratio,
                             bool bwd_tables, hipStream_t stream, int channels = 0, const float* features = nullptr);
// channels-last features, record-driven (roi_align_nhwc.hip)
void roi_align_fwd_nhwc_set_timeline(long long* device_buffer);
bool roi_align_fwd_nhwc_supported(int channels, int height, int width, int num_rois, int aligned_height,
                                  int aligned_width);
int launch_roi_align_fwd_nhwc(const float* features, const float* rois, float* output, const void* workspace,
                              int batch, int channels, int height, int width, int num_rois, int aligned_height,
                              int aligned_width, float spatial_scale, int sampling_ratio, hipStream_t stream);
int launch_roi_align_fwd_nhwc_levels(const LevelTable& lv, const float* rois, float* output, const void* workspace,
                                     int batch, int channels, int num_rois, int aligned_height, int aligned_width,
                                     int sampling_ratio, hipStream_t stream);
// records of `rois` for a table of levels (first launch of the fused paths)
int launch_roi_align_prepare_levels(const LevelTable& lv, const float* rois, const int* levels, void* workspace,
                                    int batch, int num_rois, int aligned_height, int aligned_width, int sampling_ratio,
                                    bool bwd_tables, hipStream_t stream, int channels = 0);

}  // namespace mi
