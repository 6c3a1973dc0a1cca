/*
 * oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99, fp32, compiled with -ffp-contract=off) of the arithmetic of the
 * reference's RoIAlign / RoIPool / RoICrop / NMS kernels, each function citing the reference
 * file:line it follows statement by statement.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library; the product path
 * (detectron_pytorch_amd/) never does and fails loudly when its HIP library is missing.
 *
 * Parity pin: every function here is checked against the reference's own source compiled and
 * run in the build container -- the CUDA kernels through oracle/_ref (the reference .cu files
 * compiled for the host by oracle/build_ref.py with oracle/cuda_on_cpu.h), the NMS / IoU through
 * the reference's cython_nms.pyx / cython_bbox.pyx built from /root/reference -- and against the
 * golden vectors those produced (tests/golden/, generator tests/golden/generate.py).
 *
 * All tensors are dense row-major.  `threads` > 1 parallelises with OpenMP over independent
 * units (RoIs for forward ops, channels for backward ops) without changing any result bit.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
/* uses the enclosing function's `threads` variable */
#define ORACLE_PARALLEL_FOR(nthreads) _Pragma("omp parallel for schedule(dynamic, 1) num_threads(threads)")
#else
#define ORACLE_PARALLEL_FOR(nthreads)
#endif

static int clamp_threads(int threads) { return threads < 1 ? 1 : threads; }

/* ========================================================================================
 * RoIAlign, Caffe2 semantics.  lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu
 * ====================================================================================== */

/* roi_align_kernel.cu:16-63 */
static float ra_bilinear_interpolate(const float* bottom_data, int height, int width, float y,
                                     float x) {
  if (y < -1.0 || y > height || x < -1.0 || x > width) return 0; /* :19-22 */
  if (y <= 0) y = 0;                                              /* :24-29 */
  if (x <= 0) x = 0;
  int y_low = (int)y; /* :31-32 */
  int x_low = (int)x;
  int y_high, x_high;
  if (y_low >= height - 1) { /* :36-41 */
    y_high = y_low = height - 1;
    y = (float)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= width - 1) { /* :43-48 */
    x_high = x_low = width - 1;
    x = (float)x_low;
  } else {
    x_high = x_low + 1;
  }
  float ly = y - y_low; /* :50-52 */
  float lx = x - x_low;
  float hy = 1. - ly, hx = 1. - lx;
  float v1 = bottom_data[y_low * width + x_low]; /* :54-57 */
  float v2 = bottom_data[y_low * width + x_high];
  float v3 = bottom_data[y_high * width + x_low];
  float v4 = bottom_data[y_high * width + x_high];
  float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx; /* :58 */
  float val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);          /* :60 */
  return val;
}

/* one output element; roi_align_kernel.cu:69-119 */
static float ra_forward_one(const float* bottom_data, const float* bottom_rois, int n, int c,
                            int ph, int pw, float spatial_scale, int height, int width,
                            int channels, int aligned_height, int aligned_width,
                            int sampling_ratio) {
  const float* offset_bottom_rois = bottom_rois + n * 5;
  int roi_batch_ind = offset_bottom_rois[0]; /* :76 float -> int truncation */
  float roi_start_w = offset_bottom_rois[1] * spatial_scale; /* :79-82 no rounding */
  float roi_start_h = offset_bottom_rois[2] * spatial_scale;
  float roi_end_w = offset_bottom_rois[3] * spatial_scale;
  float roi_end_h = offset_bottom_rois[4] * spatial_scale;
  float roi_width = fmaxf(roi_end_w - roi_start_w, 1.f); /* :85-86 */
  float roi_height = fmaxf(roi_end_h - roi_start_h, 1.f);
  float bin_size_h = roi_height / aligned_height; /* :87-88 */
  float bin_size_w = roi_width / aligned_width;
  const float* offset_bottom_data =
      bottom_data + (size_t)(roi_batch_ind * channels + c) * height * width; /* :90-91 */
  int roi_bin_grid_h =
      (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / aligned_height); /* :94-96 */
  int roi_bin_grid_w =
      (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / aligned_width); /* :97-98 */
  const float count = roi_bin_grid_h * roi_bin_grid_w; /* :101 */
  float output_val = 0.;
  for (int iy = 0; iy < roi_bin_grid_h; iy++) { /* :104-116 */
    const float y = roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / roi_bin_grid_h;
    for (int ix = 0; ix < roi_bin_grid_w; ix++) {
      const float x = roi_start_w + pw * bin_size_w + (ix + .5f) * bin_size_w / roi_bin_grid_w;
      float val = ra_bilinear_interpolate(offset_bottom_data, height, width, y, x);
      output_val += val;
    }
  }
  output_val /= count; /* :117 */
  return output_val;
}

/* ROIAlignForward, roi_align_kernel.cu:65-121.  features [N,C,H,W], rois [R,5], out [R,C,PH,PW] */
void oracle_roi_align_forward(const float* features, const float* rois, float* out, int batch,
                              int channels, int height, int width, int num_rois,
                              int aligned_height, int aligned_width, float spatial_scale,
                              int sampling_ratio, int threads) {
  (void)batch;
  threads = clamp_threads(threads);
  ORACLE_PARALLEL_FOR(threads)
  for (int n = 0; n < num_rois; n++)
    for (int c = 0; c < channels; c++)
      for (int ph = 0; ph < aligned_height; ph++)
        for (int pw = 0; pw < aligned_width; pw++)
          out[(((size_t)n * channels + c) * aligned_height + ph) * aligned_width + pw] =
              ra_forward_one(features, rois, n, c, ph, pw, spatial_scale, height, width, channels,
                             aligned_height, aligned_width, sampling_ratio);
}

/* roi_align_kernel.cu:150-193 */
static void ra_bilinear_interpolate_gradient(int height, int width, float y, float x, float* w1,
                                             float* w2, float* w3, float* w4, int* x_low,
                                             int* x_high, int* y_low, int* y_high) {
  if (y < -1.0 || y > height || x < -1.0 || x > width) { /* :155-160 */
    *w1 = *w2 = *w3 = *w4 = 0.;
    *x_low = *x_high = *y_low = *y_high = -1;
    return;
  }
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  *y_low = (int)y;
  *x_low = (int)x;
  if (*y_low >= height - 1) {
    *y_high = *y_low = height - 1;
    y = (float)*y_low;
  } else {
    *y_high = *y_low + 1;
  }
  if (*x_low >= width - 1) {
    *x_high = *x_low = width - 1;
    x = (float)*x_low;
  } else {
    *x_high = *x_low + 1;
  }
  float ly = y - *y_low;
  float lx = x - *x_low;
  float hy = 1. - ly, hx = 1. - lx;
  *w1 = hy * hx, *w2 = hy * lx, *w3 = ly * hx, *w4 = ly * lx; /* :190 */
}

/* ROIAlignBackward, roi_align_kernel.cu:195-270.  Accumulates into bottom_diff [N,C,H,W]
 * (caller zero-fills, functions/roi_align.py:39-40).  The reference's atomicAdd order is
 * unspecified; this restatement adds in (n, ph, pw, iy, ix, tap) order per channel. */
void oracle_roi_align_backward(const float* top_diff, const float* rois, float* bottom_diff,
                               int batch, int channels, int height, int width, int num_rois,
                               int aligned_height, int aligned_width, float spatial_scale,
                               int sampling_ratio, int threads) {
  (void)batch;
  threads = clamp_threads(threads);
  ORACLE_PARALLEL_FOR(threads)
  for (int c = 0; c < channels; c++) {
    for (int n = 0; n < num_rois; n++) {
      const float* offset_bottom_rois = rois + n * 5;
      int roi_batch_ind = offset_bottom_rois[0];
      float roi_start_w = offset_bottom_rois[1] * spatial_scale;
      float roi_start_h = offset_bottom_rois[2] * spatial_scale;
      float roi_end_w = offset_bottom_rois[3] * spatial_scale;
      float roi_end_h = offset_bottom_rois[4] * spatial_scale;
      float roi_width = fmaxf(roi_end_w - roi_start_w, 1.f);
      float roi_height = fmaxf(roi_end_h - roi_start_h, 1.f);
      float bin_size_h = roi_height / aligned_height;
      float bin_size_w = roi_width / aligned_width;
      float* offset_bottom_diff =
          bottom_diff + (size_t)(roi_batch_ind * channels + c) * height * width;
      const float* offset_top_diff =
          top_diff + (size_t)(n * channels + c) * aligned_height * aligned_width; /* :223-224 */
      int roi_bin_grid_h =
          (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / aligned_height);
      int roi_bin_grid_w =
          (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / aligned_width);
      const float count = roi_bin_grid_h * roi_bin_grid_w;
      for (int ph = 0; ph < aligned_height; ph++)
        for (int pw = 0; pw < aligned_width; pw++) {
          const float top_diff_this_bin = offset_top_diff[ph * aligned_width + pw]; /* :225 */
          for (int iy = 0; iy < roi_bin_grid_h; iy++) {
            const float y =
                roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / roi_bin_grid_h;
            for (int ix = 0; ix < roi_bin_grid_w; ix++) {
              const float x =
                  roi_start_w + pw * bin_size_w + (ix + .5f) * bin_size_w / roi_bin_grid_w;
              float w1, w2, w3, w4;
              int x_low, x_high, y_low, y_high;
              ra_bilinear_interpolate_gradient(height, width, y, x, &w1, &w2, &w3, &w4, &x_low,
                                               &x_high, &y_low, &y_high);
              float g1 = top_diff_this_bin * w1 / count; /* :252-255 */
              float g2 = top_diff_this_bin * w2 / count;
              float g3 = top_diff_this_bin * w3 / count;
              float g4 = top_diff_this_bin * w4 / count;
              if (x_low >= 0 && x_high >= 0 && y_low >= 0 && y_high >= 0) { /* :257-266 */
                offset_bottom_diff[y_low * width + x_low] += g1;
                offset_bottom_diff[y_low * width + x_high] += g2;
                offset_bottom_diff[y_high * width + x_low] += g3;
                offset_bottom_diff[y_high * width + x_high] += g4;
              }
            }
          }
        }
    }
  }
}

/* Number of distinct feature pixels (n,y,x) that some sample of some RoI reads with a non-zero
 * weight -- the `U` of SURVEY.md section 8(d)'s algorithmic-bytes formula
 * (fwd bytes = 4*R*C*PH*PW + 4*C*U + 20*R).  Same sampling arithmetic as the forward. */
int64_t oracle_roi_align_touched_pixels(const float* rois, int batch, int height, int width,
                                        int num_rois, int aligned_height, int aligned_width,
                                        float spatial_scale, int sampling_ratio) {
  size_t total = (size_t)batch * height * width;
  unsigned char* seen = (unsigned char*)calloc(total ? total : 1, 1);
  for (int n = 0; n < num_rois; n++) {
    const float* r = rois + n * 5;
    int b = r[0];
    if (b < 0 || b >= batch) continue;
    float roi_start_w = r[1] * spatial_scale, roi_start_h = r[2] * spatial_scale;
    float roi_end_w = r[3] * spatial_scale, roi_end_h = r[4] * spatial_scale;
    float roi_width = fmaxf(roi_end_w - roi_start_w, 1.f);
    float roi_height = fmaxf(roi_end_h - roi_start_h, 1.f);
    float bin_size_h = roi_height / aligned_height, bin_size_w = roi_width / aligned_width;
    int gh = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / aligned_height);
    int gw = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / aligned_width);
    for (int ph = 0; ph < aligned_height; ph++)
      for (int pw = 0; pw < aligned_width; pw++)
        for (int iy = 0; iy < gh; iy++)
          for (int ix = 0; ix < gw; ix++) {
            float y = roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / gh;
            float x = roi_start_w + pw * bin_size_w + (ix + .5f) * bin_size_w / gw;
            float w1, w2, w3, w4;
            int xl, xh, yl, yh;
            ra_bilinear_interpolate_gradient(height, width, y, x, &w1, &w2, &w3, &w4, &xl, &xh,
                                             &yl, &yh);
            if (xl < 0) continue;
            unsigned char* s = seen + (size_t)b * height * width;
            if (w1 != 0.f) s[yl * width + xl] = 1;
            if (w2 != 0.f) s[yl * width + xh] = 1;
            if (w3 != 0.f) s[yh * width + xl] = 1;
            if (w4 != 0.f) s[yh * width + xh] = 1;
          }
  }
  int64_t u = 0;
  for (size_t i = 0; i < total; i++) u += seen[i];
  free(seen);
  return u;
}

/* ========================================================================================
 * RoIAlign, legacy jwyang semantics.  lib/model/roi_align/src/roi_align_kernel.cu
 * (double-typed literals in the reference make parts of the arithmetic fp64; kept as written)
 * ====================================================================================== */

/* ROIAlignForward, model/roi_align/src/roi_align_kernel.cu:15-70 */
void oracle_roi_align_legacy_forward(const float* bottom_data, const float* bottom_rois,
                                     float* top_data, int batch, int channels, int height,
                                     int width, int num_rois, int aligned_height,
                                     int aligned_width, float spatial_scale, int threads) {
  (void)batch;
  threads = clamp_threads(threads);
  ORACLE_PARALLEL_FOR(threads)
  for (int n = 0; n < num_rois; n++)
    for (int c = 0; c < channels; c++)
      for (int ph = 0; ph < aligned_height; ph++)
        for (int pw = 0; pw < aligned_width; pw++) {
          size_t index = (((size_t)n * channels + c) * aligned_height + ph) * aligned_width + pw;
          float roi_batch_ind = bottom_rois[n * 5 + 0]; /* :32 (kept as float) */
          float roi_start_w = bottom_rois[n * 5 + 1] * spatial_scale;
          float roi_start_h = bottom_rois[n * 5 + 2] * spatial_scale;
          float roi_end_w = bottom_rois[n * 5 + 3] * spatial_scale;
          float roi_end_h = bottom_rois[n * 5 + 4] * spatial_scale;
          float roi_width = fmaxf(roi_end_w - roi_start_w + 1., 0.); /* :39-40 */
          float roi_height = fmaxf(roi_end_h - roi_start_h + 1., 0.);
          float bin_size_h = roi_height / (aligned_height - 1.); /* :41-42 */
          float bin_size_w = roi_width / (aligned_width - 1.);
          float h = (float)(ph)*bin_size_h + roi_start_h; /* :44-45 */
          float w = (float)(pw)*bin_size_w + roi_start_w;
          int hstart = fminf(floorf(h), height - 2); /* :47-48 */
          int wstart = fminf(floorf(w), width - 2);
          int img_start = roi_batch_ind * channels * height * width; /* :50 float product */
          if (h < 0 || h >= height || w < 0 || w >= width) { /* :53-54 */
            top_data[index] = 0.;
          } else {
            float h_ratio = h - (float)(hstart); /* :56-57 */
            float w_ratio = w - (float)(wstart);
            int upleft = img_start + (c * height + hstart) * width + wstart; /* :58-61 */
            int upright = upleft + 1;
            int downleft = upleft + width;
            int downright = downleft + 1;
            top_data[index] = bottom_data[upleft] * (1. - h_ratio) * (1. - w_ratio) /* :63-66 */
                              + bottom_data[upright] * (1. - h_ratio) * w_ratio +
                              bottom_data[downleft] * h_ratio * (1. - w_ratio) +
                              bottom_data[downright] * h_ratio * w_ratio;
          }
        }
}

/* ROIAlignBackward, model/roi_align/src/roi_align_kernel.cu:94-143 */
void oracle_roi_align_legacy_backward(const float* top_diff, const float* bottom_rois,
                                      float* bottom_diff, int batch, int channels, int height,
                                      int width, int num_rois, int aligned_height,
                                      int aligned_width, float spatial_scale, int threads) {
  (void)batch;
  threads = clamp_threads(threads);
  ORACLE_PARALLEL_FOR(threads)
  for (int c = 0; c < channels; c++)
    for (int n = 0; n < num_rois; n++)
      for (int ph = 0; ph < aligned_height; ph++)
        for (int pw = 0; pw < aligned_width; pw++) {
          size_t index = (((size_t)n * channels + c) * aligned_height + ph) * aligned_width + pw;
          float roi_batch_ind = bottom_rois[n * 5 + 0];
          float roi_start_w = bottom_rois[n * 5 + 1] * spatial_scale;
          float roi_start_h = bottom_rois[n * 5 + 2] * spatial_scale;
          float roi_end_w = bottom_rois[n * 5 + 3] * spatial_scale;
          float roi_end_h = bottom_rois[n * 5 + 4] * spatial_scale;
          float roi_width = fmaxf(roi_end_w - roi_start_w + 1., 0.);
          float roi_height = fmaxf(roi_end_h - roi_start_h + 1., 0.);
          float bin_size_h = roi_height / (aligned_height - 1.);
          float bin_size_w = roi_width / (aligned_width - 1.);
          float h = (float)(ph)*bin_size_h + roi_start_h;
          float w = (float)(pw)*bin_size_w + roi_start_w;
          int hstart = fminf(floorf(h), height - 2);
          int wstart = fminf(floorf(w), width - 2);
          int img_start = roi_batch_ind * channels * height * width;
          if (!(h < 0 || h >= height || w < 0 || w >= width)) { /* :127 */
            float h_ratio = h - (float)(hstart);
            float w_ratio = w - (float)(wstart);
            int upleft = img_start + (c * height + hstart) * width + wstart;
            int upright = upleft + 1;
            int downleft = upleft + width;
            int downright = downleft + 1;
            bottom_diff[upleft] += (float)(top_diff[index] * (1. - h_ratio) * (1 - w_ratio)); /* :135-138 */
            bottom_diff[upright] += (float)(top_diff[index] * (1. - h_ratio) * w_ratio);
            bottom_diff[downleft] += (float)(top_diff[index] * h_ratio * (1 - w_ratio));
            bottom_diff[downright] += (float)(top_diff[index] * h_ratio * w_ratio);
          }
        }
}

/* ========================================================================================
 * RoIPool.  lib/model/roi_pooling/src/roi_pooling_kernel.cu
 * ====================================================================================== */

/* ROIPoolForward, roi_pooling_kernel.cu:24-93.  argmax may be NULL (:90-91). */
void oracle_roi_pool_forward(const float* bottom_data, const float* bottom_rois, float* top_data,
                             int32_t* argmax_data, int batch, int channels, int height, int width,
                             int num_rois, int pooled_height, int pooled_width,
                             float spatial_scale, int threads) {
  (void)batch;
  threads = clamp_threads(threads);
  ORACLE_PARALLEL_FOR(threads)
  for (int n = 0; n < num_rois; n++)
    for (int c = 0; c < channels; c++)
      for (int ph = 0; ph < pooled_height; ph++)
        for (int pw = 0; pw < pooled_width; pw++) {
          size_t index = (((size_t)n * channels + c) * pooled_height + ph) * pooled_width + pw;
          int roi_batch_ind = bottom_rois[n * 5 + 0];                    /* :45 */
          int roi_start_w = roundf(bottom_rois[n * 5 + 1] * spatial_scale); /* :46-49 */
          int roi_start_h = roundf(bottom_rois[n * 5 + 2] * spatial_scale);
          int roi_end_w = roundf(bottom_rois[n * 5 + 3] * spatial_scale);
          int roi_end_h = roundf(bottom_rois[n * 5 + 4] * spatial_scale);
          int roi_width = fmaxf(roi_end_w - roi_start_w + 1, 1); /* :52-53 */
          int roi_height = fmaxf(roi_end_h - roi_start_h + 1, 1);
          float bin_size_h = (float)(roi_height) / (float)(pooled_height); /* :54-55 */
          float bin_size_w = (float)(roi_width) / (float)(pooled_width);
          int hstart = (int)(floorf((float)(ph)*bin_size_h)); /* :57-60 */
          int wstart = (int)(floorf((float)(pw)*bin_size_w));
          int hend = (int)(ceilf((float)(ph + 1) * bin_size_h));
          int wend = (int)(ceilf((float)(pw + 1) * bin_size_w));
          hstart = fminf(fmaxf(hstart + roi_start_h, 0), height); /* :63-66 */
          hend = fminf(fmaxf(hend + roi_start_h, 0), height);
          wstart = fminf(fmaxf(wstart + roi_start_w, 0), width);
          wend = fminf(fmaxf(wend + roi_start_w, 0), width);
          int is_empty = (hend <= hstart) || (wend <= wstart); /* :67 */
          float maxval = is_empty ? 0 : -FLT_MAX;              /* :70 */
          int maxidx = -1;                                     /* :72 */
          int bottom_data_batch_offset = roi_batch_ind * channels * height * width; /* :75-76 */
          int bottom_data_offset = bottom_data_batch_offset + c * height * width;
          for (int h = hstart; h < hend; ++h)
            for (int w = wstart; w < wend; ++w) {
              int bottom_index = h * width + w; /* :82 */
              if (bottom_data[bottom_data_offset + bottom_index] > maxval) { /* :83 strict > */
                maxval = bottom_data[bottom_data_offset + bottom_index];
                maxidx = bottom_data_offset + bottom_index; /* :85 flat index into whole tensor */
              }
            }
          top_data[index] = maxval;
          if (argmax_data != NULL) argmax_data[index] = maxidx;
        }
}

/* ROIPoolBackward, roi_pooling_kernel.cu:128-203: one gather per INPUT element over all RoIs.
 * Overwrites bottom_diff (:202). */
void oracle_roi_pool_backward(const float* top_diff, const float* bottom_rois,
                              const int32_t* argmax_data, float* bottom_diff, int batch,
                              int channels, int height, int width, int num_rois,
                              int pooled_height, int pooled_width, float spatial_scale,
                              int threads) {
  threads = clamp_threads(threads);
  int64_t nthreads = (int64_t)batch * channels * height * width;
  ORACLE_PARALLEL_FOR(threads)
  for (int64_t index = 0; index < nthreads; index++) {
    int n = (int)index; /* :136-142 */
    int w = n % width;
    n /= width;
    int h = n % height;
    n /= height;
    int c = n % channels;
    n /= channels;
    float gradient = 0;
    for (int roi_n = 0; roi_n < num_rois; ++roi_n) { /* :146 */
      const float* offset_bottom_rois = bottom_rois + roi_n * 5;
      int roi_batch_ind = offset_bottom_rois[0];
      if (n != roi_batch_ind) continue; /* :151-153 */
      int roi_start_w = roundf(offset_bottom_rois[1] * spatial_scale);
      int roi_start_h = roundf(offset_bottom_rois[2] * spatial_scale);
      int roi_end_w = roundf(offset_bottom_rois[3] * spatial_scale);
      int roi_end_h = roundf(offset_bottom_rois[4] * spatial_scale);
      const int in_roi =
          (w >= roi_start_w && w <= roi_end_w && h >= roi_start_h && h <= roi_end_h); /* :161-165 */
      if (!in_roi) continue;
      int offset = roi_n * pooled_height * pooled_width * channels; /* :167-169 */
      const float* offset_top_diff = top_diff + offset;
      const int32_t* offset_argmax_data = argmax_data + offset;
      int roi_width = fmaxf(roi_end_w - roi_start_w + 1, 1); /* :175-176 */
      int roi_height = fmaxf(roi_end_h - roi_start_h + 1, 1);
      float bin_size_h = (float)(roi_height) / (float)(pooled_height);
      float bin_size_w = (float)(roi_width) / (float)(pooled_width);
      int phstart = floorf((float)(h - roi_start_h) / bin_size_h); /* :181-184 */
      int phend = ceilf((float)(h - roi_start_h + 1) / bin_size_h);
      int pwstart = floorf((float)(w - roi_start_w) / bin_size_w);
      int pwend = ceilf((float)(w - roi_start_w + 1) / bin_size_w);
      phstart = fminf(fmaxf(phstart, 0), pooled_height); /* :186-189 */
      phend = fminf(fmaxf(phend, 0), pooled_height);
      pwstart = fminf(fmaxf(pwstart, 0), pooled_width);
      pwend = fminf(fmaxf(pwend, 0), pooled_width);
      for (int ph = phstart; ph < phend; ++ph)
        for (int pw = pwstart; pw < pwend; ++pw)
          if (offset_argmax_data[(c * pooled_height + ph) * pooled_width + pw] == index) /* :193 */
            gradient += offset_top_diff[(c * pooled_height + ph) * pooled_width + pw];
    }
    bottom_diff[index] = gradient;
  }
}

/* ========================================================================================
 * RoICrop (bilinear grid sampler).  lib/model/roi_crop/src/roi_crop_cuda_kernel.cu
 * dense NCHW input [N,C,H,W], grid [R,GH,GW,2] (y,x), output [R,C,GH,GW]
 * ====================================================================================== */

/* getTopLeft, roi_crop_cuda_kernel.cu:11-22 */
static void rc_get_top_left(float x, int width, int* point, float* weight) {
  float xcoord = (x + 1) * (width - 1) / 2;
  *point = floorf(xcoord);
  *weight = 1 - (xcoord - *point);
}
static int rc_between(int value, int lo, int hi) { return value >= lo && value <= hi; } /* :24-27 */

/* bilinearSamplingFromGrid, roi_crop_cuda_kernel.cu:47-109 (strides of dense tensors as passed by
 * roi_crop_cuda.c:23-44).  Elements whose four taps are all outside stay untouched (:92-93). */
void oracle_roi_crop_forward(const float* input, const float* grids, float* output, int batch,
                             int channels, int height, int width, int num_rois, int gh, int gw,
                             int threads) {
  threads = clamp_threads(threads);
  int roiPerImage = num_rois / batch; /* :217 */
  ORACLE_PARALLEL_FOR(threads)
  for (int b = 0; b < num_rois; b++)
    for (int cOut = 0; cOut < channels; cOut++)
      for (int yOut = 0; yOut < gh; yOut++)
        for (int xOut = 0; xOut < gw; xOut++) {
          const int b_input = b / roiPerImage; /* :64 */
          float yf = grids[((size_t)(b * gh + yOut) * gw + xOut) * 2];     /* :66 */
          float xf = grids[((size_t)(b * gh + yOut) * gw + xOut) * 2 + 1]; /* :67 */
          int yInTopLeft, xInTopLeft;
          float yWeightTopLeft, xWeightTopLeft;
          rc_get_top_left(xf, width, &xInTopLeft, &xWeightTopLeft);
          rc_get_top_left(yf, height, &yInTopLeft, &yWeightTopLeft);
          const size_t outAddress = (((size_t)b * channels + cOut) * gh + yOut) * gw + xOut;
          const long inTopLeftAddress =
              ((long)(b_input * channels + cOut) * height + yInTopLeft) * width + xInTopLeft;
          const long inTopRightAddress = inTopLeftAddress + 1;
          const long inBottomLeftAddress = inTopLeftAddress + width;
          const long inBottomRightAddress = inBottomLeftAddress + 1;
          float v = 0, inTopLeft = 0, inTopRight = 0, inBottomLeft = 0, inBottomRight = 0;
          int topLeftIsIn = rc_between(xInTopLeft, 0, width - 1) && rc_between(yInTopLeft, 0, height - 1);
          int topRightIsIn = rc_between(xInTopLeft + 1, 0, width - 1) && rc_between(yInTopLeft, 0, height - 1);
          int bottomLeftIsIn = rc_between(xInTopLeft, 0, width - 1) && rc_between(yInTopLeft + 1, 0, height - 1);
          int bottomRightIsIn = rc_between(xInTopLeft + 1, 0, width - 1) && rc_between(yInTopLeft + 1, 0, height - 1);
          if (!topLeftIsIn && !topRightIsIn && !bottomLeftIsIn && !bottomRightIsIn) continue;
          if (topLeftIsIn) inTopLeft = input[inTopLeftAddress];
          if (topRightIsIn) inTopRight = input[inTopRightAddress];
          if (bottomLeftIsIn) inBottomLeft = input[inBottomLeftAddress];
          if (bottomRightIsIn) inBottomRight = input[inBottomRightAddress];
          v = xWeightTopLeft * yWeightTopLeft * inTopLeft /* :100-103 */
              + (1 - xWeightTopLeft) * yWeightTopLeft * inTopRight +
              xWeightTopLeft * (1 - yWeightTopLeft) * inBottomLeft +
              (1 - xWeightTopLeft) * (1 - yWeightTopLeft) * inBottomRight;
          output[outAddress] = v;
        }
}

/* backwardBilinearSampling, roi_crop_cuda_kernel.cu:111-194: accumulates into grad_input only;
 * the grid gradient is computed nowhere in the reference and stays as the caller left it. */
void oracle_roi_crop_backward(const float* input, const float* grids, const float* grad_output,
                              float* grad_input, int batch, int channels, int height, int width,
                              int num_rois, int gh, int gw, int threads) {
  (void)input;
  threads = clamp_threads(threads);
  int roiPerImage = num_rois / batch;
  ORACLE_PARALLEL_FOR(threads)
  for (int cOut = 0; cOut < channels; cOut++)
    for (int b = 0; b < num_rois; b++)
      for (int yOut = 0; yOut < gh; yOut++)
        for (int xOut = 0; xOut < gw; xOut++) {
          const int b_input = b / roiPerImage;
          float yf = grids[((size_t)(b * gh + yOut) * gw + xOut) * 2];
          float xf = grids[((size_t)(b * gh + yOut) * gw + xOut) * 2 + 1];
          int yInTopLeft, xInTopLeft;
          float yWeightTopLeft, xWeightTopLeft;
          rc_get_top_left(xf, width, &xInTopLeft, &xWeightTopLeft);
          rc_get_top_left(yf, height, &yInTopLeft, &yWeightTopLeft);
          const long tl = ((long)(b_input * channels + cOut) * height + yInTopLeft) * width + xInTopLeft;
          const long tr = tl + 1, bl = tl + width, br = bl + 1;
          int topLeftIsIn = rc_between(xInTopLeft, 0, width - 1) && rc_between(yInTopLeft, 0, height - 1);
          int topRightIsIn = rc_between(xInTopLeft + 1, 0, width - 1) && rc_between(yInTopLeft, 0, height - 1);
          int bottomLeftIsIn = rc_between(xInTopLeft, 0, width - 1) && rc_between(yInTopLeft + 1, 0, height - 1);
          int bottomRightIsIn = rc_between(xInTopLeft + 1, 0, width - 1) && rc_between(yInTopLeft + 1, 0, height - 1);
          float gradOutValue = grad_output[(((size_t)b * channels + cOut) * gh + yOut) * gw + xOut];
          if (topLeftIsIn) grad_input[tl] += xWeightTopLeft * yWeightTopLeft * gradOutValue; /* :169 */
          if (topRightIsIn) grad_input[tr] += (1 - xWeightTopLeft) * yWeightTopLeft * gradOutValue;
          if (bottomLeftIsIn) grad_input[bl] += xWeightTopLeft * (1 - yWeightTopLeft) * gradOutValue;
          if (bottomRightIsIn)
            grad_input[br] += (1 - xWeightTopLeft) * (1 - yWeightTopLeft) * gradOutValue;
        }
}

/* ========================================================================================
 * NMS
 * ====================================================================================== */

static float f32max(float a, float b) { return a >= b ? a : b; } /* cython_nms.pyx:28-29 */
static float f32min(float a, float b) { return a <= b ? a : b; } /* :31-32 */

typedef struct {
  float score;
  int idx;
} oracle_sort_item;

/* ascending by (score, index); reversed by the caller: np.argsort(kind='stable')[::-1] */
static int oracle_sort_cmp(const void* pa, const void* pb) {
  const oracle_sort_item* a = (const oracle_sort_item*)pa;
  const oracle_sort_item* b = (const oracle_sort_item*)pb;
  if (a->score < b->score) return -1;
  if (a->score > b->score) return 1;
  return (a->idx > b->idx) - (a->idx < b->idx);
}

/* utils.cython_nms.nms, lib/utils/cython_nms.pyx:37-87.  dets [n,5]; writes the kept ORIGINAL
 * indices in ascending order (np.where(suppressed == 0)[0], :87) to keep[], returns their count.
 * The reference's `scores.argsort()[::-1]` (:45) has no defined order for tied scores; this
 * restatement fixes the rule to descending score, then descending index. */
int oracle_nms_cython(const float* dets, int ndets, float thresh, int64_t* keep) {
  if (ndets <= 0) return 0;
  float* areas = (float*)malloc(sizeof(float) * ndets);
  int* order = (int*)malloc(sizeof(int) * ndets);
  int* suppressed = (int*)calloc(ndets, sizeof(int));
  oracle_sort_item* items = (oracle_sort_item*)malloc(sizeof(oracle_sort_item) * ndets);
  for (int i = 0; i < ndets; i++) {
    const float* d = dets + i * 5;
    areas[i] = (d[2] - d[0] + 1) * (d[3] - d[1] + 1); /* :44 */
    items[i].score = d[4];
    items[i].idx = i;
  }
  qsort(items, ndets, sizeof(oracle_sort_item), oracle_sort_cmp);
  for (int i = 0; i < ndets; i++) order[i] = items[ndets - 1 - i].idx; /* :45 */
  for (int _i = 0; _i < ndets; _i++) { /* :62-85 */
    int i = order[_i];
    if (suppressed[i] == 1) continue;
    float ix1 = dets[i * 5 + 0], iy1 = dets[i * 5 + 1], ix2 = dets[i * 5 + 2],
          iy2 = dets[i * 5 + 3];
    float iarea = areas[i];
    for (int _j = _i + 1; _j < ndets; _j++) {
      int j = order[_j];
      if (suppressed[j] == 1) continue;
      float xx1 = f32max(ix1, dets[j * 5 + 0]);
      float yy1 = f32max(iy1, dets[j * 5 + 1]);
      float xx2 = f32min(ix2, dets[j * 5 + 2]);
      float yy2 = f32min(iy2, dets[j * 5 + 3]);
      float w = f32max(0.0f, xx2 - xx1 + 1); /* :80-81 */
      float h = f32max(0.0f, yy2 - yy1 + 1);
      float inter = w * h;                            /* :82 */
      float ovr = inter / (iarea + areas[j] - inter); /* :83 */
      if (ovr >= thresh) suppressed[j] = 1;           /* :84 */
    }
  }
  int k = 0;
  for (int i = 0; i < ndets; i++)
    if (suppressed[i] == 0) keep[k++] = i;
  free(areas);
  free(order);
  free(suppressed);
  free(items);
  return k;
}

/* utils.cython_nms.soft_nms, lib/utils/cython_nms.pyx:98-203.  boxes_in [n,5] (x1,y1,x2,y2,score);
 * `boxes` [n,5] and `inds` [n] receive the working copies (:108,:116); the first <return value> rows
 * are the result (boxes[:N], inds[:N], :203).  method 0 = hard, 1 = linear, 2 = gaussian (:177-190).
 * Typing follows the Cython declarations (:110-115) AND the C that Cython generates from them: variables are C
 * floats, but integer literals inside float expressions become double constants (see below); the gaussian
 * weight is np.exp of a Python float, i.e. exp in double of the float argument, cast back to float. */
int oracle_soft_nms(const float* boxes_in, int n, float sigma, float Nt, float threshold, int method,
                    float* boxes, int64_t* inds) {
  unsigned int N = (unsigned int)n; /* :109 */
  memcpy(boxes, boxes_in, sizeof(float) * 5 * (size_t)n);
  for (int i = 0; i < n; i++) inds[i] = i; /* :116 */
  for (int i = 0; i < n; i++) {            /* :118, range(N) evaluated once */
    float maxscore = boxes[i * 5 + 4];     /* :119-120 */
    int maxpos = i;
    float tx1 = boxes[i * 5 + 0], ty1 = boxes[i * 5 + 1], tx2 = boxes[i * 5 + 2], ty2 = boxes[i * 5 + 3];
    float ts = boxes[i * 5 + 4];
    int64_t ti = inds[i];
    int pos = i + 1;
    while (pos < (int)N) { /* :131-135: first maximum wins (strict <) */
      if (maxscore < boxes[pos * 5 + 4]) {
        maxscore = boxes[pos * 5 + 4];
        maxpos = pos;
      }
      pos = pos + 1;
    }
    for (int k = 0; k < 5; k++) boxes[i * 5 + k] = boxes[maxpos * 5 + k]; /* :138-143 */
    inds[i] = inds[maxpos];
    boxes[maxpos * 5 + 0] = tx1; /* :146-151 */
    boxes[maxpos * 5 + 1] = ty1;
    boxes[maxpos * 5 + 2] = tx2;
    boxes[maxpos * 5 + 3] = ty2;
    boxes[maxpos * 5 + 4] = ts;
    inds[maxpos] = ti;
    tx1 = boxes[i * 5 + 0]; /* :153-157 */
    ty1 = boxes[i * 5 + 1];
    tx2 = boxes[i * 5 + 2];
    ty2 = boxes[i * 5 + 3];
    pos = i + 1;
    while (pos < (int)N) { /* :162-201 */
      float x1 = boxes[pos * 5 + 0], y1 = boxes[pos * 5 + 1], x2 = boxes[pos * 5 + 2], y2 = boxes[pos * 5 + 3];
      /* Cython turns the literal 1 of these float expressions into the double constant 1.0 (both in the reference's
       * shipped cython_nms.c:3882-3938 and in a fresh cythonization), so the sums and the products below are
       * evaluated in double and rounded to float on assignment. */
      float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0)); /* :169 */
      float iw = (float)((double)(f32min(tx2, x2) - f32max(tx1, x1)) + 1.0);       /* :170 */
      if (iw > 0) {
        float ih = (float)((double)(f32min(ty2, y2) - f32max(ty1, y1)) + 1.0); /* :172 */
        if (ih > 0) {
          float ua = (float)(((((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0)) + (double)area) -
                             (double)(iw * ih)); /* :174 */
          float ov = iw * ih / ua;                                                          /* :175 */
          float weight;
          if (method == 1) { /* :177-181 */
            if (ov > Nt) weight = (float)(1.0 - (double)ov);
            else weight = 1;
          } else if (method == 2) { /* :182-183 */
            weight = (float)exp((double)(-(ov * ov) / sigma));
          } else { /* :184-188 */
            if (ov > Nt) weight = 0;
            else weight = 1;
          }
          boxes[pos * 5 + 4] = weight * boxes[pos * 5 + 4]; /* :190 */
          if (boxes[pos * 5 + 4] < threshold) {             /* :194-201: swap with the last box, shrink */
            for (int k = 0; k < 5; k++) boxes[pos * 5 + k] = boxes[(N - 1) * 5 + k];
            inds[pos] = inds[N - 1];
            N = N - 1;
            pos = pos - 1;
          }
        }
      }
      pos = pos + 1;
    }
  }
  return (int)N;
}

/* devIoU, lib/model/nms/src/nms_cuda_kernel.cu:31-39 */
static float nms_dev_iou(const float* a, const float* b) {
  float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  float interS = width * height;
  float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return interS / (Sa + Sb - interS);
}

/* nms_kernel + nms_cuda_compute, nms_cuda_kernel.cu:41-161: 64x64 bitmask tiles (strict >,
 * diagonal tiles start at j = i + 1) then the sequential greedy OR-reduce.  boxes [n,5] already
 * sorted by descending score; keep[] receives positions in that input; returns the count. */
int oracle_nms_gpu_semantics(const float* boxes, int n, float thresh, int32_t* keep) {
  if (n <= 0) return 0;
  const int tpb = 64;
  int col_blocks = n / tpb + (n % tpb > 0); /* DIVUP :28 */
  uint64_t* mask = (uint64_t*)calloc((size_t)n * col_blocks, sizeof(uint64_t));
  for (int i = 0; i < n; i++)
    for (int cb = 0; cb < col_blocks; cb++) {
      int col_size = (n - cb * tpb) < tpb ? (n - cb * tpb) : tpb;
      uint64_t t = 0;
      int start = (i / tpb == cb) ? (i % tpb) + 1 : 0; /* :73-76 */
      for (int j = start; j < col_size; j++)
        if (nms_dev_iou(boxes + i * 5, boxes + (cb * tpb + j) * 5) > thresh) t |= 1ULL << j; /* :78 */
      mask[(size_t)i * col_blocks + cb] = t;
    }
  uint64_t* remv = (uint64_t*)calloc(col_blocks, sizeof(uint64_t));
  int num_to_keep = 0;
  for (int i = 0; i < n; i++) { /* :132-144 */
    int nblock = i / tpb, inblock = i % tpb;
    if (!(remv[nblock] & (1ULL << inblock))) {
      keep[num_to_keep++] = i;
      uint64_t* p = mask + (size_t)i * col_blocks;
      for (int j = nblock; j < col_blocks; j++) remv[j] |= p[j];
    }
  }
  free(mask);
  free(remv);
  return num_to_keep;
}

/* utils.cython_bbox.bbox_overlaps, lib/utils/cython_bbox.pyx:32-73.  boxes [N,4], query [K,4],
 * overlaps [N,K] (zero where the boxes do not intersect).
 * Cython types the literal `1` in `x2 - x1 + 1` as a C double, so the generated C evaluates the
 * box areas and the union in fp64 and rounds to fp32 only on assignment to the DTYPE_t
 * variables; the casts below reproduce that (checked bit for bit against the built module). */
void oracle_bbox_overlaps(const float* boxes, int N, const float* query_boxes, int K,
                          float* overlaps) {
  memset(overlaps, 0, sizeof(float) * (size_t)N * K);
  for (int k = 0; k < K; k++) {
    const float* q = query_boxes + k * 4;
    float box_area = (float)(((double)(q[2] - q[0]) + 1.0) * ((double)(q[3] - q[1]) + 1.0)); /* :52-55 */
    for (int n = 0; n < N; n++) {
      const float* b = boxes + n * 4;
      float iw = (float)((double)(f32min(b[2], q[2]) - f32max(b[0], q[0])) + 1.0); /* :57-60 */
      if (iw > 0) {
        float ih = (float)((double)(f32min(b[3], q[3]) - f32max(b[1], q[1])) + 1.0); /* :62-65 */
        if (ih > 0) {
          float ua = (float)(((((double)(b[2] - b[0]) + 1.0) * ((double)(b[3] - b[1]) + 1.0)) +
                              (double)box_area) -
                             (double)(iw * ih)); /* :67-71 */
          overlaps[(size_t)n * K + k] = iw * ih / ua; /* :72 */
        }
      }
    }
  }
}

/* ---- polygon -> binary mask: pycocotools 2.0, common/maskApi.c rleFrPoly followed by rleDecode ------------------------
 * THIRD-PARTY ALGORITHM, RESTATED (pycocotools is not installed here and is not part of /root/reference; the reference
 * calls it through mask_util.frPyObjects / mask_util.decode in lib/utils/segms.py:66-67,114-115).  PARITY UNPINNED against
 * the real package; pinned are the call sites around it (oracle/mask_targets.py) and hand-derived vectors (tests/test_oracle_cpu.py).
 * The published procedure, in its own order:
 *   1. vertices are scaled by 5 and rounded with (int)(5 v + .5) -- a C cast, i.e. truncation towards zero;
 *   2. every edge (closed polygon) is walked along its longer axis, one point per unit step, the other coordinate
 *      (int)(start + slope * t + .5) in double precision; the points are emitted from the edge's first vertex to its second;
 *   3. wherever two consecutive points of the whole chain differ in x, a crossing is recorded at the down-sampled column
 *      x = ((min(u, u') side) + .5) / 5 - .5 when that is a whole number inside [0, w - 1], at row
 *      ceil(clamp((min(v, v') + .5) / 5 - .5, 0, h));
 *   4. the crossings, as column-major positions x h + y, are sorted; consecutive differences are the run lengths (first
 *      run = zeros), zero-length runs are merged away (two crossings at one position cancel);
 *   5. rleDecode paints the runs into a column-major h x w image.
 * `mask_cm` [h * w] (column-major, uint8) receives this polygon OR-ed over what it holds (segms.py:117-118 sums the
 * per-polygon masks of an instance and thresholds at > 0). */
static int oracle_cmp_u32(const void* a, const void* b) {
  const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return x > y ? 1 : x < y ? -1 : 0;
}

void oracle_poly_to_mask(const double* xy, int k, int h, int w, uint8_t* mask_cm) {
  if (k <= 0) return;
  const double scale = 5;
  int* px = (int*)malloc(sizeof(int) * (size_t)(k + 1));
  int* py = (int*)malloc(sizeof(int) * (size_t)(k + 1));
  for (int j = 0; j < k; j++) {
    px[j] = (int)(scale * xy[2 * j] + .5);
    py[j] = (int)(scale * xy[2 * j + 1] + .5);
  }
  px[k] = px[0];
  py[k] = py[0];
  size_t total = 0;
  for (int j = 0; j < k; j++) {
    const int ax = abs(px[j] - px[j + 1]), ay = abs(py[j] - py[j + 1]);
    total += (size_t)(ax > ay ? ax : ay) + 1;
  }
  int* u = (int*)malloc(sizeof(int) * total);
  int* v = (int*)malloc(sizeof(int) * total);
  size_t m = 0;
  for (int j = 0; j < k; j++) {
    int xs = px[j], xe = px[j + 1], ys = py[j], ye = py[j + 1];
    const int dx = abs(xe - xs), dy = abs(ys - ye);
    const int flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
    if (flip) {
      int t = xs; xs = xe; xe = t;
      t = ys; ys = ye; ye = t;
    }
    const double slope = dx >= dy ? (double)(ye - ys) / dx : (double)(xe - xs) / dy;
    if (dx >= dy) {
      for (int d = 0; d <= dx; d++) {
        const int t = flip ? dx - d : d;
        u[m] = t + xs;
        v[m] = (int)(ys + slope * t + .5);
        m++;
      }
    } else {
      for (int d = 0; d <= dy; d++) {
        const int t = flip ? dy - d : d;
        v[m] = t + ys;
        u[m] = (int)(xs + slope * t + .5);
        m++;
      }
    }
  }
  uint32_t* pos = (uint32_t*)malloc(sizeof(uint32_t) * (total + 1));
  size_t n = 0;
  for (size_t j = 1; j < m; j++) {
    if (u[j] == u[j - 1]) continue;
    double xd = (double)(u[j] < u[j - 1] ? u[j] : u[j] - 1);
    xd = (xd + .5) / scale - .5;
    if (floor(xd) != xd || xd < 0 || xd > w - 1) continue;
    double yd = (double)(v[j] < v[j - 1] ? v[j] : v[j - 1]);
    yd = (yd + .5) / scale - .5;
    if (yd < 0) yd = 0;
    else if (yd > h) yd = h;
    yd = ceil(yd);
    pos[n++] = (uint32_t)((int)xd * h + (int)yd);
  }
  pos[n++] = (uint32_t)(h * w);
  qsort(pos, n, sizeof(uint32_t), oracle_cmp_u32);
  /* differences = run lengths; a zero difference (other than a leading one) is dropped together with the toggle before it */
  uint32_t* runs = (uint32_t*)malloc(sizeof(uint32_t) * n);
  uint32_t prev = 0;
  for (size_t j = 0; j < n; j++) {
    const uint32_t t = pos[j];
    pos[j] = t - prev;
    prev = t;
  }
  size_t nr = 0, j = 0;
  runs[nr++] = pos[j++];
  while (j < n) {
    if (pos[j] > 0) {
      runs[nr++] = pos[j++];
    } else {
      j++;
      if (j < n) runs[nr - 1] += pos[j++];
    }
  }
  /* rleDecode: runs alternate 0 / 1, column-major */
  size_t at = 0;
  uint8_t value = 0;
  for (size_t r = 0; r < nr; r++) {
    for (uint32_t c = 0; c < runs[r] && at < (size_t)h * w; c++, at++)
      if (value) mask_cm[at] = 1;
    value = !value;
  }
  free(px); free(py); free(u); free(v); free(pos); free(runs);
}

int oracle_num_threads_available(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
