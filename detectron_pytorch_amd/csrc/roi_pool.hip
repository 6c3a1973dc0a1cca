// roi_pool.hip -- RoIPool forward / backward for gfx950, C-ABI mi_roi_pool_*.
//
// Arithmetic contract: lib/model/roi_pooling/src/roi_pooling_kernel.cu:24-93 (forward, max with
// first-max-wins row-major scan and int32 flat argmax) and :128-203 (backward).
//
// The reference backward launches one thread per INPUT element and loops over all R RoIs and
// their candidate bins comparing argmax == index: O(N*C*H*W*R).  The same sums are produced
// here by scattering each output gradient through its argmax (one fp32 atomic per output
// element, O(R*C*PH*PW)); the reference's extra conditions -- the argmax pixel must lie inside
// the rounded RoI rectangle [start, end] (:161-165, false for malformed RoIs whose width was
// forced to 1) -- are re-checked so the set of contributing terms is identical.  Only the
// floating-point addition order differs (reference: ascending RoI index).
#include "common.h"

#include <cfloat>

namespace {

struct PoolRoi {
  int batch_ind, start_w, start_h, end_w, end_h;
};

__device__ __forceinline__ PoolRoi pool_roi(const float* __restrict__ roi, float spatial_scale) {
  PoolRoi r;
  r.batch_ind = (int)roi[0];
  r.start_w = (int)roundf(roi[1] * spatial_scale);  // :46-49 round half away from zero
  r.start_h = (int)roundf(roi[2] * spatial_scale);
  r.end_w = (int)roundf(roi[3] * spatial_scale);
  r.end_h = (int)roundf(roi[4] * spatial_scale);
  return r;
}

__global__ void __launch_bounds__(256)
roi_pool_fwd(long long total, const float* __restrict__ bottom_data,
             const float* __restrict__ rois, float* __restrict__ top_data,
             int32_t* __restrict__ argmax_data, int batch, int channels, int height, int width,
             int pooled_height, int pooled_width, float spatial_scale) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
       index += (long long)gridDim.x * blockDim.x) {
    int pw = (int)(index % pooled_width);
    int ph = (int)((index / pooled_width) % pooled_height);
    int c = (int)((index / pooled_width / pooled_height) % channels);
    int n = (int)(index / pooled_width / pooled_height / channels);
    PoolRoi r = pool_roi(rois + (long long)n * 5, spatial_scale);
    int roi_width = (int)fmaxf((float)(r.end_w - r.start_w + 1), 1.f);  // :52-53
    int roi_height = (int)fmaxf((float)(r.end_h - r.start_h + 1), 1.f);
    float bin_size_h = (float)(roi_height) / (float)(pooled_height);  // :54-55
    float bin_size_w = (float)(roi_width) / (float)(pooled_width);
    int hstart = (int)(floorf((float)(ph)*bin_size_h));  // :57-60
    int wstart = (int)(floorf((float)(pw)*bin_size_w));
    int hend = (int)(ceilf((float)(ph + 1) * bin_size_h));
    int wend = (int)(ceilf((float)(pw + 1) * bin_size_w));
    hstart = (int)fminf(fmaxf((float)(hstart + r.start_h), 0.f), (float)height);  // :63-66
    hend = (int)fminf(fmaxf((float)(hend + r.start_h), 0.f), (float)height);
    wstart = (int)fminf(fmaxf((float)(wstart + r.start_w), 0.f), (float)width);
    wend = (int)fminf(fmaxf((float)(wend + r.start_w), 0.f), (float)width);
    bool is_empty = (hend <= hstart) || (wend <= wstart) || r.batch_ind < 0 || r.batch_ind >= batch;
    float maxval = is_empty ? 0.f : -FLT_MAX;  // :70
    int maxidx = -1;                           // :72
    if (!is_empty) {
      int bottom_data_offset = (r.batch_ind * channels + c) * height * width;  // :75-76
      for (int h = hstart; h < hend; ++h)
        for (int w = wstart; w < wend; ++w) {
          int bottom_index = h * width + w;
          float v = bottom_data[bottom_data_offset + bottom_index];
          if (v > maxval) {  // :83 strict >, first max in row-major order wins
            maxval = v;
            maxidx = bottom_data_offset + bottom_index;
          }
        }
    }
    top_data[index] = maxval;
    if (argmax_data != nullptr) argmax_data[index] = maxidx;
  }
}

__global__ void __launch_bounds__(256)
roi_pool_bwd(long long total, const float* __restrict__ top_diff, const float* __restrict__ rois,
             const int32_t* __restrict__ argmax_data, float* __restrict__ bottom_diff, int batch,
             int channels, int height, int width, int pooled_height, int pooled_width,
             float spatial_scale) {
  const long long limit = (long long)batch * channels * height * width;
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
       index += (long long)gridDim.x * blockDim.x) {
    int n = (int)(index / pooled_width / pooled_height / channels);
    int am = argmax_data[index];
    if (am < 0 || am >= limit) continue;
    PoolRoi r = pool_roi(rois + (long long)n * 5, spatial_scale);
    int w = am % width;
    int h = (am / width) % height;
    int img = am / width / height / channels;
    if (img != r.batch_ind) continue;  // :151-153
    const bool in_roi = (w >= r.start_w && w <= r.end_w && h >= r.start_h && h <= r.end_h);  // :161-165
    if (!in_roi) continue;
    atomicAdd(bottom_diff + am, top_diff[index]);
  }
}

int check_pool(const void* a, const void* rois, const void* b, int batch, int channels, int height,
               int width, int num_rois, int ph, int pw) {
  MI_REQUIRE(batch >= 0 && channels >= 0 && height >= 0 && width >= 0 && num_rois >= 0,
             "roi_pool: negative size");
  MI_REQUIRE(ph > 0 && pw > 0, "roi_pool: pooled size must be positive");
  MI_REQUIRE((long long)batch * channels * height * width < (1LL << 31),
             "roi_pool: feature tensor too large for the int32 argmax of the reference ABI");
  if ((long long)num_rois * channels > 0)
    MI_REQUIRE(a != nullptr && rois != nullptr && b != nullptr, "roi_pool: null pointer");
  return MI_OK;
}

}  // namespace

extern "C" int mi_roi_pool_forward(const float* features, const float* rois, float* output,
                                   int32_t* argmax, int batch, int channels, int height, int width,
                                   int num_rois, int pooled_height, int pooled_width,
                                   float spatial_scale, mi_stream_t stream) {
  mi::begin_call();
  int rc = check_pool(features, rois, output, batch, channels, height, width, num_rois,
                      pooled_height, pooled_width);
  if (rc != MI_OK) return rc;
  const long long total = (long long)num_rois * channels * pooled_height * pooled_width;
  if (total == 0) return MI_OK;
  const int block = 256;
  roi_pool_fwd<<<mi::grid_for(total, block), block, 0, mi::as_stream(stream)>>>(
      total, features, rois, output, argmax, batch, channels, height, width, pooled_height,
      pooled_width, spatial_scale);
  return mi::check_launch("roi_pool_fwd");
}

extern "C" int mi_roi_pool_backward(const float* top_grad, const float* rois,
                                    const int32_t* argmax, float* bottom_grad, int batch,
                                    int channels, int height, int width, int num_rois,
                                    int pooled_height, int pooled_width, float spatial_scale,
                                    mi_stream_t stream) {
  mi::begin_call();
  int rc = check_pool(top_grad, rois, bottom_grad, batch, channels, height, width, num_rois,
                      pooled_height, pooled_width);
  if (rc != MI_OK) return rc;
  MI_REQUIRE(argmax != nullptr || (long long)num_rois * channels == 0, "roi_pool: null argmax");
  const long long total = (long long)num_rois * channels * pooled_height * pooled_width;
  if (total == 0) return MI_OK;
  const int block = 256;
  roi_pool_bwd<<<mi::grid_for(total, block), block, 0, mi::as_stream(stream)>>>(
      total, top_grad, rois, argmax, bottom_grad, batch, channels, height, width, pooled_height,
      pooled_width, spatial_scale);
  return mi::check_launch("roi_pool_bwd");
}
