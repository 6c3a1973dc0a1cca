// roi_crop.hip -- RoICrop (bilinear grid sampler) forward / backward for gfx950, C-ABI mi_roi_crop_*.
//
// Arithmetic contract: lib/model/roi_crop/src/roi_crop_cuda_kernel.cu:11-27 (getTopLeft, between),
// :47-109 (bilinearSamplingFromGrid), :111-194 (backwardBilinearSampling), for the dense NCHW
// input / [R,GH,GW,2] (y,x) grid / [R,C,GH,GW] output strides that roi_crop_cuda.c:23-44 passes.
// Quirks kept: output elements whose four taps are all outside the image are not written
// (:92-93, caller zero-fills); image of RoI r is r / (R / N) (:64,:217); the backward never
// writes a grid gradient.
//
// Mapping: one lane per (roi, y, x) grid point x channel; the grid coordinates and the four
// weights are per-(roi,y,x) quantities, so lanes iterate channels in the outer grid-stride
// dimension and consecutive lanes take consecutive x (coalesced output rows).
#include "common.h"

namespace {

__device__ __forceinline__ void get_top_left(float x, int width, int& point, float& weight) {
  float xcoord = (x + 1) * (width - 1) / 2;  // :19
  point = (int)floorf(xcoord);               // :20
  weight = 1 - (xcoord - point);             // :21
}
__device__ __forceinline__ bool between(int value, int lo, int hi) { return value >= lo && value <= hi; }

struct CropTaps {
  long long tl;
  bool tl_in, tr_in, bl_in, br_in;
  float xw, yw;
};

__device__ __forceinline__ CropTaps crop_taps(const float* __restrict__ grids, long long gidx, int b,
                                              int cOut, int channels, int height, int width,
                                              int roiPerImage) {
  CropTaps t;
  const int b_input = b / roiPerImage;  // :64
  float yf = grids[gidx * 2];           // :66
  float xf = grids[gidx * 2 + 1];       // :67
  int yTL, xTL;
  get_top_left(xf, width, xTL, t.xw);
  get_top_left(yf, height, yTL, t.yw);
  t.tl = ((long long)(b_input * channels + cOut) * height + yTL) * width + xTL;
  t.tl_in = between(xTL, 0, width - 1) && between(yTL, 0, height - 1);
  t.tr_in = between(xTL + 1, 0, width - 1) && between(yTL, 0, height - 1);
  t.bl_in = between(xTL, 0, width - 1) && between(yTL + 1, 0, height - 1);
  t.br_in = between(xTL + 1, 0, width - 1) && between(yTL + 1, 0, height - 1);
  return t;
}

__global__ void __launch_bounds__(256)
roi_crop_fwd(long long total, const float* __restrict__ input, const float* __restrict__ grids,
             float* __restrict__ output, int channels, int height, int width, int gh, int gw,
             int roiPerImage) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
       index += (long long)gridDim.x * blockDim.x) {
    const int xOut = (int)(index % gw);
    const int yOut = (int)((index / gw) % gh);
    const int cOut = (int)((index / gw / gh) % channels);
    const int b = (int)(index / gw / gh / channels);
    CropTaps t = crop_taps(grids, ((long long)b * gh + yOut) * gw + xOut, b, cOut, channels, height,
                           width, roiPerImage);
    if (!t.tl_in && !t.tr_in && !t.bl_in && !t.br_in) continue;  // :92-93
    float inTopLeft = 0, inTopRight = 0, inBottomLeft = 0, inBottomRight = 0;
    if (t.tl_in) inTopLeft = input[t.tl];
    if (t.tr_in) inTopRight = input[t.tl + 1];
    if (t.bl_in) inBottomLeft = input[t.tl + width];
    if (t.br_in) inBottomRight = input[t.tl + width + 1];
    float v = t.xw * t.yw * inTopLeft  // :100-103
              + (1 - t.xw) * t.yw * inTopRight + t.xw * (1 - t.yw) * inBottomLeft +
              (1 - t.xw) * (1 - t.yw) * inBottomRight;
    output[index] = v;
  }
}

__global__ void __launch_bounds__(256)
roi_crop_bwd(long long total, const float* __restrict__ grids, const float* __restrict__ grad_output,
             float* __restrict__ grad_input, int channels, int height, int width, int gh, int gw,
             int roiPerImage) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
       index += (long long)gridDim.x * blockDim.x) {
    const int xOut = (int)(index % gw);
    const int yOut = (int)((index / gw) % gh);
    const int cOut = (int)((index / gw / gh) % channels);
    const int b = (int)(index / gw / gh / channels);
    CropTaps t = crop_taps(grids, ((long long)b * gh + yOut) * gw + xOut, b, cOut, channels, height,
                           width, roiPerImage);
    float gradOutValue = grad_output[index];
    if (t.tl_in) atomicAdd(grad_input + t.tl, t.xw * t.yw * gradOutValue);  // :169
    if (t.tr_in) atomicAdd(grad_input + t.tl + 1, (1 - t.xw) * t.yw * gradOutValue);
    if (t.bl_in) atomicAdd(grad_input + t.tl + width, t.xw * (1 - t.yw) * gradOutValue);
    if (t.br_in) atomicAdd(grad_input + t.tl + width + 1, (1 - t.xw) * (1 - t.yw) * gradOutValue);
  }
}

int check_crop(const void* a, const void* grid, const void* b, int batch, int channels, int height,
               int width, int num_rois, int gh, int gw) {
  MI_REQUIRE(batch > 0 && channels >= 0 && height > 0 && width > 0 && num_rois >= 0 && gh > 0 &&
                 gw > 0,
             "roi_crop: bad size");
  MI_REQUIRE(num_rois == 0 || num_rois / batch > 0,
             "roi_crop: fewer RoIs (%d) than images (%d): RoIs-per-image would be 0 (reference divides by it)",
             num_rois, batch);
  if ((long long)num_rois * channels > 0)
    MI_REQUIRE(a != nullptr && grid != nullptr && b != nullptr, "roi_crop: null pointer");
  return MI_OK;
}

}  // namespace

extern "C" int mi_roi_crop_forward(const float* input, const float* grid_yx, float* output,
                                   int batch, int channels, int height, int width, int num_rois,
                                   int grid_height, int grid_width, mi_stream_t stream) {
  mi::begin_call();
  int rc = check_crop(input, grid_yx, output, batch, channels, height, width, num_rois, grid_height,
                      grid_width);
  if (rc != MI_OK) return rc;
  const long long total = (long long)num_rois * channels * grid_height * grid_width;
  if (total == 0) return MI_OK;
  const int block = 256;
  roi_crop_fwd<<<mi::grid_for(total, block), block, 0, mi::as_stream(stream)>>>(
      total, input, grid_yx, output, channels, height, width, grid_height, grid_width,
      num_rois / batch);
  return mi::check_launch("roi_crop_fwd");
}

extern "C" int mi_roi_crop_backward(const float* input, const float* grid_yx,
                                    const float* grad_output, float* grad_input, int batch,
                                    int channels, int height, int width, int num_rois,
                                    int grid_height, int grid_width, mi_stream_t stream) {
  mi::begin_call();
  (void)input;  // the reference reads it only for the grid gradient it then discards (:166-190)
  int rc = check_crop(grad_output, grid_yx, grad_input, batch, channels, height, width, num_rois,
                      grid_height, grid_width);
  if (rc != MI_OK) return rc;
  const long long total = (long long)num_rois * channels * grid_height * grid_width;
  if (total == 0) return MI_OK;
  const int block = 256;
  roi_crop_bwd<<<mi::grid_for(total, block), block, 0, mi::as_stream(stream)>>>(
      total, grid_yx, grad_output, grad_input, channels, height, width, grid_height, grid_width,
      num_rois / batch);
  return mi::check_launch("roi_crop_bwd");
}
