// roi_crop.hip -- RoICrop (bilinear grid sampler) forward / backward for gfx950, C-ABI mi_roi_crop_*.
//
// Arithmetic contract: lib/model/roi_crop/src/roi_crop_cuda_kernel.cu:11-27 (getTopLeft, between),
// :47-109 (bilinearSamplingFromGrid), :111-194 (backwardBilinearSampling), for the dense NCHW
// input / [R,GH,GW,2] (y,x) grid / [R,C,GH,GW] output strides that roi_crop_cuda.c:23-44 passes.
// Quirks kept: output elements whose four taps are all outside the image are not written
// (:92-93, caller zero-fills); image of RoI r is r / (R / N) (:64,:217); the backward never
// writes a grid gradient.
//
// Round 6 (until then the reference's one-thread-per-output-element loops re-typed): one 256-lane workgroup per (RoI,
// 32-channel tile), tile = blockIdx % tiles (one XCD's L2 serves one channel slab).  A grid point's top-left tap, its four
// weights and which of its taps lie in the image do not depend on the channel: they are computed ONCE per workgroup --
// 64 points at a time, one per lane of wave 0 -- into an LDS table (the reference recomputes them in every thread: 32 x
// per point here), with the bounding box of the group's in-image taps reduced across the wave (an empty box: skip).
//   forward   lanes flattened over (channel, point) with the point fastest: the four taps straight from memory (a sampler's
//             taps are sparse in its box -- 4 x 49 of ~1000 pixels at 7 x 7: staging the box through LDS, built first this
//             round, moved five times the bytes the taps need and measured 62 us against 53 at the config-2 shape), the
//             reference's expression with the table's products, neighbouring lanes on neighbouring outputs; points
//             without a tap in the image are left unwritten.
//   backward  lanes flattened over (channel, point) with the point fastest -- the gradient block is read as it lies in
//             memory -- and the reference's four atomics per element with the table's weights.  (A sampler's taps are
//             sparse in its box -- 4 x 49 of ~1000 pixels at 7 x 7 -- so accumulating the box in LDS and flushing it would
//             issue MORE atomics than the taps themselves; the scatter stays a scatter.)
// Bit-exact forward: the same fp32 products and sums in the reference's order (-ffp-contract=off).
#include "common.h"
#include "lds_dma.h"  // uniform()

namespace {

using namespace mi;

constexpr int kCropCT = 32;        // channels per workgroup
constexpr int kCropThreads = 256;
constexpr int kCropPts = 64;       // grid points per group: one per lane of the wave that builds the table

// roi_crop_cuda_kernel.cu:17-23
__device__ __forceinline__ void get_top_left(float x, int width, int& point, float& weight) {
  float xcoord = (x + 1) * (width - 1) / 2;
  point = (int)floorf(xcoord);
  weight = 1 - (xcoord - point);
}
__device__ __forceinline__ bool between(int value, int lo, int hi) { return value >= lo && value <= hi; }

// What is identical for all channels of a grid point.
struct PointTab {
  int x[kCropPts], y[kCropPts];          // top-left tap
  float w_tl[kCropPts], w_tr[kCropPts], w_bl[kCropPts], w_br[kCropPts];  // xw*yw, (1-xw)*yw, xw*(1-yw), (1-xw)*(1-yw)
  int in[kCropPts];                      // bit 0..3: top-left, top-right, bottom-left, bottom-right tap lies in the image
  int box[4];                            // rows [y0, y1], columns [x0, x1] of the taps of the group that lie in the image
};

// wave 0: the table of points [p0, p0 + np) of RoI r and the bounding box of their in-image taps
__device__ __forceinline__ void crop_build_table(PointTab* tab, const float* __restrict__ grids, long long gbase, int np,
                                                 int lane, int height, int width, bool image_ok) {
  int xTL = 0, yTL = 0, in = 0;
  float xw = 0.f, yw = 0.f;
  if (lane < np) {
    const float yf = grids[(gbase + lane) * 2];      // :66
    const float xf = grids[(gbase + lane) * 2 + 1];  // :67
    get_top_left(xf, width, xTL, xw);
    get_top_left(yf, height, yTL, yw);
    const bool xl = between(xTL, 0, width - 1), xr = between(xTL + 1, 0, width - 1);
    const bool yt = between(yTL, 0, height - 1), yb = between(yTL + 1, 0, height - 1);
    in = image_ok ? ((xl && yt) ? 1 : 0) | ((xr && yt) ? 2 : 0) | ((xl && yb) ? 4 : 0) | ((xr && yb) ? 8 : 0) : 0;
  }
  tab->x[lane] = xTL;
  tab->y[lane] = yTL;
  tab->w_tl[lane] = xw * yw;              // the products of :100-103 / :169-172, formed as the reference forms them
  tab->w_tr[lane] = (1 - xw) * yw;
  tab->w_bl[lane] = xw * (1 - yw);
  tab->w_br[lane] = (1 - xw) * (1 - yw);
  tab->in[lane] = in;
  // rows / columns of the in-image taps: a point with a tap in the image has its rows in [max(y, 0), min(y + 1, H - 1)]
  int ylo = in ? max(yTL, 0) : 0x3fffffff, yhi = in ? min(yTL + 1, height - 1) : -1;
  int xlo = in ? max(xTL, 0) : 0x3fffffff, xhi = in ? min(xTL + 1, width - 1) : -1;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    ylo = min(ylo, __shfl_xor(ylo, d));
    yhi = max(yhi, __shfl_xor(yhi, d));
    xlo = min(xlo, __shfl_xor(xlo, d));
    xhi = max(xhi, __shfl_xor(xhi, d));
  }
  if (lane == 0) {
    tab->box[0] = ylo;
    tab->box[1] = yhi;
    tab->box[2] = xlo;
    tab->box[3] = xhi;
  }
}

__global__ void __launch_bounds__(kCropThreads)
roi_crop_fwd(const float* __restrict__ input, const float* __restrict__ grids, float* __restrict__ output, int batch,
             int channels, int height, int width, int gh, int gw, int roiPerImage) {
  __shared__ PointTab tab;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int tiles = (channels + kCropCT - 1) / kCropCT;
  const int r = blockIdx.x / tiles, c0 = (blockIdx.x - r * tiles) * kCropCT;
  const int points = gh * gw;
  const int b_input = r / roiPerImage;  // :64
  const bool image_ok = b_input < batch;  // (the reference would read past its input)
  const int cvalid = min(kCropCT, channels - c0);
  const long long plane_px = (long long)height * width;
  const float* __restrict__ src = input + ((long long)(image_ok ? b_input : 0) * channels + c0) * plane_px;
  float* __restrict__ dst = output + ((long long)r * channels + c0) * points;
  for (int p0 = 0; p0 < points; p0 += kCropPts) {
    const int np = min(kCropPts, points - p0);
    __syncthreads();  // the previous group's table is no longer read
    if (wave == 0) crop_build_table(&tab, grids, (long long)r * points + p0, np, lane, height, width, image_ok);
    __syncthreads();
    if (uniform(tab.box[1]) < uniform(tab.box[0])) continue;  // no tap of the group lies in the image: nothing is written
    // lanes over (channel, point), the point fastest: neighbouring lanes tap neighbouring pixels of one plane and write
    // neighbouring outputs
    const unsigned np_magic = (1u << 20) / (unsigned)np + 1u;
    for (int i = tid; i < cvalid * np; i += kCropThreads) {
      const int c = (int)(((unsigned)i * np_magic) >> 20), p = i - c * np;  // i / np, exact for i < 32 * 64
      const int in = tab.in[p];
      if (in == 0) continue;  // :92-93: not written
      const float* a = src + (long long)c * plane_px + (long long)tab.y[p] * width + tab.x[p];
      float inTopLeft = 0, inTopRight = 0, inBottomLeft = 0, inBottomRight = 0;
      if (in & 1) inTopLeft = a[0];
      if (in & 2) inTopRight = a[1];
      if (in & 4) inBottomLeft = a[width];
      if (in & 8) inBottomRight = a[width + 1];
      // :100-103
      dst[(long long)c * points + p0 + p] = tab.w_tl[p] * inTopLeft + tab.w_tr[p] * inTopRight + tab.w_bl[p] * inBottomLeft +
                                           tab.w_br[p] * inBottomRight;
    }
  }
}

__global__ void __launch_bounds__(kCropThreads)
roi_crop_bwd(const float* __restrict__ grids, const float* __restrict__ grad_output, float* __restrict__ grad_input,
             int batch, int channels, int height, int width, int gh, int gw, int roiPerImage) {
  __shared__ PointTab tab;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int tiles = (channels + kCropCT - 1) / kCropCT;
  const int r = blockIdx.x / tiles, c0 = (blockIdx.x - r * tiles) * kCropCT;
  const int points = gh * gw;
  const int b_input = r / roiPerImage;  // :128
  const bool image_ok = b_input < batch;
  const int cvalid = min(kCropCT, channels - c0);
  const long long plane_px = (long long)height * width;
  float* __restrict__ dst = grad_input + ((long long)(image_ok ? b_input : 0) * channels + c0) * plane_px;
  const float* __restrict__ gsrc = grad_output + ((long long)r * channels + c0) * points;
  for (int p0 = 0; p0 < points; p0 += kCropPts) {
    const int np = min(kCropPts, points - p0);
    __syncthreads();
    if (wave == 0) crop_build_table(&tab, grids, (long long)r * points + p0, np, lane, height, width, image_ok);
    __syncthreads();
    if (uniform(tab.box[1]) < uniform(tab.box[0])) continue;  // no tap of the group lies in the image
    for (int i = tid; i < cvalid * np; i += kCropThreads) {
      const int c = i / np, p = i - c * np;
      const int in = tab.in[p];
      if (in == 0) continue;
      const float gradOutValue = gsrc[(long long)c * points + p0 + p];
      float* a = dst + (long long)c * plane_px + (long long)tab.y[p] * width + tab.x[p];
      if (in & 1) atomicAdd(a, tab.w_tl[p] * gradOutValue);  // :169-172
      if (in & 2) atomicAdd(a + 1, tab.w_tr[p] * gradOutValue);
      if (in & 4) atomicAdd(a + width, tab.w_bl[p] * gradOutValue);
      if (in & 8) atomicAdd(a + width + 1, tab.w_br[p] * gradOutValue);
    }
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
  MI_REQUIRE((long long)height * width * 4 * kCropCT < (1LL << 31), "roi_crop: a 32-channel slab of the map exceeds 2 GB");
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
  const int tiles = (channels + kCropCT - 1) / kCropCT;
  roi_crop_fwd<<<num_rois * tiles, kCropThreads, 0, mi::as_stream(stream)>>>(
      input, grid_yx, output, batch, channels, height, width, grid_height, grid_width, num_rois / batch);
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
  const int tiles = (channels + kCropCT - 1) / kCropCT;
  roi_crop_bwd<<<num_rois * tiles, kCropThreads, 0, mi::as_stream(stream)>>>(
      grid_yx, grad_output, grad_input, batch, channels, height, width, grid_height, grid_width, num_rois / batch);
  return mi::check_launch("roi_crop_bwd");
}
