// roi_pool.hip -- RoIPool forward / backward for gfx950, C-ABI mi_roi_pool_*.
//
// Arithmetic contract: lib/model/roi_pooling/src/roi_pooling_kernel.cu:24-93 (forward: max over the bin's pixels, strict >,
// scanned row-major so the first maximum wins, int32 flat argmax over the whole input tensor, empty bin -> 0 / -1) and
// :128-203 (backward).
//
// Forward (round 6): one 256-lane workgroup per (RoI, 32-channel tile), tile = blockIdx % tiles so that one XCD's L2 serves
// one channel slab.  The RoI's rectangle is decoded once per workgroup (wave-uniform, SGPRs) and its bin rows / columns
// tabulated in LDS (the reference's threads each redo the rounding, floor and ceil: 49 x 32 times per workgroup here).  Lanes
// run over (channel, bin) with the bin fastest -- neighbouring lanes scan neighbouring bins of one plane and write neighbouring
// outputs -- and a bin's pixels come straight from memory, FOUR ROWS x EIGHT COLUMNS IN FLIGHT PER WAIT (masked past the bin:
// the sentinel -FLT_MAX never wins a strict >), compared in the reference's row-major order with one compare and two selects
// per pixel (the winner's position in the batch is an instruction constant; its flat index is formed once per batch).
// Bit-exact: the comparisons are the reference's, on the same values in the same order.
// Built first this round and measured against it (profiles/r06_pool_crop.txt; code: commit e830f0d): the RoI's rows staged
// through LDS by LDS-DMA in chunks, lane = channel, (max, argmax) carried across chunks in an LDS tile -- 86 us at the
// config-2 shape and 103 us on a stride-16 map with image-sized RoIs, against 79 and 96 us for this form and 100 and 112 us
// for the one-pixel-per-wait loop of rounds 1-5: a bin's taps are few and its neighbours' overlap is served by L1 / L2, so
// the image, its barriers and the chain of LDS round trips (tile entry, bin bounds, rows) cost more than they save.
//
// Backward: the reference launches one thread per INPUT element and loops over all R RoIs and their candidate bins
// comparing argmax == index: O(N*C*H*W*R).  The same sums are produced here by scattering each output gradient through
// its argmax (one fp32 atomic per output element, O(R*C*PH*PW)); the reference's extra conditions -- the argmax pixel
// must lie inside the rounded RoI rectangle [start, end] (:161-165, false for malformed RoIs whose width was forced to
// 1) -- are re-checked so the set of contributing terms is identical.  Only the floating-point addition order differs
// (reference: ascending RoI index).
#include "common.h"
#include "lds_dma.h"  // uniform()

#include <cfloat>

namespace {

using namespace mi;
using const_float_ptr = const __attribute__((address_space(4))) float*;

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

constexpr int kPoolCT = 32;        // channels per workgroup
constexpr int kPoolThreads = 256;
#ifndef MI_POOL_GATHER_ROWS
#define MI_POOL_GATHER_ROWS 4
#endif
constexpr int kGatherRows = MI_POOL_GATHER_ROWS;  // rows of a bin in flight per wait

// bin p of an axis: [start, end) in map coordinates, clamped to the map (roi_pooling_kernel.cu:57-66)
__device__ __forceinline__ void pool_bin(int p, float bin_size, int roi_start, int size, int& lo, int& hi) {
  lo = (int)floorf((float)p * bin_size);
  hi = (int)ceilf((float)(p + 1) * bin_size);
  lo = (int)fminf(fmaxf((float)(lo + roi_start), 0.f), (float)size);
  hi = (int)fminf(fmaxf((float)(hi + roi_start), 0.f), (float)size);
}

__global__ void __launch_bounds__(kPoolThreads)
roi_pool_fwd(const float* __restrict__ bottom_data, const float* __restrict__ rois, float* __restrict__ top_data,
                    int32_t* __restrict__ argmax_data, int batch, int channels, int height, int width, int pooled_height,
                    int pooled_width, float spatial_scale) {
  extern __shared__ __attribute__((aligned(16))) int tabs[];  // hb[2 PH] | wb[2 PW]
  int* hb = tabs;
  int* wb = tabs + 2 * pooled_height;
  const int tid = threadIdx.x;
  const int tiles = (channels + kPoolCT - 1) / kPoolCT;
  const int r = blockIdx.x / tiles, c0 = (blockIdx.x - r * tiles) * kPoolCT;
  const int bins = pooled_height * pooled_width;
  const const_float_ptr roi = (const_float_ptr)(uintptr_t)(rois + (long long)r * 5);
  const int batch_ind = (int)roi[0];
  const int start_w = (int)roundf(roi[1] * spatial_scale), start_h = (int)roundf(roi[2] * spatial_scale);  // :46-49
  const int end_w = (int)roundf(roi[3] * spatial_scale), end_h = (int)roundf(roi[4] * spatial_scale);
  const int roi_width = (int)fmaxf((float)(end_w - start_w + 1), 1.f);  // :52-53
  const int roi_height = (int)fmaxf((float)(end_h - start_h + 1), 1.f);
  const float bin_size_h = (float)roi_height / (float)pooled_height;  // :54-55
  const float bin_size_w = (float)roi_width / (float)pooled_width;
  const bool no_image = batch_ind < 0 || batch_ind >= batch;
  for (int p = tid; p < pooled_height + pooled_width; p += kPoolThreads) {
    int lo, hi;
    if (p < pooled_height) {
      pool_bin(p, bin_size_h, start_h, height, lo, hi);
      hb[2 * p] = lo;
      hb[2 * p + 1] = hi;
    } else {
      pool_bin(p - pooled_height, bin_size_w, start_w, width, lo, hi);
      wb[2 * (p - pooled_height)] = lo;
      wb[2 * (p - pooled_height) + 1] = hi;
    }
  }
  __syncthreads();
  const int cvalid = min(kPoolCT, channels - c0);
  float* __restrict__ dst = top_data + ((long long)r * channels + c0) * bins;
  int32_t* __restrict__ adst = argmax_data != nullptr ? argmax_data + ((long long)r * channels + c0) * bins : nullptr;
  const unsigned bins_magic = (1u << 20) / (unsigned)bins + 1u;
  for (int i = tid; i < cvalid * bins; i += kPoolThreads) {
    const int c = bins <= 128 ? (int)(((unsigned)i * bins_magic) >> 20) : i / bins, b = i - c * bins;
    const int ph = b / pooled_width, pw = b - ph * pooled_width;
    const int hs = hb[2 * ph], he = hb[2 * ph + 1], ws = wb[2 * pw], we = wb[2 * pw + 1];
    const bool is_empty = he <= hs || we <= ws || no_image;
    float maxval = is_empty ? 0.f : -FLT_MAX;  // :70
    int maxidx = -1;                           // :72
    if (!is_empty) {
      const int off = (batch_ind * channels + c0 + c) * height * width;  // :75-76
      const float* plane = bottom_data + off;
      const int ncol = we - ws;
      if (ncol <= 8) {
        for (int h = hs; h < he; h += kGatherRows) {
          const float* rowp = plane + h * width + ws;
          float v[kGatherRows][8];
#pragma unroll
          for (int k = 0; k < kGatherRows; k++)
#pragma unroll
            for (int j = 0; j < 8; j++) v[k][j] = (h + k < he && j < ncol) ? rowp[k * width + j] : -FLT_MAX;
          int best = -1;
#pragma unroll
          for (int k = 0; k < kGatherRows; k++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const bool up = v[k][j] > maxval;  // :83 strict >, row-major
              maxval = up ? v[k][j] : maxval;
              best = up ? k * 8 + j : best;
            }
          if (best >= 0) maxidx = off + (h + (best >> 3)) * width + ws + (best & 7);
        }
      } else {
        // a wide bin (the stride-16 maps of the C4 configs): row by row, eight columns per wait
        for (int h = hs; h < he; ++h)
          for (int cb = 0; cb < ncol; cb += 8) {
            const float* rowp = plane + h * width + ws + cb;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = cb + j < ncol ? rowp[j] : -FLT_MAX;
            int best = -1;
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const bool up = v[j] > maxval;
              maxval = up ? v[j] : maxval;
              best = up ? j : best;
            }
            if (best >= 0) maxidx = off + h * width + ws + cb + best;
          }
      }
    }
    dst[i] = maxval;
    if (adst != nullptr) adst[i] = maxidx;
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
  MI_REQUIRE(pooled_height + pooled_width <= 4096, "roi_pool: pooled size %d x %d beyond the bin table", pooled_height, pooled_width);
  const int tiles = (channels + kPoolCT - 1) / kPoolCT;
  roi_pool_fwd<<<num_rois * tiles, kPoolThreads, (size_t)2 * (pooled_height + pooled_width) * 4, mi::as_stream(stream)>>>(
      features, rois, output, argmax, batch, channels, height, width, pooled_height, pooled_width, spatial_scale);
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
