// results.hip -- the test-time result formats on the device, C-ABI mi_mask_paste_rle and
// mi_keypoint_decode.
//
//   mi_mask_paste_rle    one detection of segm_results (lib/core/test.py:793-847): the M x M soft mask, zero-padded by one
//       pixel, resized to the (expanded, truncated) reference box with cv2.resize's bilinear arithmetic, binarised, pasted
//       into an im_h x im_w image and run-length encoded the way pycocotools does (column-major runs, zeros first).  The
//       image is never materialised: a workgroup walks the box columns, evaluates the pasted bit of a pixel and of its
//       predecessor in column-major order directly from the M x M mask in LDS, and emits the positions where they differ
//       (ordered ballot compaction); run lengths are the differences of consecutive positions.  The reference resizes and
//       pastes on the host with OpenCV and encodes with pycocotools, one detection at a time.
//   mi_keypoint_decode   heatmaps_to_keypoints (lib/utils/keypoints.py:106-157): every heat map resized to the RoI with
//       cv2.resize's bicubic arithmetic, arg-max (first maximum in row-major order), its logit and its softmax probability
//       over the resized map (scores_to_probs, :214-222).  One workgroup per (RoI, keypoint); the resized map is never
//       stored: each lane evaluates its pixels from the 56 x 56 map in LDS and keeps (max, arg-max, running sum of exp).
//
// Arithmetic: OpenCV's scalar code paths (modules/imgproc/src/resize.cpp: resizeGeneric_, HResizeLinear / VResizeLinear,
// interpolateCubic, HResizeCubic / VResizeCubic), fp32 products and sums in their order, no FMA contraction (this library
// is compiled with -ffp-contract=off); coordinates through double exactly as there.  OpenCV and pycocotools are absent
// from the build environment: the CPU restatement these kernels are tested against (oracle/results.py) is unpinned.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxMask = 64;   // M + 2 <= 64 (the reference uses M = 28; 14 for the light-weight heads)

// source index and weight of one destination coordinate of a bilinear axis (resizeGeneric_, ksize 2)
struct Lin {
  int s;
  float w0, w1;
};
__device__ __forceinline__ Lin lin_axis_x(int d, double scale, int src) {   // horizontal: weights reset at the borders
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) {
    f = 0.f;
    s = 0;
  }
  if (s >= src - 1) {
    f = 0.f;
    s = src - 1;
  }
  return Lin{s, 1.f - f, f};
}
__device__ __forceinline__ Lin lin_axis_y(int d, double scale) {            // vertical: rows are clipped, weights kept
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  const int s = (int)floorf(f);
  f -= (float)s;
  return Lin{s, 1.f - f, f};
}

struct PasteGeom {
  int bx0, by0, w, h;            // expanded integer box: origin, size
  int x_0, x_1, y_0, y_1;        // its intersection with the image (columns / rows pasted)
  double scale_x, scale_y;
  int m2;                        // M + 2
  float thresh;
};

// the pasted bit of image pixel (x, y); s_mask: padded [m2][m2] in LDS
__device__ __forceinline__ int pasted_bit(const PasteGeom& g, const float* s_mask, int x, int y) {
  if (x < g.x_0 || x >= g.x_1 || y < g.y_0 || y >= g.y_1) return 0;
  const Lin ax = lin_axis_x(x - g.bx0, g.scale_x, g.m2);
  const Lin ay = lin_axis_y(y - g.by0, g.scale_y);
  const int sx1 = min(ax.s + 1, g.m2 - 1);
  const int r0 = min(max(ay.s, 0), g.m2 - 1), r1 = min(max(ay.s + 1, 0), g.m2 - 1);
  const float h0 = s_mask[r0 * g.m2 + ax.s] * ax.w0 + s_mask[r0 * g.m2 + sx1] * ax.w1;   // HResizeLinear
  const float h1 = s_mask[r1 * g.m2 + ax.s] * ax.w0 + s_mask[r1 * g.m2 + sx1] * ax.w1;
  const float v = h0 * ay.w0 + h1 * ay.w1;                                               // VResizeLinear
  return v > g.thresh ? 1 : 0;
}

__global__ void __launch_bounds__(kThreads)
mask_paste_rle(const float* __restrict__ masks, const int* __restrict__ boxes, int mask_size, int im_h, int im_w,
               float thresh, int cap, unsigned* __restrict__ counts, int* __restrict__ num_counts, int str_cap,
               unsigned char* __restrict__ strings, int* __restrict__ num_bytes) {
  __shared__ float s_mask[kMaxMask * kMaxMask];
  __shared__ int s_wave[kThreads / 64];
  __shared__ int s_total;
  const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m2 = mask_size + 2;
  for (int i = tid; i < m2 * m2; i += kThreads) {
    const int r = i / m2, c = i - r * m2;
    s_mask[i] = (r == 0 || c == 0 || r == m2 - 1 || c == m2 - 1)
                    ? 0.f
                    : masks[((long long)d * mask_size + (r - 1)) * mask_size + (c - 1)];
  }
  if (tid == 0) s_total = 0;
  const int* b = boxes + (long long)d * 4;
  PasteGeom g;
  g.bx0 = b[0];
  g.by0 = b[1];
  g.w = max(b[2] - b[0] + 1, 1);       // lib/core/test.py:817-820
  g.h = max(b[3] - b[1] + 1, 1);
  g.x_0 = max(b[0], 0);                // :826-829
  g.x_1 = min(b[2] + 1, im_w);
  g.y_0 = max(b[1], 0);
  g.y_1 = min(b[3] + 1, im_h);
  g.scale_x = 1.0 / ((double)g.w / (double)m2);
  g.scale_y = 1.0 / ((double)g.h / (double)m2);
  g.m2 = m2;
  g.thresh = thresh;
  __syncthreads();
  unsigned* out = counts + (long long)d * cap;
  const long long total = (long long)im_h * im_w;
  const int rh = g.y_1 - g.y_0, ncols = g.x_1 - g.x_0;
  if (rh > 0 && ncols > 0) {
    // candidates of a column: its rh pasted pixels and the pixel right after them (where a run of ones must end), unless
    // that pixel is itself a pasted pixel of the next column (full-height paste) or lies behind the image
    const long long per_col = rh + 1, ncand = per_col * ncols;
    for (long long base = 0; base < ncand; base += kThreads) {
      const long long t = base + tid;
      bool change = false, real = false;
      long long gpos = 0;
      int yy = 0, xx = 0, y = 0, cur = 0;
      if (t < ncand) {
        const int ci = (int)(t / per_col);
        yy = (int)(t - (long long)ci * per_col);
        xx = g.x_0 + ci;
        y = g.y_0 + yy;
        real = true;
        if (yy == rh && y == im_h) {          // the pixel after a full-height column is the first pixel of the next column
          y = 0;
          xx += 1;
          real = xx < im_w && !(g.y_0 == 0 && xx < g.x_1);
        }
        if (real) {
          gpos = (long long)xx * im_h + y;
          cur = pasted_bit(g, s_mask, xx, y);
        }
      }
      // the column-major predecessor of a candidate with yy >= 1 is the candidate before it (a pasted pixel of the same
      // column): its bit sits in the neighbouring lane
      const int from_left = __shfl_up(cur, 1, 64);
      if (real) {
        int prev = 0;
        if (yy >= 1 && lane > 0) {
          prev = from_left;
        } else if (gpos > 0) {
          const int py = y > 0 ? y - 1 : im_h - 1, px = y > 0 ? xx : xx - 1;
          prev = pasted_bit(g, s_mask, px, py);
        }
        change = cur != prev;
      }
      const unsigned long long mm = __ballot(change);
      if (lane == 0) s_wave[wave] = __popcll(mm);
      __syncthreads();
      int off = s_total;
      for (int w = 0; w < wave; w++) off += s_wave[w];
      if (change) {
        const int k = off + __popcll(mm & ((1ull << lane) - 1ull));
        if (k < cap) out[k] = (unsigned)gpos;            // positions first; run lengths below
      }
      __syncthreads();
      if (tid == 0) {
        int add = 0;
        for (int w = 0; w < kThreads / 64; w++) add += s_wave[w];
        s_total += add;
      }
      __syncthreads();
    }
  }
  // positions p_0 < p_1 < ... -> run lengths: p_0, p_1 - p_0, ..., total - p_last  (rleEncode; p_0 == 0: first run empty)
  const int npos = s_total;
  if (tid == 0) num_counts[d] = npos + 1;
  if (npos + 1 > cap) {                                   // the caller sees the size it needs and calls again
    if (tid == 0 && num_bytes != nullptr) num_bytes[d] = 0x7fffffff;
    return;
  }
  __threadfence_block();
  __syncthreads();
  // in place, chunk by chunk from the TOP: a chunk reads the last position of the chunk below it, which must still be a
  // position; inside a chunk every read precedes every write
  for (int base = (npos / kThreads) * kThreads; base >= 0; base -= kThreads) {
    const int k = base + tid;
    unsigned lo = 0, hi = 0;
    const bool live = k <= npos;
    if (live) {
      lo = k > 0 ? out[k - 1] : 0u;
      hi = k < npos ? out[k] : (unsigned)total;
    }
    __syncthreads();
    if (live) out[k] = hi - lo;
    __syncthreads();
  }
  if (strings == nullptr) return;
  // ---- maskApi.c rleToString: run k is written as the difference to run k - 2 (from the fourth run on) in 5-bit groups,
  // least significant first, bit 5 = "more groups follow", + 48.  Byte offsets: ordered block scan of the group counts.
  __threadfence_block();
  __syncthreads();
  if (tid == 0) s_total = 0;
  __syncthreads();
  unsigned char* str = strings + (long long)d * str_cap;
  for (int base = 0; base <= npos; base += kThreads) {
    const int k = base + tid;
    unsigned char ch[8];
    int n = 0;
    if (k <= npos) {
      long long x = (long long)out[k] - (k > 2 ? (long long)out[k - 2] : 0LL);
      bool more = true;
      while (more && n < 8) {
        int c = (int)(x & 0x1f);
        x >>= 5;                                            // arithmetic shift of a signed long
        more = (c & 0x10) ? x != -1 : x != 0;
        if (more) c |= 0x20;
        ch[n++] = (unsigned char)(c + 48);
      }
    }
    int incl = n;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const int up = __shfl_up(incl, dd, 64);
      if (lane >= dd) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int off = s_total + incl - n;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    for (int j = 0; j < n; j++)
      if (off + j < str_cap) str[off + j] = ch[j];
    __syncthreads();
    if (tid == 0) {
      int add = 0;
      for (int w = 0; w < kThreads / 64; w++) add += s_wave[w];
      s_total += add;
    }
    __syncthreads();
  }
  if (tid == 0) num_bytes[d] = s_total;
}

// ---- keypoints -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_coeffs(float x, float* c) {            // resize.cpp interpolateCubic
  const float a = -0.75f;
  c[0] = ((a * (x + 1.f) - 5.f * a) * (x + 1.f) + 8.f * a) * (x + 1.f) - 4.f * a;
  c[1] = ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  c[2] = ((a + 2.f) * (1.f - x) - (a + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

constexpr int kMaxHeat = 64;   // HEATMAP_SIZE <= 64 (the reference uses 56)

__global__ void __launch_bounds__(kThreads)
keypoint_decode(const float* __restrict__ maps, const float* __restrict__ rois, int num_keypoints, int heat, int min_size,
                float* __restrict__ xy_preds) {
  __shared__ float s_map[kMaxHeat * kMaxHeat];
  __shared__ float s_max[kThreads / 64], s_sum[kThreads / 64];
  __shared__ long long s_arg[kThreads / 64];
  const int r = blockIdx.x / num_keypoints, k = blockIdx.x - r * num_keypoints;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* src = maps + ((long long)r * num_keypoints + k) * heat * heat;
  for (int i = tid; i < heat * heat; i += kThreads) s_map[i] = src[i];
  const float* roi = rois + (long long)r * 4;
  const float offset_x = roi[0], offset_y = roi[1];                          // lib/utils/keypoints.py:116-124
  const float width = fmaxf(roi[2] - roi[0], 1.f), height = fmaxf(roi[3] - roi[1], 1.f);
  int map_w = (int)ceilf(width), map_h = (int)ceilf(height);
  if (min_size > 0) {                                                       // :131-136
    map_w = max(map_w, min_size);
    map_h = max(map_h, min_size);
  }
  const float width_correction = width / (float)map_w, height_correction = height / (float)map_h;
  const double scale_x = 1.0 / ((double)map_w / (double)heat), scale_y = 1.0 / ((double)map_h / (double)heat);
  __syncthreads();
  // lane-private running state over its pixels (row-major order within the lane is ascending, so "first maximum" holds)
  float best = -__builtin_inff(), run_max = -__builtin_inff(), run_sum = 0.f;
  long long best_pos = 0x7fffffffffffffffLL;
  const long long npix = (long long)map_w * map_h;
  for (long long p = tid; p < npix; p += kThreads) {
    const int y = (int)(p / map_w), x = (int)(p - (long long)y * map_w);
    float fx = (float)(((double)x + 0.5) * scale_x - 0.5), fy = (float)(((double)y + 0.5) * scale_y - 0.5);
    const int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= (float)sx;
    fy -= (float)sy;
    float cx[4], cy[4];
    cubic_coeffs(fx, cx);
    cubic_coeffs(fy, cy);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {                                            // VResizeCubic over the four clipped rows
      const int row = min(max(sy - 1 + j, 0), heat - 1);
      float hsum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; i++) {                                          // HResizeCubic, replicated border
        const int col = min(max(sx - 1 + i, 0), heat - 1);
        hsum = hsum + s_map[row * heat + col] * cx[i];
      }
      v = v + hsum * cy[j];
    }
    if (v > best) {
      best = v;
      best_pos = p;
    }
    if (v > run_max) {                                                       // online softmax denominator
      run_sum = run_sum * __expf(run_max - v) + 1.f;
      run_max = v;
    } else {
      run_sum += __expf(v - run_max);
    }
  }
  // block reduction: maximum (ties: lowest position), then the sum rescaled to it
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const float ob = __shfl_xor(best, d, 64);
    const long long op = __shfl_xor(best_pos, d, 64);
    if (ob > best || (ob == best && op < best_pos)) {
      best = ob;
      best_pos = op;
    }
  }
  if (lane == 0) {
    s_max[wave] = best;
    s_arg[wave] = best_pos;
  }
  __syncthreads();
  float gmax = s_max[0];
  long long garg = s_arg[0];
  for (int w = 1; w < kThreads / 64; w++)
    if (s_max[w] > gmax || (s_max[w] == gmax && s_arg[w] < garg)) {
      gmax = s_max[w];
      garg = s_arg[w];
    }
  float part = run_max > -__builtin_inff() ? run_sum * expf(run_max - gmax) : 0.f;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
  if (lane == 0) s_sum[wave] = part;
  __syncthreads();
  if (tid == 0) {
    float denom = 0.f;
    for (int w = 0; w < kThreads / 64; w++) denom += s_sum[w];
    const int x_int = (int)(garg % map_w), y_int = (int)(garg / map_w);
    // :150-155; (x_int + 0.5) is float64 in the reference (np.int64 + python float), the products follow
    float* o = xy_preds + (long long)r * 4 * num_keypoints + k;
    o[0 * num_keypoints] = (float)(((double)x_int + 0.5) * (double)width_correction + (double)offset_x);
    o[1 * num_keypoints] = (float)(((double)y_int + 0.5) * (double)height_correction + (double)offset_y);
    o[2 * num_keypoints] = gmax;
    o[3 * num_keypoints] = 1.f / denom;                                      // exp(max - max) / sum
  }
}


// ---- OKS-NMS of keypoint predictions (lib/utils/keypoints.py:225-266; call site lib/core/test.py:857-862) ---------
// One workgroup: scores = numpy's fp32 mean of the 17 keypoint logits (its pairwise order), rank by counting, the
// N x N relation "OKS(src = i, dst = j) > thresh" as bit rows in LDS (fp32 squared distances, then fp64 exactly as
// numpy promotes them: / vars / (area + spacing(1)) / 2, exp, pairwise sum, / 17), then the greedy walk.
constexpr int kOksMax = 512;  // persons per image
constexpr int kOksK = 17;     // the reference's sigma table is COCO's 17 keypoints
__device__ __forceinline__ double oks_var(int k) {
  const double s[kOksK] = {.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89};
  const double v = (s[k] / 10.0) * 2;
  return v * v;
}
template <typename T>
__device__ __forceinline__ T numpy_sum17(const T* a) {  // numpy's pairwise_sum for 8 <= n <= 128, n = 17
  T r[8];
#pragma unroll
  for (int j = 0; j < 8; j++) r[j] = a[j];
#pragma unroll
  for (int j = 0; j < 8; j++) r[j] += a[8 + j];
  T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  res += a[16];
  return res;
}
__global__ void __launch_bounds__(kThreads)
keypoint_nms_oks(const float* __restrict__ kp, const float* __restrict__ rois, int n, double thresh,
                 int64_t* __restrict__ keep, int32_t* __restrict__ num_keep) {
  __shared__ float score[kOksMax];
  __shared__ int order[kOksMax];
  __shared__ unsigned long long sup[kOksMax * (kOksMax / 64)];
  const int tid = threadIdx.x, words = (n + 63) / 64;
  for (int i = tid; i < n; i += kThreads) score[i] = numpy_sum17(kp + ((long long)i * 4 + 2) * kOksK) / (float)kOksK;
  for (int i = tid; i < n * words; i += kThreads) sup[i] = 0ull;
  __syncthreads();
  // scores.argsort()[::-1]: descending; equal scores: the higher index first (a stable ascending sort, reversed)
  for (int i = tid; i < n; i += kThreads) {
    const float s = score[i];
    int rank = 0;
    for (int j = 0; j < n; j++) rank += (score[j] > s || (score[j] == s && j > i)) ? 1 : 0;
    order[rank] = i;
  }
  for (int p = tid; p < n * n; p += kThreads) {
    const int i = p / n, j = p - i * n;  // src i, dst j
    if (i == j) continue;
    const float* ri = rois + (long long)i * 4;
    const float area = (ri[2] - ri[0] + 1.f) * (ri[3] - ri[1] + 1.f);
    const double denom = (double)area + 2.220446049250313e-16;  // np.spacing(1)
    const float* si = kp + (long long)i * 4 * kOksK;
    const float* dj = kp + (long long)j * 4 * kOksK;
    double ex[kOksK];
#pragma unroll
    for (int k = 0; k < kOksK; k++) {
      const float dx = dj[k] - si[k], dy = dj[kOksK + k] - si[kOksK + k];
      const float d2 = dx * dx + dy * dy;
      ex[k] = exp(-((double)d2 / oks_var(k) / denom / 2));
    }
    const double oks = numpy_sum17(ex) / (double)kOksK;
    if (!(oks <= thresh)) atomicOr(&sup[i * words + (j >> 6)], 1ull << (j & 63));
  }
  __syncthreads();
  if (tid == 0) {
    unsigned long long dead[kOksMax / 64];
    for (int w = 0; w < words; w++) dead[w] = 0ull;
    int kept = 0;
    for (int r = 0; r < n; r++) {
      const int i = order[r];
      if ((dead[i >> 6] >> (i & 63)) & 1ull) continue;
      keep[kept++] = i;
      for (int w = 0; w < words; w++) dead[w] |= sup[i * words + w];
    }
    *num_keep = kept;
  }
}

}  // namespace

extern "C" int mi_mask_paste_rle(const float* masks, const int32_t* boxes, int num_masks, int mask_size, int im_height,
                                 int im_width, float thresh, int capacity, uint32_t* counts, int32_t* num_counts,
                                 int string_capacity, uint8_t* strings, int32_t* num_bytes, mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(num_masks >= 0 && mask_size > 0 && im_height > 0 && im_width > 0 && capacity > 0, "mask_paste_rle: bad size");
  MI_REQUIRE(mask_size + 2 <= kMaxMask, "mask_paste_rle: masks of at most %d x %d are supported", kMaxMask - 2, kMaxMask - 2);
  MI_REQUIRE((long long)im_height * im_width < (1LL << 32), "mask_paste_rle: image too large for 32-bit run lengths");
  if (num_masks == 0) return MI_OK;
  MI_REQUIRE(masks != nullptr && boxes != nullptr && counts != nullptr && num_counts != nullptr,
             "mask_paste_rle: null pointer");
  MI_REQUIRE(strings == nullptr || (num_bytes != nullptr && string_capacity > 0), "mask_paste_rle: strings without a size");
  mask_paste_rle<<<num_masks, kThreads, 0, mi::as_stream(stream)>>>(masks, boxes, mask_size, im_height, im_width, thresh,
                                                                   capacity, counts, num_counts, string_capacity, strings,
                                                                   num_bytes);
  return mi::check_launch("mask_paste_rle");
}

extern "C" int mi_keypoint_decode(const float* heatmaps, const float* rois, int num_rois, int num_keypoints,
                                  int heatmap_size, int min_size, float* xy_preds, mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(num_rois >= 0 && num_keypoints > 0 && heatmap_size > 0, "keypoint_decode: bad size");
  MI_REQUIRE(heatmap_size <= kMaxHeat, "keypoint_decode: heat maps of at most %d x %d are supported", kMaxHeat, kMaxHeat);
  if (num_rois == 0) return MI_OK;
  MI_REQUIRE(heatmaps != nullptr && rois != nullptr && xy_preds != nullptr, "keypoint_decode: null pointer");
  keypoint_decode<<<num_rois * num_keypoints, kThreads, 0, mi::as_stream(stream)>>>(heatmaps, rois, num_keypoints,
                                                                                   heatmap_size, min_size, xy_preds);
  return mi::check_launch("keypoint_decode");
}

extern "C" int mi_keypoint_nms_oks(const float* xy_preds, const float* rois, int num_rois, int num_keypoints, double thresh,
                                   int64_t* keep, int32_t* num_keep, mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(num_rois >= 0, "keypoint_nms_oks: negative size");
  MI_REQUIRE(num_keypoints == kOksK, "keypoint_nms_oks: the reference's sigma table has %d keypoints (got %d)", kOksK,
             num_keypoints);
  MI_REQUIRE(num_rois <= kOksMax, "keypoint_nms_oks: at most %d persons per call (got %d)", kOksMax, num_rois);
  MI_REQUIRE(num_keep != nullptr, "keypoint_nms_oks: null pointer");
  if (num_rois == 0) {
    if (hipMemsetAsync(num_keep, 0, sizeof(int32_t), mi::as_stream(stream)) != hipSuccess)
      return mi::check_launch("keypoint_nms_oks: zero count");
    return MI_OK;
  }
  MI_REQUIRE(xy_preds != nullptr && rois != nullptr && keep != nullptr, "keypoint_nms_oks: null pointer");
  keypoint_nms_oks<<<1, kThreads, 0, mi::as_stream(stream)>>>(xy_preds, rois, num_rois, thresh, keep, num_keep);
  return mi::check_launch("keypoint_nms_oks");
}
