// roi_align_stream.hip -- the NCHW fast path of RoIAlign BACKWARD (Caffe2 semantics) for gfx950
// (forward fast path: roi_align_fwd_tile.hip).
//
// Same arithmetic, operation for operation, as roi_align_fwd_direct / roi_align_bwd_direct in
// roi_align.hip (reference: lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu:16-121, :150-270);
// what changes is where the data moves.
//
// The reference mapping (one lane per output element) issues 4*samples scattered 4-byte
// loads per output on a channel-planar tensor: neighbouring lanes hit different rows/columns,
// nothing coalesces, and every feature pixel under a RoI is fetched from L2 ~5 times.
//
// Here one 256-lane workgroup owns (RoI, 32-channel tile) and streams the RoI's feature window
// through LDS exactly once:
//   * window rows are loaded top to bottom into an 8-row LDS ring, 32 lanes per row segment
//     (coalesced), only when the next row of output bins needs them -- LDS use is bounded
//     (33 KB) no matter how tall the RoI is, so 3 workgroups share a CU;
//   * a lane owns one channel (lane & 31); the two 32-lane halves of each wavefront take
//     different output columns pw.  The ring's per-channel plane stride is odd (257 words), so
//     the 32 lanes of a half always hit 32 distinct LDS banks: every bilinear tap is a
//     conflict-free ds_read_b32, and all sampling geometry is identical across a half-wave;
//   * per-RoI sampling tables (row/column of each tap, its two weights) are built once per
//     workgroup in LDS instead of 4*samples times per output element;
//   * results are staged in LDS as [channel][bin] (odd stride) and leave as one contiguous
//     6.3 KB run per workgroup.
// The backward is the mirror image: top-gradient tile staged through LDS, gradients accumulated
// into the LDS ring with ds_add_f32, each ring row flushed to HBM with ONE coalesced atomic row
// add when the sweep leaves it -- window-size atomics instead of 16 per output element.
//
// RoIs the ring cannot hold (window wider than 32 columns, a bin row taller than 8 feature rows,
// more than 64 samples per axis) take the in-kernel direct path; results are identical.
#include "common.h"
#include "roi_align_device.h"

namespace mi {
namespace {

constexpr int kCT = 32;                 // channels per workgroup
constexpr int kNR = 8;                  // ring rows (power of two)
constexpr int kRW = 32;                 // ring row width == widest window on the fast path
constexpr int kPS = kNR * kRW + 1;      // odd plane stride (words)
constexpr int kMaxS = 64;               // samples per axis the tables hold
constexpr int kThreads = 256;
constexpr int kSlots = kThreads / 32;   // half-waves

struct AxisEntry {
  int lo, hi;     // absolute row (y table) or column (x table); lo < 0: sample outside the band -> contributes 0
  float hw, lw;   // weight of lo (1 - frac) and of hi (frac)
};

// One axis of roi_align_kernel.cu:16-52 (the y and x halves of bilinear_interpolate are independent).
__device__ __forceinline__ AxisEntry axis_entry(float v, int size) {
  AxisEntry e;
  if (v < -1.0f || v > (float)size) {
    e.lo = e.hi = -1;
    e.hw = e.lw = 0.f;
    return e;
  }
  if (v <= 0) v = 0;
  int low = (int)v, high;
  if (low >= size - 1) {
    high = low = size - 1;
    v = (float)low;
  } else {
    high = low + 1;
  }
  const float l = v - (float)low;
  e.lo = low;
  e.hi = high;
  e.lw = l;
  e.hw = 1.f - l;
  return e;
}

struct Shared {
  float* ring;     // [kCT][kPS]
  float* tile;     // [kCT][OS]  forward: outputs; backward: top gradients
  AxisEntry* ty;   // [kMaxS]
  AxisEntry* tx;   // [kMaxS]
  int* misc;       // [0]=wx0 [1]=wx1
};

__device__ __forceinline__ Shared carve_shared(float* smem, int os) {
  Shared s;
  s.ring = smem;
  s.tile = s.ring + kCT * kPS;
  float* p = s.tile + kCT * os;
  p += (4 - ((kCT * kPS + kCT * os) & 3)) & 3;  // 16-byte align the tables
  s.ty = reinterpret_cast<AxisEntry*>(p);
  s.tx = s.ty + kMaxS;
  s.misc = reinterpret_cast<int*>(s.tx + kMaxS);
  return s;
}

__host__ __device__ inline int tile_stride(int bins) { return bins | 1; }

__host__ inline size_t shared_bytes(int bins) {
  size_t words = (size_t)kCT * kPS + (size_t)kCT * tile_stride(bins);
  words = (words + 3) & ~size_t(3);
  return words * 4 + 2 * kMaxS * sizeof(AxisEntry) + 16;
}

// Build the per-RoI tables; returns true when the ring can serve this RoI.
__device__ __forceinline__ bool build_tables(const Shared& s, const RoiGeom& g, int height, int width,
                                             int aligned_height, int aligned_width, int tid) {
  const int nsy = aligned_height * g.grid_h, nsx = aligned_width * g.grid_w;
  if (nsy > kMaxS || nsx > kMaxS) return false;  // uniform
  if (tid == 0) {
    s.misc[0] = 0x7fffffff;
    s.misc[1] = -1;
  }
  __syncthreads();
  if (tid < nsy) {
    s.ty[tid] = axis_entry(sample_y(g, tid / g.grid_h, tid % g.grid_h), height);
  } else if (tid >= 64 && tid - 64 < nsx) {
    const int k = tid - 64;
    AxisEntry e = axis_entry(sample_x(g, k / g.grid_w, k % g.grid_w), width);
    s.tx[k] = e;
    if (e.lo >= 0) {
      atomicMin(&s.misc[0], e.lo);
      atomicMax(&s.misc[1], e.hi);
    }
  }
  __syncthreads();
  const int wx0 = s.misc[0], wx1 = s.misc[1];
  if (wx1 >= 0 && wx1 - wx0 + 1 > kRW) return false;
  for (int ph = 0; ph < aligned_height; ph++) {  // each bin row must fit the ring
    int ya = 0x7fffffff, yb = -1;
    for (int iy = 0; iy < g.grid_h; iy++) {
      const AxisEntry e = s.ty[ph * g.grid_h + iy];
      if (e.lo >= 0) {
        ya = min(ya, e.lo);
        yb = max(yb, e.hi);
      }
    }
    if (yb >= 0 && yb - ya + 1 > kNR) return false;
  }
  return true;
}

__device__ __forceinline__ void band_rows(const Shared& s, const RoiGeom& g, int ph, int& ya, int& yb) {
  ya = 0x7fffffff;
  yb = -1;
  for (int iy = 0; iy < g.grid_h; iy++) {
    const AxisEntry e = s.ty[ph * g.grid_h + iy];
    if (e.lo >= 0) {
      ya = min(ya, e.lo);
      yb = max(yb, e.hi);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
roi_align_bwd_stream(const float* __restrict__ top_diff, const float* __restrict__ rois,
                     float* __restrict__ bottom_diff, int batch, int channels, int height, int width,
                     int aligned_height, int aligned_width, float spatial_scale, int sampling_ratio) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bins = aligned_height * aligned_width;
  const int os = tile_stride(bins);
  const Shared s = carve_shared(smem, os);
  const int tid = threadIdx.x;
  const int tiles = channels / kCT;
  const int r = blockIdx.x / tiles;
  const int c0 = (blockIdx.x - r * tiles) * kCT;
  const RoiGeom g = roi_geometry(rois + (long long)r * 5, spatial_scale, aligned_height, aligned_width,
                                 sampling_ratio);
  if (g.batch_ind < 0 || g.batch_ind >= batch) return;
  const float* __restrict__ gsrc = top_diff + ((long long)r * channels + c0) * bins;
  float* __restrict__ gdst = bottom_diff + ((long long)g.batch_ind * channels + c0) * height * width;
  const bool fast = build_tables(s, g, height, width, aligned_height, aligned_width, tid);

  if (!fast) {
    for (int i = tid; i < kCT * bins; i += kThreads) {
      const int c = i / bins, bin = i - c * bins;
      const int ph = bin / aligned_width, pw = bin - ph * aligned_width;
      float* plane = gdst + (long long)c * height * width;
      const float top_diff_this_bin = gsrc[i];
      for (int iy = 0; iy < g.grid_h; iy++) {
        const float y = sample_y(g, ph, iy);
        for (int ix = 0; ix < g.grid_w; ix++) {
          const float x = sample_x(g, pw, ix);
          const Taps t = sample_taps(height, width, y, x);
          if (t.y_low < 0) continue;
          atomicAdd(plane + t.y_low * width + t.x_low, top_diff_this_bin * t.w1 / g.count);
          atomicAdd(plane + t.y_low * width + t.x_high, top_diff_this_bin * t.w2 / g.count);
          atomicAdd(plane + t.y_high * width + t.x_low, top_diff_this_bin * t.w3 / g.count);
          atomicAdd(plane + t.y_high * width + t.x_high, top_diff_this_bin * t.w4 / g.count);
        }
      }
    }
    return;
  }

  // stage the top-gradient tile [kCT][bins] (one contiguous run in HBM) and clear the ring
  for (int i = tid; i < kCT * bins; i += kThreads) {
    const int c = i / bins, bin = i - c * bins;
    s.tile[c * os + bin] = gsrc[i];
  }
  for (int i = tid; i < kCT * kPS; i += kThreads) s.ring[i] = 0.f;

  const int wx0 = s.misc[0], wx1 = s.misc[1];
  const int ww = wx1 - wx0 + 1;
  const int lx = tid & 31, slot = tid >> 5;
  const int cl = tid & 31;
  float* ring_c = s.ring + cl * kPS;
  int live_lo = -1, live_hi = -1;  // rows currently accumulated in the ring: [live_lo, live_hi]
  __syncthreads();

  // flush rows [from, to] of the ring to HBM (coalesced row segments, one atomic per pixel) and clear them
  auto flush = [&](int from, int to) {
    if (lx < ww) {
#pragma unroll
      for (int cc = 0; cc < kCT / kSlots; cc++) {
        const int c = slot + cc * kSlots;
        float* plane = gdst + (long long)c * height * width + wx0 + lx;
        float* rc = s.ring + c * kPS + lx;
        for (int y = from; y <= to; y++) {
          float* cell = rc + (y & (kNR - 1)) * kRW;
          const float v = *cell;
          if (v != 0.f) atomicAdd(plane + y * width, v);
          *cell = 0.f;
        }
      }
    }
  };

  for (int ph = 0; ph < aligned_height; ph++) {
    int ya, yb;
    band_rows(s, g, ph, ya, yb);
    if (yb < 0 || ww <= 0) continue;  // uniform: this bin row has no sample inside the band
    if (live_hi >= 0 && ya > live_lo) {
      // rows below ya are final: retire them before their slots are reused
      const int to = min(ya - 1, live_hi);
      __syncthreads();
      flush(live_lo, to);
      __syncthreads();
      live_lo = to + 1;
    }
    if (live_hi < 0 || live_lo > live_hi) live_lo = ya;
    live_hi = max(live_hi, yb);
    for (int pw = slot; pw < aligned_width; pw += kSlots) {
      const float top_diff_this_bin = s.tile[cl * os + ph * aligned_width + pw];
      for (int iy = 0; iy < g.grid_h; iy++) {
        const AxisEntry ey = s.ty[ph * g.grid_h + iy];
        const int rlo = (ey.lo & (kNR - 1)) * kRW, rhi = (ey.hi & (kNR - 1)) * kRW;
        for (int ix = 0; ix < g.grid_w; ix++) {
          const AxisEntry ex = s.tx[pw * g.grid_w + ix];
          if (ey.lo < 0 || ex.lo < 0) continue;
          const int xl = ex.lo - wx0, xh = ex.hi - wx0;
          const float w1 = ey.hw * ex.hw, w2 = ey.hw * ex.lw, w3 = ey.lw * ex.hw, w4 = ey.lw * ex.lw;
          atomicAdd(ring_c + rlo + xl, top_diff_this_bin * w1 / g.count);  // roi_align_kernel.cu:252-265
          atomicAdd(ring_c + rlo + xh, top_diff_this_bin * w2 / g.count);
          atomicAdd(ring_c + rhi + xl, top_diff_this_bin * w3 / g.count);
          atomicAdd(ring_c + rhi + xh, top_diff_this_bin * w4 / g.count);
        }
      }
    }
  }
  __syncthreads();
  if (live_hi >= 0 && live_lo <= live_hi) flush(live_lo, live_hi);
}

}  // namespace

bool roi_align_stream_supported(int channels, int aligned_height, int aligned_width) {
  return channels > 0 && channels % kCT == 0 && shared_bytes(aligned_height * aligned_width) <= 64 * 1024;
}

int launch_roi_align_bwd_stream(const float* top_grad, const float* rois, float* bottom_grad, int batch,
                                int channels, int height, int width, int num_rois, int aligned_height,
                                int aligned_width, float spatial_scale, int sampling_ratio,
                                hipStream_t stream) {
  const int grid = num_rois * (channels / kCT);
  roi_align_bwd_stream<<<grid, kThreads, shared_bytes(aligned_height * aligned_width), stream>>>(
      top_grad, rois, bottom_grad, batch, channels, height, width, aligned_height, aligned_width,
      spatial_scale, sampling_ratio);
  return check_launch("roi_align_bwd_stream");
}

}  // namespace mi
