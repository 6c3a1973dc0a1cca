// roi_align_fwd_tile.hip -- the NCHW fast path of RoIAlign forward (Caffe2 semantics) for gfx950.
//
// Arithmetic: the products and sums of roi_align_fwd_direct in roi_align.hip (reference:
// lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu:16-121), in the same order, so outputs are equal to
// the CPU oracle (-ffp-contract=off).  What changes is where the data moves and what a tap costs.
//
// Why not the reference mapping: one lane per output element issues 4*samples scattered 4-byte loads per
// output on a channel-planar tensor.  Neighbouring lanes hit different rows and columns, a wave-load touches
// ~14 cache lines, the texture-address unit serialises them (rocprofv3: 103 us at 512x256x7x7 on a 200x336
// map) and every lane recomputes the tap geometry that is identical for all 256 channels of a RoI.
//
// One 256-lane workgroup owns (RoI, 32-channel tile):
//   * window = the feature rows/columns any sample of the RoI touches.  It is copied HBM/L2 -> LDS once, by
//     LDS-DMA (buffer_load_dword ... lds): no VGPR staging, no ds_write pass, and the WHOLE window of the
//     tile is in flight at once (8 channels x ~5 pieces of 256 B per wave) instead of register-sized batches;
//     a lane's piece index is flattened over (row, column) so that the LDS image is compact ([row][ww]) and
//     every piece moves 64 useful pixels whatever the window width;
//   * LDS image: one plane per channel with an ODD plane stride, lane & 31 = channel: the 32 lanes of a
//     half-wave read the same (row, column) of 32 different planes -> 32 distinct banks, every bilinear tap
//     is a conflict-free ds_read (two taps of a row with one ds_read2_b32), and all sampling geometry is
//     identical across the half-wave;
//   * the 8 half-waves take different output columns pw; tap rows/columns (as LDS byte offsets) and the two
//     weights per axis sample are computed once per workgroup into two small LDS tables;
//   * clamped border samples are expressed as the pixel pair (size-2, size-1) with weights (0, 1) instead of
//     the reference's (size-1, size-1) with (1, 0): the same value for finite features, and "high = low + 1"
//     holds for every sample, which is what makes the fixed +4 / +pitch tap addressing possible;
//   * results are staged in LDS as [channel][bin] and leave as contiguous 16-byte stores.
// Windows larger than the LDS image are processed in groups of bin rows (each group: DMA, barrier, compute);
// output tiles larger than the LDS staging area likewise.  RoIs the scheme cannot serve (a sample outside the
// [-1, size] band, one bin row larger than the LDS image, > 64 samples per axis, H or W < 2) take the in-kernel
// direct path with identical results.
#include "common.h"
#include "roi_align_device.h"

namespace mi {
namespace {

constexpr int kCT = 32;                    // channels per workgroup
constexpr int kThreads = 256;
constexpr int kSlots = kThreads / 32;      // half-waves; each owns output columns pw = slot, slot + 8, ...
constexpr int kWaves = kThreads / 64;
constexpr int kChPerWave = kCT / kWaves;   // planes a wave fills
constexpr int kMaxS = 64;                  // samples per axis the tables hold
constexpr int kTileBins = 64;              // output bins per channel staged in LDS between two stores

struct AxisEntry {
  int off;       // y table: (row_lo - wy0) * ww * 4 ; x table: (col_lo - wx0) * 4   (LDS byte offsets)
  float hw, lw;  // weight of lo and of lo + 1
  int lo;        // absolute row / column of the lower tap
};

// One axis of roi_align_kernel.cu:16-52 for a sample already known to lie inside the [-1, size] band.
// Returns the lower tap and the weights of (lo, lo + 1); see the header comment for the border case.
__device__ __forceinline__ void axis_taps(float v, int size, int& lo, float& hw, float& lw) {
  if (v <= 0) v = 0;
  int low = (int)v;
  if (low >= size - 1) {
    lo = size - 2;
    hw = 0.f;  // reference: low = high = size - 1, l = 0, h = 1
    lw = 1.f;
  } else {
    lo = low;
    lw = v - (float)low;
    hw = 1.f - lw;
  }
}

template <int kCap>
struct Lds {
  static constexpr int kPlane = kCap + 1;              // odd plane stride (words); kCap is a multiple of 64
  static constexpr int kTileWords = kCT * (kTileBins + 1);
  float* img;     // [kCT][kPlane]
  float* tile;    // [kCT][ts]
  AxisEntry* ty;  // [kMaxS]
  AxisEntry* tx;  // [kMaxS]
  __device__ __forceinline__ explicit Lds(float* smem) {
    ty = reinterpret_cast<AxisEntry*>(smem);
    tx = ty + kMaxS;
    tile = reinterpret_cast<float*>(tx + kMaxS);
    img = tile + kTileWords;
  }
  static constexpr size_t bytes() { return 2 * kMaxS * sizeof(AxisEntry) + (size_t)(kTileWords + kCT * kPlane) * 4; }
};

__device__ __forceinline__ void store_zero_tile(float* dst, int n, int tid) {
  for (int i = tid; i < n; i += kThreads) dst[i] = 0.f;
}

__device__ __forceinline__ float lds_f(const float* base, int byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

using lds_ptr_t = __attribute__((address_space(3))) void*;

// kSR > 0: sampling_ratio == kSR at compile time.  kSR == 0: run-time grid (adaptive ratio, or any other value).
template <int kSR, int kCap>
__global__ void __launch_bounds__(kThreads)
roi_align_fwd_tile(const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
                   int batch, int channels, int height, int width, int aligned_height, int aligned_width,
                   float spatial_scale, int sampling_ratio) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Lds<kCap> s(smem);
  constexpr int kPlane = Lds<kCap>::kPlane;
  const int bins = aligned_height * aligned_width;
  const int tid = threadIdx.x;
  const int tiles = channels / kCT;
  const int r = blockIdx.x / tiles;
  const int c0 = (blockIdx.x - r * tiles) * kCT;
  float* __restrict__ dst = out + ((long long)r * channels + c0) * bins;
  const RoiGeom g = roi_geometry(rois + (long long)r * 5, spatial_scale, aligned_height, aligned_width,
                                 sampling_ratio);
  if (g.batch_ind < 0 || g.batch_ind >= batch) {  // same guard as the direct kernel
    store_zero_tile(dst, kCT * bins, tid);
    return;
  }
  const int batch_ind = __builtin_amdgcn_readfirstlane(g.batch_ind);
  const float* __restrict__ src = feat + ((long long)batch_ind * channels + c0) * height * width;
  const int gh = kSR > 0 ? kSR : g.grid_h, gw = kSR > 0 ? kSR : g.grid_w;
  const int nsy = aligned_height * gh, nsx = aligned_width * gw;

  // ---- window: first and last sample of each axis (sample coordinates are monotonic in the sample index) ----
  const float yf = sample_y(g, 0, 0), yl = sample_y(g, aligned_height - 1, gh - 1);
  const float xf = sample_x(g, 0, 0), xl = sample_x(g, aligned_width - 1, gw - 1);
  bool fast = nsy <= kMaxS && nsx <= kMaxS && height >= 2 && width >= 2 &&
              !(yf < -1.0f || yl > (float)height || xf < -1.0f || xl > (float)width) && yl >= yf && xl >= xf;
  int wy0 = 0, wy1 = 0, wx0 = 0, wx1 = 0;
  {
    float hw, lw;
    axis_taps(yf, height, wy0, hw, lw);
    axis_taps(yl, height, wy1, hw, lw);
    axis_taps(xf, width, wx0, hw, lw);
    axis_taps(xl, width, wx1, hw, lw);
    wy0 = __builtin_amdgcn_readfirstlane(wy0);
    wy1 = __builtin_amdgcn_readfirstlane(wy1) + 1;
    wx0 = __builtin_amdgcn_readfirstlane(wx0);
    wx1 = __builtin_amdgcn_readfirstlane(wx1) + 1;
  }
  fast = __builtin_amdgcn_readfirstlane(fast);
  const int ww = wx1 - wx0 + 1, nrows = wy1 - wy0 + 1;
  const int pitch = ww * 4;                                  // bytes between two window rows in a plane
  const bool single = fast && nrows * ww <= kCap;            // the whole window fits the LDS image

  // ---- LDS-DMA of window rows [row0, row0 + nr) -> img[channel][(row - row0) * ww + col] ----
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned plane_bytes = (unsigned)height * (unsigned)width * 4u;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(src + (long long)wave * kChPerWave * height * width), /*stride*/ 0,
      (int)((unsigned)kChPerWave * plane_bytes), 0x00020000);
  const unsigned magic = fast ? ((1u << 20) / (unsigned)ww + 1u) : 0u;  // p / ww == (p * magic) >> 20 for p*ww < 2^20
  auto issue_dma = [&](int row0, int nr) {
    const int npx = nr * ww;
    float* plane0 = s.img + wave * kChPerWave * kPlane;
    for (int k = 0; k * 64 < npx; k++) {
      const unsigned p = (unsigned)(k * 64 + lane);
      const unsigned q = (p * magic) >> 20;
      const unsigned col = p - q * (unsigned)ww;
      // lanes past the window carry an out-of-range offset: the buffer bounds check answers 0 without a memory access
      const unsigned voff = p < (unsigned)npx ? (((unsigned)row0 + q) * (unsigned)width + (unsigned)wx0 + col) * 4u
                                              : 0xffffff00u;
#pragma unroll
      for (int c = 0; c < kChPerWave; c++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(plane0 + c * kPlane + k * 64), 4, voff,
                                                 c * plane_bytes, 0, 0);
    }
  };
  if (single) issue_dma(wy0, nrows);

  // ---- tables (threads 0.. : y samples, threads 64.. : x samples) ----
  if (fast) {
    if (tid < nsy) {
      AxisEntry e;
      axis_taps(sample_y(g, tid / gh, tid % gh), height, e.lo, e.hw, e.lw);
      e.lo = min(max(e.lo, wy0), wy1 - 1);
      e.off = (e.lo - wy0) * pitch;
      s.ty[tid] = e;
    } else if (tid >= 64 && tid - 64 < nsx) {
      const int k = tid - 64;
      AxisEntry e;
      axis_taps(sample_x(g, k / gw, k % gw), width, e.lo, e.hw, e.lw);
      e.lo = min(max(e.lo, wx0), wx1 - 1);
      e.off = (e.lo - wx0) * 4;
      s.tx[k] = e;
    }
    __syncthreads();  // tables visible; the DMA of a single-pass window has landed (vmcnt(0) precedes the barrier)
    if (!single) {    // every bin row must fit the LDS image on its own
      for (int ph = 0; fast && ph < aligned_height; ph++) {
        const int lo = s.ty[ph * gh].lo, hi = s.ty[ph * gh + gh - 1].lo + 1;
        if ((hi - lo + 1) * ww > kCap) fast = false;
      }
      fast = __builtin_amdgcn_readfirstlane(fast);
    }
  }

  if (!fast) {
    // direct path for this (RoI, channel tile): reference mapping, coalesced stores
    for (int i = tid; i < kCT * bins; i += kThreads) {
      const int c = i / bins, bin = i - c * bins;
      const int ph = bin / aligned_width, pw = bin - ph * aligned_width;
      const float* plane = src + (long long)c * height * width;
      float output_val = 0.f;
      for (int iy = 0; iy < g.grid_h; iy++) {
        const float y = sample_y(g, ph, iy);
        for (int ix = 0; ix < g.grid_w; ix++) {
          const float x = sample_x(g, pw, ix);
          const Taps t = sample_taps(height, width, y, x);
          float val = 0.f;
          if (t.y_low >= 0) {
            const float v1 = plane[t.y_low * width + t.x_low], v2 = plane[t.y_low * width + t.x_high];
            const float v3 = plane[t.y_high * width + t.x_low], v4 = plane[t.y_high * width + t.x_high];
            val = (t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4);
          }
          output_val += val;
        }
      }
      dst[i] = output_val / g.count;
    }
    return;
  }

  const int cl = tid & 31, slot = tid >> 5;
  const float* img_c = s.img + cl * kPlane;
  const int max_rows_tile = max(kTileBins / aligned_width, 1);  // bin rows per output staging pass

  for (int ph0 = 0; ph0 < aligned_height;) {
    // ---- group of bin rows [ph0, ph1): fits the output staging tile and (when streaming) the LDS image ----
    int ph1 = min(aligned_height, ph0 + max_rows_tile);
    int row0 = wy0;
    if (!single) {
      row0 = s.ty[ph0 * gh].lo;
      int e = ph0 + 1;
      while (e < ph1 && (s.ty[e * gh + gh - 1].lo + 1 - row0 + 1) * ww <= kCap) e++;
      ph1 = e;
      const int row1 = s.ty[(ph1 - 1) * gh + gh - 1].lo + 1;
      row0 = __builtin_amdgcn_readfirstlane(row0);
      ph1 = __builtin_amdgcn_readfirstlane(ph1);
      issue_dma(row0, __builtin_amdgcn_readfirstlane(row1) - row0 + 1);
      __syncthreads();
    }
    const int base_off = (row0 - wy0) * pitch;  // table offsets are relative to wy0
    const int nb = (ph1 - ph0) * aligned_width;
    const int ts = nb | 1;

    for (int pw = slot; pw < aligned_width; pw += kSlots) {
      if (kSR > 0) {
        constexpr int kS = kSR > 0 ? kSR : 1;
        AxisEntry ex[kS];
        const float* xa[kS];
#pragma unroll
        for (int i = 0; i < kS; i++) {
          ex[i] = s.tx[pw * kS + i];
          xa[i] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(img_c) + (ex[i].off - base_off));
        }
        for (int ph = ph0; ph < ph1; ph++) {
          AxisEntry ey[kS];
#pragma unroll
          for (int i = 0; i < kS; i++) ey[i] = s.ty[ph * kS + i];
          float v[kS][kS][4];
#pragma unroll
          for (int iy = 0; iy < kS; iy++) {
#pragma unroll
            for (int ix = 0; ix < kS; ix++) {
              const float* a = reinterpret_cast<const float*>(reinterpret_cast<const char*>(xa[ix]) + ey[iy].off);
              const float* b = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a) + pitch);
              v[iy][ix][0] = a[0];
              v[iy][ix][1] = a[1];
              v[iy][ix][2] = b[0];
              v[iy][ix][3] = b[1];
            }
          }
          float output_val = 0.f;
#pragma unroll
          for (int iy = 0; iy < kS; iy++) {
#pragma unroll
            for (int ix = 0; ix < kS; ix++) {
              const float w1 = ey[iy].hw * ex[ix].hw, w2 = ey[iy].hw * ex[ix].lw;
              const float w3 = ey[iy].lw * ex[ix].hw, w4 = ey[iy].lw * ex[ix].lw;
              const float val = (w1 * v[iy][ix][0] + w2 * v[iy][ix][1] + w3 * v[iy][ix][2] + w4 * v[iy][ix][3]);
              output_val += val;
            }
          }
          constexpr float kInvCount = 1.f / (float)(kS * kS);
          // count = kSR^2: for a power of two the reciprocal multiply is exact and equals the division bit for bit
          output_val = ((kS & (kS - 1)) == 0) ? output_val * kInvCount : output_val / g.count;
          s.tile[cl * ts + (ph - ph0) * aligned_width + pw] = output_val;
        }
      } else {
        for (int ph = ph0; ph < ph1; ph++) {
          float output_val = 0.f;
          for (int iy = 0; iy < gh; iy++) {
            const AxisEntry ey = s.ty[ph * gh + iy];
            for (int ix = 0; ix < gw; ix++) {
              const AxisEntry ex = s.tx[pw * gw + ix];
              const int o = ey.off + ex.off - base_off;
              const float v1 = lds_f(img_c, o), v2 = lds_f(img_c, o + 4);
              const float v3 = lds_f(img_c, o + pitch), v4 = lds_f(img_c, o + pitch + 4);
              const float w1 = ey.hw * ex.hw, w2 = ey.hw * ex.lw, w3 = ey.lw * ex.hw, w4 = ey.lw * ex.lw;
              const float val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);  // roi_align_kernel.cu:58-60
              output_val += val;                                          // :113
            }
          }
          output_val = output_val / g.count;  // :117
          s.tile[cl * ts + (ph - ph0) * aligned_width + pw] = output_val;
        }
      }
    }
    __syncthreads();

    // ---- staged outputs -> HBM ----
    float* gdst = dst + ph0 * aligned_width;
    if (nb == bins && ts == nb && ((kCT * nb) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
      // the whole [kCT][bins] block is one contiguous run and the LDS tile has the same layout
      const float4* t4 = reinterpret_cast<const float4*>(s.tile);
      float4* d4 = reinterpret_cast<float4*>(dst);
      for (int i = tid; i < kCT * nb / 4; i += kThreads) d4[i] = t4[i];
    } else {
      const unsigned nb_magic = (1u << 20) / (unsigned)nb + 1u;
      for (int i = tid; i < kCT * nb; i += kThreads) {
        const int c = (int)(((unsigned)i * nb_magic) >> 20), b = i - c * nb;
        gdst[(long long)c * bins + b] = s.tile[c * ts + b];
      }
    }
    ph0 = ph1;
    if (ph0 < aligned_height) __syncthreads();  // the tile (and, when streaming, the image) is reused
  }
}

template <int kCap>
int launch_cap(const float* features, const float* rois, float* output, int batch, int channels, int height,
               int width, int num_rois, int aligned_height, int aligned_width, float spatial_scale,
               int sampling_ratio, hipStream_t stream) {
  const int grid = num_rois * (channels / kCT);
  const size_t lds = Lds<kCap>::bytes();
  if (sampling_ratio == 2)
    roi_align_fwd_tile<2, kCap><<<grid, kThreads, lds, stream>>>(features, rois, output, batch, channels, height,
                                                                 width, aligned_height, aligned_width,
                                                                 spatial_scale, sampling_ratio);
  else
    roi_align_fwd_tile<0, kCap><<<grid, kThreads, lds, stream>>>(features, rois, output, batch, channels, height,
                                                                 width, aligned_height, aligned_width,
                                                                 spatial_scale, sampling_ratio);
  return check_launch("roi_align_fwd_tile");
}

}  // namespace

bool roi_align_fwd_tile_supported(int channels, int height, int width, int aligned_height, int aligned_width) {
  // 32-bit byte offsets inside one (image, channel tile) slab of the DMA descriptor; a bin row fits the staging tile
  return channels > 0 && channels % kCT == 0 && aligned_width <= kTileBins && aligned_height > 0 &&
         (long long)kCT * height * width * 4 < (1LL << 31);
}

int launch_roi_align_fwd_tile(const float* features, const float* rois, float* output, int batch, int channels,
                              int height, int width, int num_rois, int aligned_height, int aligned_width,
                              float spatial_scale, int sampling_ratio, int cap_px, hipStream_t stream) {
  if (cap_px >= 384)
    return launch_cap<384>(features, rois, output, batch, channels, height, width, num_rois, aligned_height,
                           aligned_width, spatial_scale, sampling_ratio, stream);
  if (cap_px >= 320)
    return launch_cap<320>(features, rois, output, batch, channels, height, width, num_rois, aligned_height,
                           aligned_width, spatial_scale, sampling_ratio, stream);
  if (cap_px >= 256)
    return launch_cap<256>(features, rois, output, batch, channels, height, width, num_rois, aligned_height,
                           aligned_width, spatial_scale, sampling_ratio, stream);
  return launch_cap<192>(features, rois, output, batch, channels, height, width, num_rois, aligned_height,
                         aligned_width, spatial_scale, sampling_ratio, stream);
}

}  // namespace mi
