// roi_align_fwd_tile.hip -- the NCHW fast path of RoIAlign forward (Caffe2 semantics) for gfx950.
//
// Arithmetic contract: sample coordinates, tap rows/columns and interpolation weights are computed with exactly the
// reference's fp32 operations (lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu:74-110, :16-52), so every
// output reads the same feature pixels with the same weights as the reference.  The weighted sum itself is evaluated
// separably with fused multiply-adds -- per bin  0.25 * sum_iy (hy*R(y_lo) + ly*R(y_lo+1)),
// R(row) = sum_ix (hx*F[row][x_lo] + lx*F[row][x_lo+1])  -- instead of the reference's 16 products w_y*w_x*F added
// left to right.  The two differ by fp32 rounding only (measured max |diff| vs the CPU oracle ~5e-7 on unit-variance
// features; contract 1e-4, BASELINE.json north_star; the reference's own nvcc build contracts to FMA as well).
// The generic direct kernel in roi_align.hip keeps the reference's operation order bit for bit.
//
// Why not the reference mapping: one lane per output element issues 4*samples scattered 4-byte loads per output on a
// channel-planar tensor.  Neighbouring lanes hit different rows and columns, a wave-load touches ~14 cache lines,
// the texture-address unit serialises them (rocprofv3: 103 us at 512x256x7x7 on a 200x336 map) and every lane
// recomputes tap geometry that is identical for all 256 channels of a RoI.
//
// One 256-lane workgroup owns (RoI, 32-channel tile):
//   * window = the feature rows/columns any sample of the RoI touches.  It is copied L2 -> LDS once, by LDS-DMA
//     (buffer_load_dword ... lds): no VGPR staging, no ds_write pass, and the WHOLE window of the tile is in flight
//     at once (8 channels x ~5 pieces of 256 B per wave).  A lane's pixel index is flattened over (row, column),
//     so the LDS image is compact ([row][ww]) and every piece moves 64 useful pixels whatever the window width;
//   * LDS image: one plane per channel with an ODD plane stride, lane & 31 = channel: the 32 lanes of a half-wave
//     read the same (row, column) of 32 different planes -> 32 distinct banks, every tap pair is one conflict-free
//     ds_read2_b32, and all sampling geometry is identical across the half-wave;
//   * the 8 half-waves take different output columns pw; tap rows/columns (as LDS byte offsets) and the two weights
//     per axis sample are computed once per workgroup into two small LDS tables;
//   * clamped border samples are expressed as the pixel pair (size-1, size) with weights (1, 0), the reference's
//     (size-1, size-1) with (1, 0): the window ends one row / column past the map and the DMA reads that one from the
//     clamped address, so "high = low + 1" holds for every sample -- which is what makes the fixed +4 / +pitch tap
//     addressing possible -- and a non-finite border pixel propagates exactly as in the reference (1 * f + 0 * f);
//   * results are staged in LDS as [channel][bin] and leave as contiguous 16-byte stores.
// Windows larger than the LDS image are processed in groups of bin rows (each group: DMA, barrier, compute); output
// tiles larger than the LDS staging area likewise.  RoIs the scheme cannot serve (a sample outside the [-1, size]
// band, one bin row larger than the LDS image, > 64 samples per axis, H or W < 2) take the in-kernel direct path,
// which is the reference arithmetic bit for bit.
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
  float hw, lw;  // weight of lo and of lo + 1 (y table: already divided by the sample count)
  int lo;        // absolute row / column of the lower tap
};

// One axis of roi_align_kernel.cu:16-52 for a sample already known to lie inside the [-1, size] band.
// Returns the lower tap and the weights of (lo, lo + 1); see the header comment for the border case.
__device__ __forceinline__ void axis_taps(float v, int size, int& lo, float& hw, float& lw) {
  if (v <= 0) v = 0;
  int low = (int)v;
  if (low >= size - 1) {
    lo = size - 1;  // reference: low = high = size - 1, l = 0, h = 1; the pair is (size - 1, size), `size` read clamped
    hw = 1.f;
    lw = 0.f;
  } else {
    lo = low;
    lw = v - (float)low;
    hw = 1.f - lw;
  }
}
__device__ __forceinline__ int axis_lo(float v, int size) {
  if (v <= 0) v = 0;
  return min((int)v, size - 1);
}

// Sample coordinates of roi_align_kernel.cu:106-110 with the grid size known at compile time when kS > 0
// (x / 2.0f is evaluated as the exact x * 0.5f).
template <int kS>
__device__ __forceinline__ float coord(float start, float bin, int p, int i, int grid) {
  const float gridf = kS > 0 ? (float)kS : (float)grid;
  return start + (float)p * bin + ((float)i + .5f) * bin / gridf;
}

template <int kCap>
struct Lds {
  static constexpr int kPlane = kCap | 1;              // odd plane stride (words)
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

__device__ __forceinline__ const float* lds_at(const float* base, int byte_off) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

using lds_ptr_t = __attribute__((address_space(3))) void*;
using lds_cfloat_t = __attribute__((address_space(3))) const float*;

// 32-bit LDS byte address of a __shared__ location, made opaque to the optimiser: otherwise it re-adds the constant
// base of the image to every tap address instead of keeping (base + lane part) in one register.
__device__ __forceinline__ unsigned lds_addr_opaque(const void* p) {
  unsigned a = (unsigned)(uintptr_t)(lds_cfloat_t)p;
  asm volatile("" : "+v"(a));
  return a;
}
// the tap pair (column lo, lo + 1) at LDS byte address a: one ds_read2_b32
__device__ __forceinline__ void lds_pair(unsigned a, float& v0, float& v1) {
  const lds_cfloat_t q = (lds_cfloat_t)(uintptr_t)a;
  v0 = q[0];
  v1 = q[1];
}

// kSR > 0: sampling_ratio == kSR at compile time.  kSR == 0: run-time grid (adaptive ratio, or any other value).
// kAH > 0: aligned_height == kAH at compile time (unrolled single-pass path).
// Occupancy is bounded by LDS (3-4 workgroups = 3-4 waves per SIMD), so the register allocator is told to plan for
// that: up to 128 VGPRs let the scheduler keep several bins' LDS reads in flight.
template <int kSR, int kCap, int kAH>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
roi_align_fwd_tile(const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
                   int batch, int channels, int height, int width, int aligned_height, int aligned_width,
                   float spatial_scale, int sampling_ratio, long long* __restrict__ timeline, int ablate_arg) {
  const int ablate = MI_ABLATE(ablate_arg);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Lds<kCap> s(smem);
  constexpr int kPlane = Lds<kCap>::kPlane;
  // tuning aid (tools/timeline.py): s_memtime stamps per workgroup, null in normal operation
  auto stamp = [&](int k) {
    if (timeline != nullptr && threadIdx.x == 0) timeline[(long long)blockIdx.x * 8 + k] = (long long)clock64();
  };
  stamp(0);
  const int bins = aligned_height * aligned_width;
  const int tid = threadIdx.x;
  const int tiles = channels / kCT;
  const int r = blockIdx.x / tiles;
  const int c0 = (blockIdx.x - r * tiles) * kCT;
  float* __restrict__ dst = out + ((long long)r * channels + c0) * bins;

  // ---- RoI geometry (roi_align_kernel.cu:74-101), same fp32 operations ----
  const float* __restrict__ roi = rois + (long long)r * 5;
  const int batch_ind = (int)roi[0];
  if (batch_ind < 0 || batch_ind >= batch) {  // same guard as the direct kernel
    store_zero_tile(dst, kCT * bins, tid);
    return;
  }
  const float start_w = roi[1] * spatial_scale, start_h = roi[2] * spatial_scale;
  const float roi_width = fmaxf(roi[3] * spatial_scale - start_w, 1.f);
  const float roi_height = fmaxf(roi[4] * spatial_scale - start_h, 1.f);
  const float bin_h = roi_height / (float)aligned_height, bin_w = roi_width / (float)aligned_width;
  const int gh = kSR > 0 ? kSR : (sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)aligned_height));
  const int gw = kSR > 0 ? kSR : (sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)aligned_width));
  const float count = (float)(gh * gw);
  const int nsy = aligned_height * gh, nsx = aligned_width * gw;
  const float* __restrict__ src = feat + ((long long)batch_ind * channels + c0) * height * width;

  // ---- window: first and last sample of each axis (sample coordinates are monotonic in the sample index) ----
  const float yf = coord<kSR>(start_h, bin_h, 0, 0, gh), yl = coord<kSR>(start_h, bin_h, aligned_height - 1, gh - 1, gh);
  const float xf = coord<kSR>(start_w, bin_w, 0, 0, gw), xl = coord<kSR>(start_w, bin_w, aligned_width - 1, gw - 1, gw);
  bool fast = nsy <= kMaxS && nsx <= kMaxS && height >= 2 && width >= 2 &&
              !(yf < -1.0f || yl > (float)height || xf < -1.0f || xl > (float)width) && yl >= yf && xl >= xf;
  fast = __builtin_amdgcn_readfirstlane(fast);
  const int wy0 = __builtin_amdgcn_readfirstlane(axis_lo(yf, height));
  const int wy1 = __builtin_amdgcn_readfirstlane(axis_lo(yl, height)) + 1;
  const int wx0 = __builtin_amdgcn_readfirstlane(axis_lo(xf, width));
  const int wx1 = __builtin_amdgcn_readfirstlane(axis_lo(xl, width)) + 1;
  const int ww = wx1 - wx0 + 1, nrows = wy1 - wy0 + 1;
  const int pitch = ww * 4;                                  // bytes between two window rows in a plane
  const bool single = fast && nrows * ww <= kCap;            // the whole window fits the LDS image
  stamp(1);

  // ---- LDS-DMA of window rows [row0, row0 + nr) -> img[channel][(row - row0) * ww + col] ----
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned plane_bytes = (unsigned)height * (unsigned)width * 4u;
  // descriptor inputs through readfirstlane: the compiler must be able to PROVE them wave-uniform, otherwise it wraps
  // every buffer operation in a waterfall loop (cdna_hip_programming.md T20)
  const uintptr_t slab = reinterpret_cast<uintptr_t>(src + (long long)wave * kChPerWave * height * width);
  const uintptr_t slab_u = ((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(slab >> 32)) << 32) |
                           (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(slab & 0xffffffffu));
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<float*>(slab_u), /*stride*/ 0,
      __builtin_amdgcn_readfirstlane((int)((unsigned)kChPerWave * plane_bytes)), 0x00020000);
  // p / ww == (p * magic) >> 20 for p * ww < 2^20 (p < 2 * kCap, ww <= kCap)
  const unsigned magic = fast ? ((1u << 20) / (unsigned)ww + 1u) : 0u;
  auto issue_dma = [&](int row0, int nr) {
    const int npx = nr * ww;
    float* plane0 = s.img + wave * kChPerWave * kPlane;
    for (int k = 0; k * 64 < npx; k++) {
      const unsigned p = (unsigned)(k * 64 + lane);
      const unsigned q = (p * magic) >> 20;
      const unsigned col = p - q * (unsigned)ww;
      const unsigned voff = (min((unsigned)row0 + q, (unsigned)height - 1u) * (unsigned)width +
                             min((unsigned)wx0 + col, (unsigned)width - 1u)) * 4u;
      if (p < (unsigned)npx) {  // lanes past the window neither read memory nor write LDS
#pragma unroll
        for (int c = 0; c < kChPerWave; c++)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(plane0 + c * kPlane + k * 64), 4, voff,
                                                   c * plane_bytes, 0, 0);
      }
    }
  };
  if (single && !(ablate & 1)) issue_dma(wy0, nrows);
  stamp(2);

  // ---- tables (threads 0.. : y samples, threads 64.. : x samples) ----
  if (fast) {
    if (tid < nsy) {
      AxisEntry e;
      const int ph = kSR > 0 ? tid / (kSR > 0 ? kSR : 1) : tid / gh;
      axis_taps(coord<kSR>(start_h, bin_h, ph, tid - ph * gh, gh), height, e.lo, e.hw, e.lw);
      e.lo = min(max(e.lo, wy0), wy1 - 1);
      e.off = (e.lo - wy0) * pitch;
      // 1 / count folded into the y weights: exact for power-of-two counts
      if (kSR > 0 && ((kSR * kSR) & (kSR * kSR - 1)) == 0) {
        e.hw *= 1.f / (float)(kSR > 0 ? kSR * kSR : 1);
        e.lw *= 1.f / (float)(kSR > 0 ? kSR * kSR : 1);
      } else {
        e.hw /= count;
        e.lw /= count;
      }
      s.ty[tid] = e;
    } else if (tid >= 64 && tid - 64 < nsx) {
      const int k = tid - 64;
      AxisEntry e;
      const int pw = kSR > 0 ? k / (kSR > 0 ? kSR : 1) : k / gw;
      axis_taps(coord<kSR>(start_w, bin_w, pw, k - pw * gw, gw), width, e.lo, e.hw, e.lw);
      e.lo = min(max(e.lo, wx0), wx1 - 1);
      e.off = (e.lo - wx0) * 4;
      s.tx[k] = e;
    }
    stamp(3);
    __syncthreads();  // tables visible; the DMA of a single-pass window has landed (vmcnt(0) precedes the barrier)
    stamp(4);
    if (!single) {    // every bin row must fit the LDS image on its own
      for (int ph = 0; fast && ph < aligned_height; ph++) {
        const int lo = s.ty[ph * gh].lo, hi = s.ty[ph * gh + gh - 1].lo + 1;
        if ((hi - lo + 1) * ww > kCap) fast = false;
      }
      fast = __builtin_amdgcn_readfirstlane(fast);
    }
  }

  if (!fast) {
    // direct path for this (RoI, channel tile): reference mapping and operation order, coalesced stores
    RoiGeom g;
    g.batch_ind = batch_ind;
    g.start_w = start_w;
    g.start_h = start_h;
    g.bin_h = bin_h;
    g.bin_w = bin_w;
    g.grid_h = gh;
    g.grid_w = gw;
    g.count = count;
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
            val = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(t.w1, v1), __fmul_rn(t.w2, v2)), __fmul_rn(t.w3, v3)),
                            __fmul_rn(t.w4, v4));
          }
          output_val = __fadd_rn(output_val, val);
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
      if (!(ablate & 1)) issue_dma(row0, __builtin_amdgcn_readfirstlane(row1) - row0 + 1);
      __syncthreads();
    }
    const int base_off = (row0 - wy0) * pitch;  // table offsets are relative to wy0
    const int nb = (ph1 - ph0) * aligned_width;
    const int ts = nb | 1;

    if (ablate & 2) {
    } else if (kSR > 0) {
      constexpr int kS = kSR > 0 ? kSR : 1;
      // one output bin, separable: rows (lo, lo+1) of each y sample, column pairs (lo, lo+1) of each x sample
      auto bin_static = [&](const AxisEntry (&ey)[kS], const float (&hx)[kS], const float (&lx)[kS],
                            const unsigned (&xa)[kS]) -> float {
        float v[kS][2][kS][2];
#pragma unroll
        for (int iy = 0; iy < kS; iy++) {
#pragma unroll
          for (int ix = 0; ix < kS; ix++) {
            const unsigned a = xa[ix] + (unsigned)ey[iy].off;
            lds_pair(a, v[iy][0][ix][0], v[iy][0][ix][1]);
            lds_pair(a + (unsigned)pitch, v[iy][1][ix][0], v[iy][1][ix][1]);
          }
        }
        float acc = 0.f;
#pragma unroll
        for (int iy = 0; iy < kS; iy++) {
#pragma unroll
          for (int k = 0; k < 2; k++) {
            float rsum = hx[0] * v[iy][k][0][0];
            rsum = __builtin_fmaf(lx[0], v[iy][k][0][1], rsum);
#pragma unroll
            for (int ix = 1; ix < kS; ix++) {
              rsum = __builtin_fmaf(hx[ix], v[iy][k][ix][0], rsum);
              rsum = __builtin_fmaf(lx[ix], v[iy][k][ix][1], rsum);
            }
            acc = __builtin_fmaf(k == 0 ? ey[iy].hw : ey[iy].lw, rsum, acc);
          }
        }
        return acc;
      };

      if (kAH > 0 && ph0 == 0 && ph1 == kAH && aligned_width <= kSlots) {
        // whole RoI in one pass, one output column per half-wave: the bin-row loop is unrolled so that the LDS reads
        // of several bins overlap
        if (slot < aligned_width) {
          const int pw = slot;
          float hx[kS], lx[kS];
          unsigned xa[kS];
#pragma unroll
          for (int i = 0; i < kS; i++) {
            const AxisEntry ex = s.tx[pw * kS + i];
            hx[i] = ex.hw;
            lx[i] = ex.lw;
            xa[i] = lds_addr_opaque(lds_at(img_c, ex.off));
          }
          float res[kAH > 0 ? kAH : 1];
#pragma unroll
          for (int ph = 0; ph < kAH; ph++) {
            AxisEntry ey[kS];
#pragma unroll
            for (int i = 0; i < kS; i++) ey[i] = s.ty[ph * kS + i];
            res[ph] = bin_static(ey, hx, lx, xa);
          }
#pragma unroll
          for (int ph = 0; ph < kAH; ph++) s.tile[cl * ts + ph * aligned_width + pw] = res[ph];
        }
      } else {
        for (int pw = slot; pw < aligned_width; pw += kSlots) {
          float hx[kS], lx[kS];
          unsigned xa[kS];
#pragma unroll
          for (int i = 0; i < kS; i++) {
            const AxisEntry ex = s.tx[pw * kS + i];
            hx[i] = ex.hw;
            lx[i] = ex.lw;
            xa[i] = lds_addr_opaque(lds_at(img_c, ex.off - base_off));
          }
          for (int ph = ph0; ph < ph1; ph++) {
            AxisEntry ey[kS];
#pragma unroll
            for (int i = 0; i < kS; i++) ey[i] = s.ty[ph * kS + i];
            s.tile[cl * ts + (ph - ph0) * aligned_width + pw] = bin_static(ey, hx, lx, xa);
          }
        }
      }
    } else {
      for (int pw = slot; pw < aligned_width; pw += kSlots) {
        for (int ph = ph0; ph < ph1; ph++) {
          float acc = 0.f;
          for (int iy = 0; iy < gh; iy++) {
            const AxisEntry ey = s.ty[ph * gh + iy];
            float r0 = 0.f, r1 = 0.f;
            for (int ix = 0; ix < gw; ix++) {
              const AxisEntry ex = s.tx[pw * gw + ix];
              const float* a = lds_at(img_c, ey.off + ex.off - base_off);
              const float* b = lds_at(a, pitch);
              r0 = __builtin_fmaf(ex.hw, a[0], r0);
              r0 = __builtin_fmaf(ex.lw, a[1], r0);
              r1 = __builtin_fmaf(ex.hw, b[0], r1);
              r1 = __builtin_fmaf(ex.lw, b[1], r1);
            }
            acc = __builtin_fmaf(ey.hw, r0, acc);
            acc = __builtin_fmaf(ey.lw, r1, acc);
          }
          s.tile[cl * ts + (ph - ph0) * aligned_width + pw] = acc;
        }
      }
    }
    stamp(5);
    __syncthreads();
    stamp(6);

    // ---- staged outputs -> HBM ----
    float* gdst = dst + ph0 * aligned_width;
    if (ablate & 4) {
    } else if (nb == bins && ts == nb && ((kCT * nb) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
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
  stamp(7);
}

long long* g_timeline = nullptr;

template <int kCap>
int launch_cap(const float* features, const float* rois, float* output, int batch, int channels, int height,
               int width, int num_rois, int aligned_height, int aligned_width, float spatial_scale,
               int sampling_ratio, hipStream_t stream) {
  const int grid = num_rois * (channels / kCT);
  const size_t lds = Lds<kCap>::bytes();
  if (sampling_ratio == 2 && aligned_height == 7)
    roi_align_fwd_tile<2, kCap, 7><<<grid, kThreads, lds, stream>>>(features, rois, output, batch, channels, height,
                                                                    width, aligned_height, aligned_width,
                                                                    spatial_scale, sampling_ratio, g_timeline, tuning().ablate);
  else if (sampling_ratio == 2)
    roi_align_fwd_tile<2, kCap, 0><<<grid, kThreads, lds, stream>>>(features, rois, output, batch, channels, height,
                                                                    width, aligned_height, aligned_width,
                                                                    spatial_scale, sampling_ratio, g_timeline, tuning().ablate);
  else
    roi_align_fwd_tile<0, kCap, 0><<<grid, kThreads, lds, stream>>>(features, rois, output, batch, channels, height,
                                                                    width, aligned_height, aligned_width,
                                                                    spatial_scale, sampling_ratio, g_timeline, tuning().ablate);
  return check_launch("roi_align_fwd_tile");
}

}  // namespace

void roi_align_fwd_tile_set_timeline(long long* device_buffer) { g_timeline = device_buffer; }

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
