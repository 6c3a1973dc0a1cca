// roi_align_fwd_tile.hip -- the NCHW fast path of RoIAlign forward (Caffe2 semantics) for gfx950.
//
// Arithmetic: identical, operation for operation, to roi_align_fwd_direct in roi_align.hip
// (reference: lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu:16-121); outputs are bit-equal.
// What changes is where the data moves and how many instructions a tap costs.
//
// The reference mapping (one lane per output element) issues 4*samples scattered 4-byte loads per
// output on a channel-planar tensor -- neighbouring lanes hit different rows/columns, nothing
// coalesces -- and recomputes the tap geometry (float->int, clamps, 4 weight products) for every
// one of them.  rocprofv3 PMC on MI355X shows both this mapping and a naive LDS version VALU-bound
// (SQ_ACTIVE_INST_VALU ~ half of the kernel time), so the design below minimises vector
// instructions per tap as much as bytes per tap.
//
// One 256-lane workgroup owns (RoI, 32-channel tile) and moves the RoI's feature window through LDS
// exactly once:
//   * the window [wy0..wy1] x [wx0..wx1] lives in an LDS ring of `nr = kCap / ww` rows per channel
//     (ww = window width <= 32).  Small windows are resident after one prologue fill; taller ones
//     stream top to bottom: each bin-row iteration first ISSUES the global loads of the rows the
//     ring can take next (32-bit offsets off one scalar base, no 64-bit vector address math), then
//     computes one row of output bins from rows that are already resident, and only then parks
//     the loaded registers in LDS, so HBM/L2 latency hides under the arithmetic; one workgroup
//     barrier per bin row while streaming, none once the window is resident;
//   * row segments are loaded 32 lanes wide (coalesced 128-byte runs);
//   * a lane owns one channel (lane & 31); the two 32-lane halves of each wavefront take different
//     output columns.  The per-channel plane stride (kCap + 1 words) is odd, so the 32 lanes of a
//     half always hit 32 distinct LDS banks: every bilinear tap is a conflict-free ds_read_b32 at
//     lane_base + row_bytes + col_bytes (one v_add per tap);
//   * tap rows/columns (as LDS byte offsets) and weights are computed once per workgroup into
//     small LDS tables instead of 4*samples times per output element;
//   * results are staged in LDS as [channel][bin] (odd stride) and leave as one contiguous run.
// RoIs the ring cannot serve (window wider than 32 columns, a bin row taller than the ring, more
// than 64 samples per axis) take the in-kernel direct path; results are identical.
#include "common.h"
#include "roi_align_device.h"

namespace mi {
namespace {

constexpr int kCT = 32;          // channels per workgroup
constexpr int kMaxWW = 32;       // widest window on the ring path (one 32-lane segment per row)
constexpr int kMaxS = 64;        // samples per axis the tables hold
constexpr int kThreads = 256;
constexpr int kSlots = kThreads / 32;   // half-waves
constexpr int kChPerSlot = kCT / kSlots;
constexpr int kPF = 8;           // max rows fetched per lane and channel in one batch

struct AxisEntry {
  int lo, hi;    // BYTE offsets: y table -> ring row start inside a channel plane, x table -> column inside a row
  float hw, lw;  // weight of lo (1 - frac) and of hi (frac); hw < 0 marks a sample outside the [-1, size] band
};

struct AxisRaw {
  int lo, hi;  // absolute row / column, -1 when the sample is outside the band
  float hw, lw;
};

// One axis of roi_align_kernel.cu:16-52 (the y and x halves of bilinear_interpolate are independent).
__device__ __forceinline__ AxisRaw axis_raw(float v, int size) {
  AxisRaw e;
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

template <int kCap>
struct Lds {
  static constexpr int kPlane = kCap + 1;  // odd
  float* ring;    // [kCT][kPlane]
  float* tile;    // [kCT][os]
  AxisEntry* ty;  // [kMaxS]
  AxisEntry* tx;  // [kMaxS]
  int* band_lo;   // [kMaxS] first / last absolute feature row each output-bin row reads (hi = -1: none)
  int* band_hi;
  int* misc;      // wx0, wx1, wy0, wy1, invalid-sample count
  __device__ __forceinline__ Lds(float* smem, int os) {
    ring = smem;
    tile = ring + kCT * kPlane;
    float* p = tile + kCT * os;
    p += (4 - ((kCT * kPlane + kCT * os) & 3)) & 3;  // 16-byte align the tables
    ty = reinterpret_cast<AxisEntry*>(p);
    tx = ty + kMaxS;
    band_lo = reinterpret_cast<int*>(tx + kMaxS);
    band_hi = band_lo + kMaxS;
    misc = band_hi + kMaxS;
  }
  static size_t bytes(int bins) {
    size_t words = (size_t)kCT * kPlane + (size_t)kCT * (bins | 1);
    words = (words + 3) & ~size_t(3);
    return words * 4 + 2 * kMaxS * sizeof(AxisEntry) + 2 * kMaxS * 4 + 32;
  }
};

__device__ __forceinline__ void store_zero_tile(float* dst, int n, int tid) {
  for (int i = tid; i < n; i += kThreads) dst[i] = 0.f;
}

__device__ __forceinline__ float lds_at(const float* base, int byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

// N row loads per channel of this lane through a raw buffer descriptor: the per-lane part of the address
// (column, channel slot) is one 32-bit voffset computed once per workgroup, the per-row / per-channel part is
// a scalar soffset -- no vector address arithmetic per load.
template <int N>
__device__ __forceinline__ void issue_rows(float (&pf)[kChPerSlot][kPF], __amdgpu_buffer_rsrc_t rsrc,
                                           unsigned voff_bytes, unsigned row0_bytes, unsigned plane_step_bytes,
                                           unsigned width_bytes) {
#pragma unroll
  for (int cc = 0; cc < kChPerSlot; cc++) {
#pragma unroll
    for (int k = 0; k < N; k++)
      pf[cc][k] = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_bytes,
                                                      row0_bytes + cc * plane_step_bytes + k * width_bytes, 0));
  }
}

template <int kPlane, int N>
__device__ __forceinline__ void park_rows(const float (&pf)[kChPerSlot][kPF], float* lrow, int next_slot, int nr,
                                          int ww, bool loader) {
  if (!loader) return;  // only the stores are predicated
#pragma unroll
  for (int k = 0; k < N; k++) {
    int rr = next_slot + k;
    if (rr >= nr) rr -= nr;
    float* p = lrow + rr * ww;
#pragma unroll
    for (int cc = 0; cc < kChPerSlot; cc++) p[cc * kSlots * kPlane] = pf[cc][k];
  }
}

#define MI_SWITCH_N(n, STMT)                      \
  switch (n) {                                    \
    case 1: { constexpr int N = 1; STMT; } break; \
    case 2: { constexpr int N = 2; STMT; } break; \
    case 3: { constexpr int N = 3; STMT; } break; \
    case 4: { constexpr int N = 4; STMT; } break; \
    case 5: { constexpr int N = 5; STMT; } break; \
    case 6: { constexpr int N = 6; STMT; } break; \
    case 7: { constexpr int N = 7; STMT; } break; \
    case 8: { constexpr int N = 8; STMT; } break; \
    default: break;                               \
  }

// kSR > 0: sampling_ratio == kSR at compile time (sample loops unrolled; all 4*kSR*kSR tap reads of a bin
// are in flight before the first use).  kSR == 0: run-time grid (adaptive ratio, or any other value).
template <int kSR, int kCap>
__global__ void __launch_bounds__(kThreads)
roi_align_fwd_tile(const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
                   int batch, int channels, int height, int width, int aligned_height, int aligned_width,
                   float spatial_scale, int sampling_ratio, unsigned bins_magic) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bins = aligned_height * aligned_width;
  const int os = bins | 1;
  const Lds<kCap> s(smem, os);
  constexpr int kPlane = Lds<kCap>::kPlane;
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
  // wave-uniform values that the compiler cannot prove uniform go through readfirstlane (scalar control flow,
  // SGPR buffer descriptor: cdna_hip_programming.md T20)
  const int batch_ind = __builtin_amdgcn_readfirstlane(g.batch_ind);
  const float* __restrict__ src = feat + ((long long)batch_ind * channels + c0) * height * width;
  const int gh = kSR > 0 ? kSR : g.grid_h, gw = kSR > 0 ? kSR : g.grid_w;
  const int nsy = aligned_height * gh, nsx = aligned_width * gw;

  // ---- tables, pass 1: raw taps, window and per-bin-row extents ------------------------------------
  bool fast = nsy <= kMaxS && nsx <= kMaxS;  // uniform
  AxisRaw raw;
  raw.lo = raw.hi = -1;
  raw.hw = raw.lw = 0.f;
  if (fast) {
    if (tid < kMaxS) {
      s.band_lo[tid] = 0x7fffffff;
      s.band_hi[tid] = -1;
    }
    if (tid == 0) {
      s.misc[0] = 0x7fffffff;
      s.misc[1] = -1;
      s.misc[2] = 0x7fffffff;
      s.misc[3] = -1;
      s.misc[4] = 0;
    }
    __syncthreads();
    if (tid < nsy) {
      raw = axis_raw(sample_y(g, tid / gh, tid % gh), height);
      if (raw.lo >= 0) {
        atomicMin(&s.band_lo[tid / gh], raw.lo);
        atomicMax(&s.band_hi[tid / gh], raw.hi);
        atomicMin(&s.misc[2], raw.lo);
        atomicMax(&s.misc[3], raw.hi);
      } else {
        atomicAdd(&s.misc[4], 1);
      }
    } else if (tid >= 64 && tid - 64 < nsx) {
      const int k = tid - 64;
      raw = axis_raw(sample_x(g, k / gw, k % gw), width);
      if (raw.lo >= 0) {
        atomicMin(&s.misc[0], raw.lo);
        atomicMax(&s.misc[1], raw.hi);
      } else {
        atomicAdd(&s.misc[4], 1);
      }
    }
    __syncthreads();
  }
  const int wx0 = fast ? __builtin_amdgcn_readfirstlane(s.misc[0]) : 0;
  const int wx1 = fast ? __builtin_amdgcn_readfirstlane(s.misc[1]) : -1;
  const int wy0 = fast ? __builtin_amdgcn_readfirstlane(s.misc[2]) : 0;
  const int wy1 = fast ? __builtin_amdgcn_readfirstlane(s.misc[3]) : -1;
  // no sample of this RoI is outside the band
  const bool all_valid = fast && __builtin_amdgcn_readfirstlane(s.misc[4]) == 0;
  const int ww = wx1 - wx0 + 1;
  const bool any = wx1 >= 0 && wy1 >= 0;               // some sample lands inside the band
  const int nr = any && ww <= kMaxWW ? kCap / ww : 0;  // ring rows
  if (fast && any) {
    fast = ww <= kMaxWW;
    for (int ph = 0; fast && ph < aligned_height; ph++) {
      const int lo = __builtin_amdgcn_readfirstlane(s.band_lo[ph]), hi = __builtin_amdgcn_readfirstlane(s.band_hi[ph]);
      if (hi >= 0 && hi - lo + 1 > nr) fast = false;
    }
  }

  if (!fast) {
    // direct path for this (RoI, channel tile): reference mapping, coalesced stores
    for (int i = tid; i < kCT * bins; i += kThreads) {
      const int c = (int)__umulhi((unsigned)i, bins_magic), bin = i - c * bins;
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
  if (!any) {  // every sample outside the band: all outputs are 0 / count = 0
    store_zero_tile(dst, kCT * bins, tid);
    return;
  }

  // ---- tables, pass 2: LDS byte offsets (invalid samples point at word 0 and carry hw = -1) -------
  if (tid < nsy) {
    AxisEntry e;
    const bool ok = raw.lo >= 0;
    const int slot_lo = ok ? (raw.lo - wy0) % nr : 0;
    int slot_hi = slot_lo + (raw.hi - raw.lo);  // hi is lo or lo + 1
    if (slot_hi >= nr) slot_hi -= nr;
    e.lo = slot_lo * ww * 4;
    e.hi = ok ? slot_hi * ww * 4 : 0;
    e.hw = ok ? raw.hw : -1.f;
    e.lw = raw.lw;
    s.ty[tid] = e;
  } else if (tid >= 64 && tid - 64 < nsx) {
    AxisEntry e;
    const bool ok = raw.lo >= 0;
    e.lo = ok ? (raw.lo - wx0) * 4 : 0;
    e.hi = ok ? (raw.hi - wx0) * 4 : 0;
    e.hw = ok ? raw.hw : -1.f;
    e.lw = raw.lw;
    s.tx[tid - 64] = e;
  }
  // (visibility of the tables is covered by the barrier after the prologue fill)

  const int lx = tid & 31, slot = tid >> 5;
  const int cl = tid & 31;  // channel of this lane in the compute phase
  const float* ring_c = s.ring + cl * kPlane;
  const bool loader = lx < ww;
  const unsigned width_bytes = (unsigned)width * 4u;
  const unsigned plane_step_bytes = (unsigned)kSlots * (unsigned)height * width_bytes;
  const unsigned voff_bytes =
      loader ? (unsigned)slot * (unsigned)height * width_bytes + (unsigned)(wx0 + lx) * 4u : 0xffffff00u;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(src), /*stride*/ 0, (int)((unsigned)kCT * (unsigned)height * width_bytes), 0x00020000);
  float* lrow = s.ring + slot * kPlane + lx;  // + cc*kSlots*kPlane, + ringrow*ww

  int resident_hi = wy0 - 1;  // rows <= resident_hi are in the ring (after the next barrier)
  int next_slot = 0;          // ring row that row resident_hi + 1 goes to
  float pf[kChPerSlot][kPF];

  // Control flow around the loads is wave-uniform (scalar branches, no per-lane predication): lanes right of the
  // window carry an out-of-range voffset, which the buffer bounds check answers with 0 without touching memory.
  auto issue = [&](int n) {  // loads of rows (resident_hi, resident_hi + n] -> pf
    const unsigned row0_bytes = (unsigned)(resident_hi + 1) * width_bytes;
    MI_SWITCH_N(n, (issue_rows<N>(pf, rsrc, voff_bytes, row0_bytes, plane_step_bytes, width_bytes)))
  };
  auto park = [&](int n) {  // pf -> ring; advance resident_hi / next_slot
    MI_SWITCH_N(n, (park_rows<kPlane, N>(pf, lrow, next_slot, nr, ww, loader)))
    resident_hi += n;
    next_slot += n;
    if (next_slot >= nr) next_slot -= nr;
  };

  // one row of output bins (all of whose feature rows are resident) -> s.tile
  auto compute_bin_row = [&](int ph) {
    for (int pw = slot; pw < aligned_width; pw += kSlots) {
      float output_val = 0.f;
      if (kSR > 0) {
        constexpr int kS = kSR > 0 ? kSR : 1;
        AxisEntry ey[kS], ex[kS];
#pragma unroll
        for (int i = 0; i < kS; i++) {
          ey[i] = s.ty[ph * kS + i];
          ex[i] = s.tx[pw * kS + i];
        }
        float v[kS][kS][4];
#pragma unroll
        for (int iy = 0; iy < kS; iy++) {
          const float* ra = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ring_c) + ey[iy].lo);
          const float* rb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ring_c) + ey[iy].hi);
#pragma unroll
          for (int ix = 0; ix < kS; ix++) {
            v[iy][ix][0] = lds_at(ra, ex[ix].lo);
            v[iy][ix][1] = lds_at(ra, ex[ix].hi);
            v[iy][ix][2] = lds_at(rb, ex[ix].lo);
            v[iy][ix][3] = lds_at(rb, ex[ix].hi);
          }
        }
#pragma unroll
        for (int iy = 0; iy < kS; iy++) {
#pragma unroll
          for (int ix = 0; ix < kS; ix++) {
            const float w1 = ey[iy].hw * ex[ix].hw, w2 = ey[iy].hw * ex[ix].lw;
            const float w3 = ey[iy].lw * ex[ix].hw, w4 = ey[iy].lw * ex[ix].lw;
            float val = (w1 * v[iy][ix][0] + w2 * v[iy][ix][1] + w3 * v[iy][ix][2] + w4 * v[iy][ix][3]);
            if (!all_valid)  // uniform branch; roi_align_kernel.cu:19-22: outside the band -> 0
              val = (ey[iy].hw >= 0.f && ex[ix].hw >= 0.f) ? val : 0.f;
            output_val += val;
          }
        }
        constexpr float kInvCount = 1.f / (float)(kS * kS);
        // count = kSR^2: for a power of two the reciprocal multiply is exact and equals the division bit for bit
        output_val = ((kS & (kS - 1)) == 0) ? output_val * kInvCount : output_val / g.count;
      } else {
        for (int iy = 0; iy < gh; iy++) {
          const AxisEntry ey = s.ty[ph * gh + iy];
          for (int ix = 0; ix < gw; ix++) {
            const AxisEntry ex = s.tx[pw * gw + ix];
            float val = 0.f;
            if (ey.hw >= 0.f && ex.hw >= 0.f) {
              const float* ra = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ring_c) + ey.lo);
              const float* rb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ring_c) + ey.hi);
              const float v1 = lds_at(ra, ex.lo), v2 = lds_at(ra, ex.hi);
              const float v3 = lds_at(rb, ex.lo), v4 = lds_at(rb, ex.hi);
              const float w1 = ey.hw * ex.hw, w2 = ey.hw * ex.lw, w3 = ey.lw * ex.hw, w4 = ey.lw * ex.lw;
              val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);  // roi_align_kernel.cu:58-60
            }
            output_val += val;  // :113
          }
        }
        output_val = output_val / g.count;  // :117
      }
      s.tile[cl * os + ph * aligned_width + pw] = output_val;
    }
  };

  // ---- prologue: fill the ring (the whole window when it fits) ---------------------------------
  {
    const int target = min(wy1, wy0 + nr - 1);
    while (resident_hi < target) {
      const int n = min(kPF, target - resident_hi);
      issue(n);
      park(n);
    }
  }
  __syncthreads();

  for (int ph = 0; ph < aligned_height; ph++) {
    const int ya = __builtin_amdgcn_readfirstlane(s.band_lo[ph]);
    const int yb = __builtin_amdgcn_readfirstlane(s.band_hi[ph]);
    const bool more = resident_hi < wy1;  // uniform: the window is still streaming
    int n = 0;
    if (yb >= 0) {
      if (yb > resident_hi) {
        // rare: this bin row and the previous one do not fit the ring together -> synchronous refill
        // (every wave is past the barrier that ended the previous iteration, so its reads are done)
        if (resident_hi < ya - 1) {  // rows between two bin rows that nobody reads are skipped
          resident_hi = ya - 1;
          next_slot = (ya - wy0) % nr;
        }
        while (resident_hi < yb) {
          const int m = min(kPF, yb - resident_hi);
          issue(m);
          park(m);
        }
        __syncthreads();
      }
      if (more) {
        // rows the ring can take without touching a row >= ya; at most kPF per iteration
        n = __builtin_amdgcn_readfirstlane(max(min(min(wy1, ya + nr - 1) - resident_hi, kPF), 0));
        issue(n);
      }
      compute_bin_row(ph);
    } else {
      for (int pw = slot; pw < aligned_width; pw += kSlots) s.tile[cl * os + ph * aligned_width + pw] = 0.f;
    }
    if (more) {
      park(n);
      __syncthreads();
    }
  }
  __syncthreads();
  for (int i = tid; i < kCT * bins; i += kThreads) {
    const int c = (int)__umulhi((unsigned)i, bins_magic), bin = i - c * bins;
    dst[i] = s.tile[c * os + bin];
  }
}

template <int kCap>
int launch_cap(const float* features, const float* rois, float* output, int batch, int channels, int height,
               int width, int num_rois, int aligned_height, int aligned_width, float spatial_scale,
               int sampling_ratio, hipStream_t stream) {
  const int bins = aligned_height * aligned_width;
  const int grid = num_rois * (channels / kCT);
  const size_t lds = Lds<kCap>::bytes(bins);
  // c = floor(i / bins) for 0 <= i < kCT * bins via one mul-hi: magic = ceil(2^32 / bins)
  const unsigned magic = (unsigned)(((1ULL << 32) + bins - 1) / bins);
  if (sampling_ratio == 2)
    roi_align_fwd_tile<2, kCap><<<grid, kThreads, lds, stream>>>(features, rois, output, batch, channels, height,
                                                                 width, aligned_height, aligned_width,
                                                                 spatial_scale, sampling_ratio, magic);
  else
    roi_align_fwd_tile<0, kCap><<<grid, kThreads, lds, stream>>>(features, rois, output, batch, channels, height,
                                                                 width, aligned_height, aligned_width,
                                                                 spatial_scale, sampling_ratio, magic);
  return check_launch("roi_align_fwd_tile");
}

}  // namespace

bool roi_align_fwd_tile_supported(int channels, int height, int width, int aligned_height, int aligned_width) {
  const int bins = aligned_height * aligned_width;
  // 32-bit element offsets inside one (image, 32-channel) slab; LDS budget of the smallest ring
  return channels > 0 && channels % kCT == 0 && bins <= 2048 && (long long)kCT * height * width < (1LL << 31) &&
         Lds<192>::bytes(bins) <= 64 * 1024;
}

int launch_roi_align_fwd_tile(const float* features, const float* rois, float* output, int batch, int channels,
                              int height, int width, int num_rois, int aligned_height, int aligned_width,
                              float spatial_scale, int sampling_ratio, int ring_words, hipStream_t stream) {
  const int bins = aligned_height * aligned_width;
  if (ring_words >= 320 && Lds<320>::bytes(bins) <= 64 * 1024)
    return launch_cap<320>(features, rois, output, batch, channels, height, width, num_rois, aligned_height,
                           aligned_width, spatial_scale, sampling_ratio, stream);
  if (ring_words >= 256 && Lds<256>::bytes(bins) <= 64 * 1024)
    return launch_cap<256>(features, rois, output, batch, channels, height, width, num_rois, aligned_height,
                           aligned_width, spatial_scale, sampling_ratio, stream);
  return launch_cap<192>(features, rois, output, batch, channels, height, width, num_rois, aligned_height,
                         aligned_width, spatial_scale, sampling_ratio, stream);
}

}  // namespace mi
