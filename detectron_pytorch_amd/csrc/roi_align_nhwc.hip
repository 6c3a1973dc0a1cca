// roi_align_nhwc.hip -- RoIAlign forward (Caffe2 semantics) over features stored channels-last ([N][H][W][C], what
// MIOpen's NHWC convolutions produce on gfx950), output in the reference's dense [R][C][PH][PW].
//
// Why a separate kernel.  In NHWC one pixel of a 256-channel map is 1 KB of consecutive memory, so with lane = channel
// group a bilinear tap is one fully coalesced wave load (64 lanes x 16 B) and -- unlike NCHW, where a window row of
// ~17 px drags in 1.5 cache lines per channel -- no byte is fetched that is not used.  The window does not have to be
// staged in LDS at all: the tap positions and the interpolation weights are identical for every lane of the wave, so
// they live in SGPRs (scalar loads from the RoI's record, see roi_align_record_layout.h) and the taps go straight
// from L1/L2 into the FMAs.
//
//   workgroup = (RoI at sweep rank `pos`, chunk of 64*V channels), wave = bin row ph (ph, ph + nwaves, ...).
//   For every feature row the bin row touches (the y samples of the bin row are merged per row: a row that is the
//   lower tap of one sample and the upper tap of another is read once, with the summed weight) and every output
//   column pw:   S = sum_ix ( hx * F[row][xlo] + lx * F[row][xlo + 1] ),   acc[pw] += wy(row) * S      (FMAs)
//   -- the separable evaluation of roi_align_fwd_records (fp32 rounding differences only w.r.t. the reference's
//   summation order; contract 1e-4, asserted 1e-5).  Border samples come from the record as the pair
//   (size-1, size) with weights (1, 0); the upper tap's address is clamped to size-1 (the reference reads the border
//   pixel twice, 1 * f + 0 * f).
//   The 64*V x PH x PW results of the workgroup are transposed through LDS ([k][lane][bin] with an odd bin stride:
//   conflict-free) and leave as one contiguous run of 16-byte stores (the chunk's output is contiguous in [R][C][PH][PW]).
//
// RoIs whose samples the tables cannot describe (roi_align_prepare clears kFlagTabs: samples outside [-1, size],
// more than 32 samples per axis, 1-pixel maps) are evaluated in the same kernel with the reference's operation order
// (roi_align_kernel.cu:16-63,103-117; bit-exact); RoIs of a non-existent image give zeros.
//
// Workgroups are dealt so that each XCD (blockIdx % 8) owns one contiguous eighth of the sweep: the windows of
// neighbouring RoIs overlap 2.2x on the config-2 input and then meet in one L2.
#include "common.h"
#include "roi_align_device.h"
#include "roi_align_record_layout.h"

namespace mi {
namespace {

using const_int_ptr = const __attribute__((address_space(4))) int*;
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// One tap = one buffer load: the image's descriptor and the byte offset of the pixel are wave-uniform (SGPRs), the
// lane contributes only its channel offset -- no per-tap address arithmetic on the vector unit.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <int V>
__device__ __forceinline__ void load_tap(__amdgpu_buffer_rsrc_t img, unsigned lane_bytes, int pixel_bytes, float (&d)[V]);
template <>
__device__ __forceinline__ void load_tap<4>(__amdgpu_buffer_rsrc_t img, unsigned lane_bytes, int pixel_bytes,
                                            float (&d)[4]) {
  const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(img, (int)lane_bytes, pixel_bytes, 0));
  d[0] = v.x;
  d[1] = v.y;
  d[2] = v.z;
  d[3] = v.w;
}
template <>
__device__ __forceinline__ void load_tap<2>(__amdgpu_buffer_rsrc_t img, unsigned lane_bytes, int pixel_bytes,
                                            float (&d)[2]) {
  const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(img, (int)lane_bytes, pixel_bytes, 0));
  d[0] = v.x;
  d[1] = v.y;
}
template <>
__device__ __forceinline__ void load_tap<1>(__amdgpu_buffer_rsrc_t img, unsigned lane_bytes, int pixel_bytes,
                                            float (&d)[1]) {
  d[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(img, (int)lane_bytes, pixel_bytes, 0));
}

// kSR > 0: sampling_ratio == kSR at compile time.  PW: aligned_width.  V: channels per lane.  PB: output columns
// whose taps are in flight together (kSR > 0 only): all 2 * kSR * PB loads of a feature row are issued before the
// first FMA -- written as two phases because hipcc otherwise waits for each column's taps before issuing the next
// column's, and a wave's life becomes a chain of L2 latencies.
//
// Record access.  kSR > 0: the header, this bin row's y entries and the whole x table are fetched with three vector
// loads issued together (lane i holds dword i) and picked apart with v_readlane -- ONE memory latency at the head of
// the wave instead of a chain of dependent scalar loads (header -> y entries -> x entries, each a K-cache miss on
// records that roi_align_prepare has just written).  kSR == 0 (adaptive grid, not a hot configuration): scalar loads.
template <int kSR, int PW, int V, int PB>
__global__ void __launch_bounds__(PW <= 7 ? 448 : 1024)
    __attribute__((amdgpu_waves_per_eu(PW <= 7 && PB <= 4 ? 4 : 1, PW <= 7 && PB <= 4 ? 4 : 8)))
roi_align_fwd_nhwc(const float* __restrict__ rois, float* __restrict__ out, const int* __restrict__ ws, int batch,
                   int channels, int aligned_height, int sampling_ratio, int chunks, int tile_stride, int order_mul,
                   int zigzag, const LevelTable lv,  // scalars first: they arrive preloaded in SGPRs (build.py)
                   long long* __restrict__ timeline) {
  extern __shared__ float tile[];  // [V][64][tile_stride]
  // tuning builds only (tools/timeline_nhwc.py): clock stamps of wave 0 of every workgroup; the release kernel has none
  const auto stamp = [&](int k) {
#if MI_TUNING
    if (timeline != nullptr && threadIdx.x == 0) timeline[(long long)blockIdx.x * 8 + k] = (long long)clock64();
#else
    (void)k;
#endif
  };
  stamp(0);
  const int lane = threadIdx.x & 63;
  const int wave = uniform((int)(threadIdx.x >> 6)), nwaves = (int)(blockDim.x >> 6);
  // item of this workgroup: XCD x takes the x-th contiguous eighth of the (rank, chunk) items
  int item;
  {
    const int total = (int)gridDim.x, x = (int)(blockIdx.x & 7), idx = (int)(blockIdx.x >> 3);
    const int q = total >> 3, rem = total & 7;
    item = x * q + min(x, rem) + idx;
  }
  int pos = item / chunks;
  const int chunk = item - pos * chunks;
  if (order_mul != 1) pos = (int)(((long long)pos * order_mul + 11) % ((int)gridDim.x / chunks));  // ablation: no sweep
  const int c0 = chunk * 64 * V;
  const int* __restrict__ recg = ws + kCounterDwords + (long long)pos * kRecDwords;
  const const_int_ptr rec = (const_int_ptr)(uintptr_t)recg;
  int hdr_v = 0, x_v0 = 0, x_v1 = 0;
  if constexpr (kSR > 0) {
    hdr_v = recg[lane < kRecHeader ? lane : 0];
    x_v0 = recg[kRecX + lane];
    if constexpr (4 * PW * kSR > 64) x_v1 = recg[kRecX + 64 + lane];
  }
  const auto hdr = [&](int i) { return kSR > 0 ? __builtin_amdgcn_readlane(hdr_v, i) : rec[i]; };
  const auto xtab = [&](int i) {  // dword i of the x table
    if constexpr (kSR > 0) return i < 64 ? __builtin_amdgcn_readlane(x_v0, i) : __builtin_amdgcn_readlane(x_v1, i - 64);
    else return rec[kRecX + i];
  };
  const int flags = hdr(0), batch_ind = hdr(1), r = hdr(8), lvl = hdr(11);
  // the (channels-last) map of the RoI's level: one entry unless the call is an FPN-fused one
  const float* __restrict__ feat = lv.feat[lvl];
  const int height = lv.height[lvl], width = lv.width[lvl];
  const float spatial_scale = lv.scale[lvl];
  const int gh = kSR > 0 ? kSR : rec[6], gw = kSR > 0 ? kSR : rec[7];
  if (flags >= 0) stamp(1);  // record words have arrived
#if MI_TUNING
  if (timeline != nullptr && threadIdx.x == 0) {  // where the workgroup ran (XCC_ID, HW_ID) and which RoI it pooled
    timeline[(long long)blockIdx.x * 8 + 6] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                              (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    timeline[(long long)blockIdx.x * 8 + 7] = r;
  }
#endif
  const int cl = c0 + lane * V;
  // lanes past the last channel read channel 0 (valid memory); their results are never copied out
  const unsigned lane_off = (unsigned)(cl < channels ? cl : 0) * 4u;  // bytes
  // descriptor of the RoI's image; its inputs go through readfirstlane so that hipcc can prove them wave-uniform
  // (otherwise every buffer load is wrapped in a waterfall loop)
  const int img_ind = min(max(batch_ind, 0), batch - 1);
  const unsigned img_bytes = (unsigned)height * (unsigned)width * (unsigned)channels * 4u;
  const uintptr_t img_addr = reinterpret_cast<uintptr_t>(feat) + (uintptr_t)img_ind * img_bytes;
  const uintptr_t img_addr_u = ((uintptr_t)(unsigned)uniform((int)(img_addr >> 32)) << 32) |
                               (uintptr_t)(unsigned)uniform((int)(img_addr & 0xffffffffu));
  const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(img_addr_u), 0,
                                                                        uniform((int)img_bytes), 0x00020000);
  const int pixel_bytes = channels * 4;
  const int bins = aligned_height * PW;
  for (int ph = wave; ph < aligned_height; ph += nwaves) {
    float acc[PW][V];
#pragma unroll
    for (int pw = 0; pw < PW; pw++)
#pragma unroll
      for (int k = 0; k < V; k++) acc[pw][k] = 0.f;

    int y_v = 0;
    if constexpr (kSR > 0) y_v = recg[kRecY + 4 * kSR * ph + (lane < 4 * kSR ? lane : 0)];
    const auto ytab = [&](int i) {  // dword i of this bin row's y entries
      return kSR > 0 ? __builtin_amdgcn_readlane(y_v, i) : rec[kRecY + 4 * ph * gh + i];
    };
    if (flags & kFlagTabs) {
      const int row_first = ytab(3);
      const int row_last = ytab(4 * (gh - 1) + 3) + 1;
      // Neighbouring bin rows share their boundary feature row.  Even bin rows walk their rows downwards, odd ones
      // upwards, so the two waves that need a shared row ask for it in the same step of their walk and the second
      // request finds it in L1 (seven waves x 18 KB per step thrash a CU's L1 otherwise: 194 MB L2->L1 for 148 MB
      // of window pixels).  MI_ROI_ALIGN_NHWC_ZIGZAG=0 walks every bin row downwards.
      const int nrows = row_last - row_first + 1;
      // the column part of every tap's address does not depend on the feature row: byte offset of the lower tap and the
      // step to the upper one (0 at the map's last column: a border sample reads the border pixel twice), once per bin row
      // in SGPRs instead of a v_readlane + six scalar operations per tap pair and row
      constexpr int kNX = kSR > 0 ? PW * kSR : 1;
      int tap_lo[kNX], tap_up[kNX];
      if constexpr (kSR > 0) {
#pragma unroll
        for (int j = 0; j < kNX; j++) {
          const int xlo = xtab(4 * j + 3);
          tap_lo[j] = xlo * pixel_bytes;
          tap_up[j] = xlo + 1 < width ? pixel_bytes : 0;
        }
      }
      for (int step = 0; step < nrows; step++) {
        const int row = (zigzag && (ph & 1)) ? row_last - step : row_first + step;
        // weight of this feature row in the bin row: sum over the y samples that tap it
        float wy = 0.f;
        bool touched = false;
#pragma unroll
        for (int iy = 0; iy < gh; iy++) {
          const int ylo = ytab(4 * iy + 3);
          if (ylo == row) {
            wy += __int_as_float(ytab(4 * iy + 1));
            touched = true;
          }
          if (ylo + 1 == row) {
            wy += __int_as_float(ytab(4 * iy + 2));
            touched = true;
          }
        }
        if (!touched) continue;
        // row == height / xlo + 1 == width: the upper tap of a border sample (weight 0), read from the last row / column
        const int row_bytes = min(row, height - 1) * width * pixel_bytes;
        if constexpr (kSR > 0) {
#pragma unroll
          for (int pw0 = 0; pw0 < PW; pw0 += PB) {
            float f[PB][kSR][2][V];
#pragma unroll
            for (int j = 0; j < PB; j++)
#pragma unroll
              for (int ix = 0; ix < kSR; ix++)
                if (pw0 + j < PW) {
                  const int tap = row_bytes + tap_lo[(pw0 + j) * kSR + ix];
                  load_tap<V>(img, lane_off, tap, f[j][ix][0]);
                  load_tap<V>(img, lane_off, tap + tap_up[(pw0 + j) * kSR + ix], f[j][ix][1]);
                }
            __builtin_amdgcn_sched_barrier(0);  // every tap of the batch is in flight before the first FMA
#pragma unroll
            for (int j = 0; j < PB; j++)
              if (pw0 + j < PW) {
                float s[V];
#pragma unroll
                for (int k = 0; k < V; k++) s[k] = 0.f;
#pragma unroll
                for (int ix = 0; ix < kSR; ix++) {
                  const int e = 4 * ((pw0 + j) * kSR + ix);
                  const float hx = __int_as_float(xtab(e + 1)), lx = __int_as_float(xtab(e + 2));
#pragma unroll
                  for (int k = 0; k < V; k++) s[k] = fmaf(lx, f[j][ix][1][k], fmaf(hx, f[j][ix][0][k], s[k]));
                }
#pragma unroll
                for (int k = 0; k < V; k++) acc[pw0 + j][k] = fmaf(wy, s[k], acc[pw0 + j][k]);
              }
          }
        } else {
#pragma unroll
          for (int pw = 0; pw < PW; pw++) {
            float s[V];
#pragma unroll
            for (int k = 0; k < V; k++) s[k] = 0.f;
            for (int ix = 0; ix < gw; ix++) {
              const int e = 4 * (pw * gw + ix);
              const int xlo = xtab(e + 3);
              const float hx = __int_as_float(xtab(e + 1)), lx = __int_as_float(xtab(e + 2));
              float f0[V], f1[V];
              const int tap = row_bytes + xlo * pixel_bytes;
              load_tap<V>(img, lane_off, tap, f0);
              load_tap<V>(img, lane_off, tap + (xlo + 1 < width ? pixel_bytes : 0), f1);
#pragma unroll
              for (int k = 0; k < V; k++) s[k] = fmaf(lx, f1[k], fmaf(hx, f0[k], s[k]));
            }
#pragma unroll
            for (int k = 0; k < V; k++) acc[pw][k] = fmaf(wy, s[k], acc[pw][k]);
          }
        }
      }
    } else if (!(flags & kFlagZero)) {
      // reference operation order, one output at a time (roi_align_kernel.cu:103-117)
      const RoiGeom g = roi_geometry(rois + (long long)r * 5, spatial_scale, aligned_height, PW, sampling_ratio);
#pragma unroll
      for (int pw = 0; pw < PW; pw++) {
        float o[V];
#pragma unroll
        for (int k = 0; k < V; k++) o[k] = 0.f;
        for (int iy = 0; iy < g.grid_h; iy++) {
          const float y = sample_y(g, ph, iy);
          for (int ix = 0; ix < g.grid_w; ix++) {
            const float x = sample_x(g, pw, ix);
            const Taps t = sample_taps(height, width, y, x);
            if (t.y_low >= 0) {
              float v1[V], v2[V], v3[V], v4[V];
              load_tap<V>(img, lane_off, (t.y_low * width + t.x_low) * pixel_bytes, v1);
              load_tap<V>(img, lane_off, (t.y_low * width + t.x_high) * pixel_bytes, v2);
              load_tap<V>(img, lane_off, (t.y_high * width + t.x_low) * pixel_bytes, v3);
              load_tap<V>(img, lane_off, (t.y_high * width + t.x_high) * pixel_bytes, v4);
#pragma unroll
              for (int k = 0; k < V; k++) o[k] += (t.w1 * v1[k] + t.w2 * v2[k] + t.w3 * v3[k] + t.w4 * v4[k]);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < V; k++) acc[pw][k] = o[k] / g.count;
      }
    }
    if (acc[0][0] == acc[0][0]) stamp(2);  // taps consumed (a use of the accumulators: the stamp waits for them)
#pragma unroll
    for (int pw = 0; pw < PW; pw++)
#pragma unroll
      for (int k = 0; k < V; k++) tile[(k * 64 + lane) * tile_stride + ph * PW + pw] = acc[pw][k];
  }
  stamp(3);
  __syncthreads();
  stamp(4);

  // tile -> out[r][c0 .. c0 + valid)[PH][PW]: one contiguous run
  const int valid = min(64 * V, channels - c0);
  const int nflt = valid * bins;
  float* __restrict__ dst = out + ((long long)r * channels + c0) * bins;
  if ((((long long)channels * bins) & 3) == 0) {
    for (int q4 = (int)threadIdx.x; q4 < (nflt >> 2); q4 += (int)blockDim.x) {
      int c = (q4 << 2) / bins, b = (q4 << 2) - c * bins;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        v[j] = tile[((c % V) * 64 + c / V) * tile_stride + b];
        if (++b == bins) {
          b = 0;
          c++;
        }
      }
      reinterpret_cast<float4*>(dst)[q4] = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {
    for (int q = (int)threadIdx.x; q < nflt; q += (int)blockDim.x) {
      const int c = q / bins, b = q - c * bins;
      dst[q] = tile[((c % V) * 64 + c / V) * tile_stride + b];
    }
  }
  stamp(5);
}

long long* g_nhwc_timeline = nullptr;

int pick_vec(int channels, int aligned_height, int aligned_width) {
  const int stride = (aligned_height * aligned_width) | 1;
  const auto fits = [&](int v) { return channels % (64 * v) == 0 && (size_t)64 * v * stride * 4 <= 64 * 1024; };
  const int want = tuning().nhwc_vec;
  if (want == 2 && fits(2)) return 2;
  if (want != 1 && fits(4)) return 4;
  if (want != 1 && fits(2)) return 2;
  if ((size_t)64 * stride * 4 <= 64 * 1024) return 1;
  return 0;
}

}  // namespace

void roi_align_fwd_nhwc_set_timeline(long long* device_buffer) { g_nhwc_timeline = device_buffer; }

bool roi_align_fwd_nhwc_supported(int channels, int height, int width, int num_rois, int aligned_height,
                                  int aligned_width) {
  return channels > 0 && height > 0 && width > 0 && (aligned_width == 7 || aligned_width == 14) &&
         aligned_height > 0 && aligned_height <= kMaxStages && pick_vec(channels, aligned_height, aligned_width) != 0 &&
         (long long)num_rois * ((channels + 63) / 64) < (1LL << 30) &&
         (long long)height * width * channels * 4 < (1LL << 31);  // 32-bit byte offsets inside one image
}

// The records of `rois` must already be in `workspace` (launch_roi_align_prepare on the same stream).
int launch_roi_align_fwd_nhwc_levels(const LevelTable& lv, const float* rois, float* output, const void* workspace,
                                     int batch, int channels, int num_rois, int aligned_height, int aligned_width,
                                     int sampling_ratio, hipStream_t stream) {
  const int* ws = static_cast<const int*>(workspace);
  const int v = pick_vec(channels, aligned_height, aligned_width);
  const int per = 64 * v, chunks = (channels + per - 1) / per;
  const int stride = (aligned_height * aligned_width) | 1;
  const size_t lds = (size_t)per * stride * sizeof(float);
  const int max_waves = aligned_width <= 7 ? 7 : 16;
  const int nwaves = aligned_height < max_waves ? aligned_height : max_waves;
  const int grid = num_rois * chunks;
  const bool sr2 = sampling_ratio == 2;
  // taps of 4 (then 3) output columns in flight (106 VGPRs: two 7-wave workgroups per CU) unless the whole row is
  // asked for (MI_ROI_ALIGN_NHWC_PB=7: 156 VGPRs, one workgroup per CU); measured equal at 512 RoIs
  const bool split = tuning().nhwc_pb != 7;
#define MI_LAUNCH_NHWC(SR, PW, V, PB)                                                                                 \
  roi_align_fwd_nhwc<SR, PW, V, PB><<<grid, 64 * nwaves, lds, stream>>>(                                              \
      rois, output, ws, batch, channels, aligned_height, sampling_ratio, chunks, stride, tuning().nhwc_order_mul,      \
      tuning().nhwc_zigzag, lv, g_nhwc_timeline)
#define MI_LAUNCH_NHWC_V(PW, V)                                                                                       \
  do {                                                                                                                \
    if (sr2 && split)                                                                                                 \
      MI_LAUNCH_NHWC(2, PW, V, 4);                                                                                    \
    else if (sr2)                                                                                                     \
      MI_LAUNCH_NHWC(2, PW, V, 7);                                                                                    \
    else                                                                                                              \
      MI_LAUNCH_NHWC(0, PW, V, 1);                                                                                    \
  } while (0)
  if (aligned_width == 7) {
    if (v == 4)
      MI_LAUNCH_NHWC_V(7, 4);
    else if (v == 2)
      MI_LAUNCH_NHWC_V(7, 2);
    else
      MI_LAUNCH_NHWC_V(7, 1);
  } else {
    if (v == 4)
      MI_LAUNCH_NHWC_V(14, 4);
    else if (v == 2)
      MI_LAUNCH_NHWC_V(14, 2);
    else
      MI_LAUNCH_NHWC_V(14, 1);
  }
#undef MI_LAUNCH_NHWC_V
#undef MI_LAUNCH_NHWC
  return check_launch("roi_align_fwd_nhwc");
}

int launch_roi_align_fwd_nhwc(const float* features, const float* rois, float* output, const void* workspace,
                              int batch, int channels, int height, int width, int num_rois, int aligned_height,
                              int aligned_width, float spatial_scale, int sampling_ratio, hipStream_t stream) {
  return launch_roi_align_fwd_nhwc_levels(single_level(features, nullptr, batch, height, width, spatial_scale), rois,
                                          output, workspace, batch, channels, num_rois, aligned_height, aligned_width,
                                          sampling_ratio, stream);
}

}  // namespace mi
