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
// The forward's workgroup = (RoI, 8 channels) of 128 lanes on a grid (8 R, ceil(C / 64)): an XCD samples ONE 8-channel slab of
// the map at a time (it fits its L2; roi_pool.hip, roi_align_records.hip: roi_align_fwd_slab).  Config-2 shape incl. the zero
// fill: 37.3 -> 31.4 us (8 x 64 lanes 32.2, 8 x 256 35.8, 16 x 256 33.5; tools/build_defines.sh MI_CROP_CT / MI_CROP_THREADS).
#ifndef MI_CROP_CT
#define MI_CROP_CT 8
#endif
#ifndef MI_CROP_THREADS
#define MI_CROP_THREADS 128
#endif
constexpr int kCropFwdCT = MI_CROP_CT, kCropFwdThreads = MI_CROP_THREADS;  // the forward's workgroup
constexpr bool kCropSlab = MI_CROP_CT < 32;  // grid (8 R, phases): XCD x samples channel tile 8 * phase + x (see roi_pool.hip)

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

__global__ void __launch_bounds__(kCropFwdThreads)
roi_crop_fwd(const float* __restrict__ input, const float* __restrict__ grids, float* __restrict__ output, int batch,
             int channels, int height, int width, int gh, int gw, int roiPerImage) {
  __shared__ PointTab tab;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int tiles = (channels + kCropFwdCT - 1) / kCropFwdCT;
  const int r = kCropSlab ? (int)(blockIdx.x >> 3) : (int)(blockIdx.x / tiles);
  const int c0 = kCropSlab ? (int)(blockIdx.y * 8 + (blockIdx.x & 7)) * kCropFwdCT : (int)(blockIdx.x - r * tiles) * kCropFwdCT;
  if (c0 >= channels) return;
  const int points = gh * gw;
  const int b_input = r / roiPerImage;  // :64
  const bool image_ok = b_input < batch;  // (the reference would read past its input)
  const int cvalid = min(kCropFwdCT, channels - c0);
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
    for (int i = tid; i < cvalid * np; i += kCropFwdThreads) {
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

// ---- backward, tile form (mi_roi_crop_backward_ws) ----------------------------------------------------------------------
// The reference adds every output element's four products into the image gradient with global atomics (:169-190).  Here
// a workgroup owns an 8 x 32-pixel tile of one image for 32 channels, its sums live in LDS and it OVERWRITES the tile (no
// zero fill, no global atomics).  A first launch leaves every RoI's bounding box of in-image taps in the caller's
// workspace (16 bytes per RoI); a tile lists the RoIs whose box meets it.  Per (RoI, 64 grid points, tile) one wave
// tabulates the points that have a tap in the tile (tile-local top-left, which taps, the four weight products of :169-172);
// every wave then walks the entries for its eight channels, lane = (channel, point), eight points per instruction.
// The additions are plain LDS read / add / write (ds_add_f32 retires one lane per ~3 clocks on gfx950,
// tools/micro/lds_atomic_bench.hip): one tap index at a time, so two lanes of one instruction meet in a pixel only if two
// points of the entry share their top-left pixel -- the tabulating wave finds that (each point writes its lane into a map
// of the tile and reads it back) and such entries (grids denser than the pixels) take ds_add_f32 instead.  The sums are
// the reference's terms, (x weight * y weight) * gradient, in another order (the reference's is undefined).
#ifndef MI_CROP_ACC_STRIDE
#define MI_CROP_ACC_STRIDE 260
#endif
#ifndef MI_CROP_SCAN
#define MI_CROP_SCAN 256
#endif
#ifndef MI_CROP_KC
#define MI_CROP_KC 32
#endif
#ifndef MI_CROP_SUB
#define MI_CROP_SUB 4
#endif
constexpr int kBtH = 8, kBtW = 32, kBtKC = MI_CROP_KC, kBtCW = 8, kBtThreads = kBtKC / kBtCW * 64, kBtWaves = kBtThreads / 64;
constexpr int kBtAcc = MI_CROP_ACC_STRIDE;  // accumulator stride of a channel (dwords): rows of 32, 16-byte aligned, banks spread
constexpr int kBtScan = MI_CROP_SCAN;     // RoIs scanned per round
constexpr int kBtSub = MI_CROP_SUB;       // entries tabulated at once
constexpr int kBtTab = 64 * 5;            // dwords of an entry's table: 64 x meta, 64 x 4 weights
constexpr int kBtMap = (kBtH + 1) * (kBtW + 2);  // top-left cells a point with a tap in the tile can have (+ padding)
static_assert(kBtMap <= kBtTab, "the map of top-left cells is laid over the table it precedes");
// LDS, dwords
constexpr int kBtHits = kBtKC * kBtAcc;
constexpr int kBtTabs = kBtHits + kBtScan;             // [kBtSub][kBtTab]: 64 x meta = (ty + 1) | (tx + 1) << 8 | taps << 16 | point << 20,
constexpr int kBtTabWts = 64;                          //                   then 64 x 4 weights (16-byte aligned)
constexpr int kBtEnt = kBtTabs + kBtSub * kBtTab;      // [kBtSub][4]  RoI, points kept, some two points share a top-left
constexpr int kBtWaveHits = kBtEnt + kBtSub * 4;       // [scan passes][waves]
constexpr int kBtScratch = kBtWaveHits + (kBtScan + kBtThreads - 1) / kBtThreads * kBtWaves;  // [threads] a dword per lane
constexpr int kBtDwords = kBtScratch + kBtThreads;
static_assert((kBtTabs + kBtTabWts) % 4 == 0 && kBtTab % 4 == 0 && kBtAcc % 4 == 0 && kBtAcc >= kBtH * kBtW, "16-byte alignment of weights and accumulator rows");

using lds_f32_ptr = __attribute__((address_space(3))) float*;

// one wave per RoI: rows [y0, y1] and columns [x0, x1] of its taps that lie in the image (y1 < y0: none)
__global__ void __launch_bounds__(256)
roi_crop_boxes(const float* __restrict__ grids, int4* __restrict__ boxes, int batch, int height, int width, int num_rois,
               int points, int roiPerImage) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= num_rois) return;
  const bool image_ok = r / roiPerImage < batch;
  int ylo = 0x3fffffff, yhi = -1, xlo = 0x3fffffff, xhi = -1;
  for (int p = lane; p < points && image_ok; p += 64) {
    int xTL, yTL;
    float xw, yw;
    get_top_left(grids[((long long)r * points + p) * 2 + 1], width, xTL, xw);
    get_top_left(grids[((long long)r * points + p) * 2], height, yTL, yw);
    const bool anyx = between(xTL, 0, width - 1) || between(xTL + 1, 0, width - 1);
    const bool anyy = between(yTL, 0, height - 1) || between(yTL + 1, 0, height - 1);
    if (anyx && anyy) {
      ylo = min(ylo, max(yTL, 0));
      yhi = max(yhi, min(yTL + 1, height - 1));
      xlo = min(xlo, max(xTL, 0));
      xhi = max(xhi, min(xTL + 1, width - 1));
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    ylo = min(ylo, __shfl_xor(ylo, d));
    yhi = max(yhi, __shfl_xor(yhi, d));
    xlo = min(xlo, __shfl_xor(xlo, d));
    xhi = max(xhi, __shfl_xor(xhi, d));
  }
  if (lane == 0) boxes[r] = make_int4(ylo, yhi, xlo, xhi);
}

#ifdef MI_TILE_TIMELINE
__device__ unsigned long long crop_tl[4 * 8192];  // per workgroup: start, end (100 MHz), entries walked, hardware id
#endif

__global__ void __launch_bounds__(kBtThreads)
roi_crop_bwd_tiles(const float* __restrict__ grids, const float* __restrict__ grad_output, float* __restrict__ grad_input,
                   const int4* __restrict__ boxes, int batch, int channels, int height, int width, int num_rois, int points,
                   int roiPerImage, int tiles_x, int tiles_y, int cgroups, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) float crop_lds[];
  float* acc = crop_lds;
  int* ilds = (int*)crop_lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int cg = blockIdx.x % cgroups;  // one XCD's L2 serves the gradients of its channel groups
  int tile = blockIdx.x / cgroups;
  const int tx = tile % tiles_x;
  tile /= tiles_x;
  const int ty = tile % tiles_y, n = tile / tiles_y;
  const int th0 = ty * kBtH, tw0 = tx * kBtW, vh = min(kBtH, height - th0), vw = min(kBtW, width - tw0);
  const int c0 = cg * kBtKC;
  const int groups = (points + 63) / 64;
#ifdef MI_TILE_TIMELINE
  const unsigned long long tl_start = wall_clock64();
  int tl_entries = 0;
#endif

  for (int i = tid; i < kBtKC * kBtAcc / 4; i += kBtThreads) ((float4*)acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  // this lane in the walk: channel cl of the wave's eight, point slot k of eight
  const int cl = lane & (kBtCW - 1), k = lane >> 3;
  const int c = c0 + wave * kBtCW + cl;
  const bool cvalid = c < channels;
  const unsigned acc_bytes = (unsigned)((wave * kBtCW + cl) * kBtAcc) * 4u;
  const unsigned scratch_dword = (unsigned)(kBtScratch + tid) * 4u;
  const int out_bytes = (int)((unsigned)num_rois * (unsigned)channels * (unsigned)points * 4u);
  const __amdgpu_buffer_rsrc_t top_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(grad_output), 0, out_bytes, 0x00020000);

  // the RoIs of image n (:128): r / roiPerImage == n
  const int r_first = n * roiPerImage, r_end = min(num_rois, r_first + roiPerImage);
  for (int base = r_first; base < r_end || base == r_first; base += kBtScan) {
    constexpr int kPasses = (kBtScan + kBtThreads - 1) / kBtThreads;
    unsigned long long votes[kPasses];
#pragma unroll
    for (int m = 0; m < kPasses; m++) {
      const int r = base + m * kBtThreads + tid;
      bool hit = false;
      if (r < r_end && m * kBtThreads + tid < kBtScan) {
        const int4 b = boxes[r];
        hit = b.x < th0 + vh && b.y >= th0 && b.z < tw0 + vw && b.w >= tw0 && b.y >= b.x;
      }
      votes[m] = __ballot(hit);
      if (lane == 0) ilds[kBtWaveHits + m * kBtWaves + wave] = __popcll(votes[m]);
    }
    __syncthreads();  // (the first round: also the zeroed accumulators)
    int total = 0;
#pragma unroll
    for (int m = 0; m < kPasses; m++) {
      int before = total;
#pragma unroll
      for (int v = 0; v < kBtWaves; v++) {
        const int h = ilds[kBtWaveHits + m * kBtWaves + v];
        before += v < wave ? h : 0;
        total += h;
      }
      if ((votes[m] >> lane) & 1ull) ilds[kBtHits + before + __popcll(votes[m] & ((1ull << lane) - 1ull))] = base + m * kBtThreads + tid;
    }
    const int entries = uniform(total) * groups;  // an entry: (RoI, 64 of its grid points)
#ifdef MI_TILE_TIMELINE
    tl_entries += entries;
#endif

    for (int sub = 0; sub < entries; sub += kBtSub) {
      const int nsub = min(kBtSub, entries - sub);
      __syncthreads();  // the hit list is written / the previous tables are no longer read
      // ---- tabulate, a wave per entry, a lane per grid point
      for (int e = wave; e < nsub; e += kBtWaves) {
        const int r = uniform(ilds[kBtHits + (sub + e) / groups]), p = ((sub + e) % groups) * 64 + lane;
        int xTL = 0, yTL = 0, taps = 0;
        float xw = 0.f, yw = 0.f;
        if (p < points) {
          get_top_left(grids[((long long)r * points + p) * 2 + 1], width, xTL, xw);  // :140-141
          get_top_left(grids[((long long)r * points + p) * 2], height, yTL, yw);
          const int lx = xTL - tw0, ly = yTL - th0;  // in the image AND in the tile
          const bool xl = lx >= 0 && lx < vw, xr = lx + 1 >= 0 && lx + 1 < vw, yt = ly >= 0 && ly < vh, yb = ly + 1 >= 0 && ly + 1 < vh;
          taps = ((xl && yt) ? 1 : 0) | ((xr && yt) ? 2 : 0) | ((xl && yb) ? 4 : 0) | ((xr && yb) ? 8 : 0);
        }
        const int lx1 = xTL - tw0 + 1, ly1 = yTL - th0 + 1;  // 0 .. kBtW, 0 .. kBtH where taps != 0
        int* owner = ilds + kBtTabs + e * kBtTab;  // the map of top-left cells, laid over the table this entry is about to get
        const int cell = taps ? ly1 * (kBtW + 2) + lx1 : 0;
        if (taps) owner[cell] = lane;
        __builtin_amdgcn_wave_barrier();
        const bool shared = taps && owner[cell] != lane;   // another point of the entry has this top-left pixel
        __builtin_amdgcn_wave_barrier();                   // (the map is read before the table goes over it)
        const unsigned long long keep = __ballot(taps != 0);
        const int slot = __popcll(keep & ((1ull << lane) - 1ull));
        if (taps) {
          ilds[kBtTabs + e * kBtTab + slot] = ly1 | (lx1 << 8) | (taps << 16) | (p << 20);
          // the products of :169-172, formed as the reference forms them
          *(float4*)(crop_lds + kBtTabs + e * kBtTab + kBtTabWts + slot * 4) = make_float4(xw * yw, (1 - xw) * yw, xw * (1 - yw), (1 - xw) * (1 - yw));
        }
        const bool any_shared = __ballot(shared) != 0;
        if (lane == 0) {
          ilds[kBtEnt + e * 4] = r;
          ilds[kBtEnt + e * 4 + 1] = __popcll(keep);
          ilds[kBtEnt + e * 4 + 2] = any_shared ? 1 : 0;
        }
      }
      __syncthreads();

      // ---- the walk: every wave, every entry, its own eight channels
      for (int e = 0; e < nsub; e++) {
        const int r = uniform(ilds[kBtEnt + e * 4]), npts = uniform(ilds[kBtEnt + e * 4 + 1]);
        const bool shared = uniform(ilds[kBtEnt + e * 4 + 2]) != 0;
        if (npts == 0) continue;
        const unsigned roi_bytes = (unsigned)r * (unsigned)channels * (unsigned)points * 4u;  // wave-uniform
        const int chan_bytes = c * points * 4;
        for (int q0 = 0; q0 < npts; q0 += 32) {  // 32 points at a time: four per lane
          unsigned at[4][4];
          float term[4][4];
#pragma unroll
          for (int s = 0; s < 4; s++) {
            const int q = q0 + s * 8 + k;
            const bool live = cvalid && q < npts;
            const int meta = ilds[kBtTabs + e * kBtTab + min(q, 63)];
            const float4 w = *(const float4*)(crop_lds + kBtTabs + e * kBtTab + kBtTabWts + min(q, 63) * 4);
            const float g = __builtin_bit_cast(  // (a lane without a point reads 0 from beyond the descriptor)
                float, __builtin_amdgcn_raw_buffer_load_b32(top_srd, live ? chan_bytes + (int)((unsigned)meta >> 20) * 4 : -64, roi_bytes, 0));
            const int taps = live ? (meta >> 16) & 15 : 0;
            // the top-left pixel, tile-local, may lie one row above / one column left of the tile: only taps in it are used
            const unsigned tl = acc_bytes + (unsigned)((((meta & 0xff) - 1) * kBtW + ((meta >> 8) & 0xff) - 1) * 4);
            at[s][0] = (taps & 1) ? tl : scratch_dword;
            at[s][1] = (taps & 2) ? tl + 4u : scratch_dword;
            at[s][2] = (taps & 4) ? tl + kBtW * 4u : scratch_dword;
            at[s][3] = (taps & 8) ? tl + kBtW * 4u + 4u : scratch_dword;
            term[s][0] = w.x * g;
            term[s][1] = w.y * g;
            term[s][2] = w.z * g;
            term[s][3] = w.w * g;
          }
          if (!shared) {
            // one tap index at a time: the pixels of one instruction's lanes are distinct, so read all, add, write all
#pragma unroll
            for (int t = 0; t < 4; t++) {
              float v[4];
#pragma unroll
              for (int s = 0; s < 4; s++) v[s] = *(lds_f32_ptr)(uintptr_t)at[s][t];
#pragma unroll
              for (int s = 0; s < 4; s++) *(lds_f32_ptr)(uintptr_t)at[s][t] = v[s] + term[s][t];
              __builtin_amdgcn_wave_barrier();
            }
          } else {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
              for (int s = 0; s < 4; s++)
                if (at[s][t] != scratch_dword)
                  __hip_atomic_fetch_add((lds_f32_ptr)(uintptr_t)at[s][t], term[s][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    __syncthreads();  // the hit list is rewritten by the next round / the sums are complete
  }

  // ---- the tile leaves as rows of 128 bytes
  const int q = tid & 7, hrow = tid >> 3;  // a lane: four pixels of one row
  for (int cc = hrow / kBtH; cc < kBtKC; cc += kBtThreads / 8 / kBtH) {
    const int ch = c0 + cc, hl = hrow & (kBtH - 1);
    if (ch >= channels || hl >= vh || q * 4 >= vw) continue;
    const float4 v = *(const float4*)(acc + cc * kBtAcc + hl * kBtW + q * 4);
    float* dst = grad_input + (((long long)n * channels + ch) * height + th0 + hl) * width + tw0 + q * 4;
    if (q * 4 + 3 < vw && vec_ok) {
      *(float4*)dst = v;
    } else {
      dst[0] = v.x;
      if (q * 4 + 1 < vw) dst[1] = v.y;
      if (q * 4 + 2 < vw) dst[2] = v.z;
      if (q * 4 + 3 < vw) dst[3] = v.w;
    }
  }
#ifdef MI_TILE_TIMELINE
  __syncthreads();
  if (tid == 0 && blockIdx.x < 8192) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    crop_tl[4 * blockIdx.x] = tl_start;
    crop_tl[4 * blockIdx.x + 1] = wall_clock64();
    crop_tl[4 * blockIdx.x + 2] = tl_entries;
    crop_tl[4 * blockIdx.x + 3] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
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
  const int tiles = (channels + kCropFwdCT - 1) / kCropFwdCT;
  const dim3 grid = kCropSlab ? dim3((unsigned)num_rois * 8u, (unsigned)((tiles + 7) / 8)) : dim3((unsigned)(num_rois * tiles));
  roi_crop_fwd<<<grid, kCropFwdThreads, 0, mi::as_stream(stream)>>>(
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

extern "C" size_t mi_roi_crop_backward_workspace_bytes(int num_rois) { return (size_t)(num_rois > 0 ? num_rois : 0) * sizeof(int4) + 16; }

extern "C" int mi_roi_crop_backward_ws(const float* input, const float* grid_yx, const float* grad_output, float* grad_input,
                                       int batch, int channels, int height, int width, int num_rois, int grid_height,
                                       int grid_width, void* workspace, size_t workspace_bytes, mi_stream_t stream) {
  mi::begin_call();
  (void)input;  // the reference reads it only for the grid gradient it then discards (:166-190)
  int rc = check_crop(grad_output, grid_yx, grad_input, batch, channels, height, width, num_rois, grid_height, grid_width);
  if (rc != MI_OK) return rc;
  if ((long long)batch * channels * height * width == 0) return MI_OK;
  MI_REQUIRE(grad_input != nullptr, "roi_crop: null pointer");
  MI_REQUIRE(workspace != nullptr && workspace_bytes >= mi_roi_crop_backward_workspace_bytes(num_rois) &&
                 (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
             "roi_crop: the backward needs a 16-byte aligned workspace of mi_roi_crop_backward_workspace_bytes(num_rois) bytes");
  const int points = grid_height * grid_width;
  MI_REQUIRE(points < 4096, "roi_crop: more than 4095 grid points per RoI");
  MI_REQUIRE((long long)num_rois * channels * points * 4 < (1LL << 32), "roi_crop: output gradient beyond 4 GB");
  const int roiPerImage = num_rois > 0 ? num_rois / batch : 1;
  int4* boxes = static_cast<int4*>(workspace);
  if (num_rois > 0) {
    roi_crop_boxes<<<mi::ceil_div(num_rois, 4), 256, 0, mi::as_stream(stream)>>>(grid_yx, boxes, batch, height, width, num_rois,
                                                                                points, roiPerImage);
    rc = mi::check_launch("roi_crop_boxes");
    if (rc != MI_OK) return rc;
  }
  const int tiles_x = mi::ceil_div(width, kBtW), tiles_y = mi::ceil_div(height, kBtH), cgroups = mi::ceil_div(channels, kBtKC);
  const long long grid = (long long)batch * tiles_y * tiles_x * cgroups;
  MI_REQUIRE(grid < (1LL << 31), "roi_crop: too many tiles");
  const size_t lds = (size_t)kBtDwords * 4;
  const int vec_ok = (width & 3) == 0 && (reinterpret_cast<uintptr_t>(grad_input) & 15) == 0;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&roi_crop_bwd_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  roi_crop_bwd_tiles<<<(int)grid, kBtThreads, lds, mi::as_stream(stream)>>>(grid_yx, grad_output, grad_input, boxes, batch, channels,
                                                                            height, width, num_rois, points, roiPerImage, tiles_x,
                                                                            tiles_y, cgroups, vec_ok);
  return mi::check_launch("roi_crop_bwd_tiles");
}

#ifdef MI_TILE_TIMELINE
// -DMI_TILE_TIMELINE builds only (tools/tile_timeline.py): start / end / entries / hardware id of the last launch's workgroups
extern "C" int mi_dbg_crop_timeline(unsigned long long* out, int workgroups) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(crop_tl), (size_t)workgroups * 4 * sizeof(unsigned long long)) == hipSuccess ? MI_OK : MI_ERR_LAUNCH;
}
#endif
