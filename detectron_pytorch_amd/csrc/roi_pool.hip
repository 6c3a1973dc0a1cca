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

// Workgroup = (RoI, 8 channels), grid (8 R, ceil(C / 64)): workgroups go round-robin over the XCDs, so XCD x pools channel tile
// 8 * phase + x for every RoI in arrival order -- ONE 8-channel slab of the map at a time (2.15 MB at 200x336: it fits the 4 MB
// L2, every line comes from the fabric once whatever the order of the RoIs; the records-free RoIAlign forward's mapping,
// roi_align_records.hip).  With 32-channel tiles an XCD's slab was 8.6 MB and the RoIs' arrival order thrashed it: config-2 shape
// 76.9 -> 66.9 us, stride-16 map with image-sized RoIs 91.2 -> 79.0 (tools/build_defines.sh MI_POOL_CT / MI_POOL_THREADS:
// 8 x 64 lanes 73.4, 8 x 128 68.9, 8 x 448 76.7, 16 x 256 69.4, 4 x 256 70.9; profiles/r06_pool_crop.txt).
#ifndef MI_POOL_CT
#define MI_POOL_CT 8
#endif
#ifndef MI_POOL_THREADS
#define MI_POOL_THREADS 256
#endif
constexpr int kPoolCT = MI_POOL_CT;        // channels per workgroup
constexpr int kPoolThreads = MI_POOL_THREADS;
constexpr bool kPoolSlab = MI_POOL_CT < 32;  // 32: rounds 1-6's mapping (RoI-major, tile = blockIdx % tiles)  // workgroups go round-robin over the XCDs: XCD x works on tile 8 * phase + x
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
  const int r = kPoolSlab ? (int)(blockIdx.x >> 3) : (int)(blockIdx.x / tiles);
  const int c0 = kPoolSlab ? (int)(blockIdx.y * 8 + (blockIdx.x & 7)) * kPoolCT : (int)(blockIdx.x - r * tiles) * kPoolCT;
  if (c0 >= channels) return;
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

// ---- backward -----------------------------------------------------------------------------------------------------------
// The reference's gather (one thread per INPUT element over all R RoIs, :128-203) turned inside out without giving up its
// result: a workgroup owns an 8 x 32-pixel tile of one image for 32 channels, its sums live in LDS (33 KB: four
// workgroups = 16 waves per CU), and it OVERWRITES the tile (:202 -- no zero fill by the caller, no global atomics).  The
// RoIs whose rounded rectangle meets the tile are listed in ascending index (the reference's outer loop order, :146);
// every wave walks the whole list for 8 of the channels, so no two waves ever touch one accumulator and nothing but
// program order inside a wave orders the sums.
// Per (RoI, tile) ONE wave tabulates: for every tile row / column the bins the reference would try for that pixel
// (:181-189), turned around by ballots into, per bin row / column of the span, the SET of tile rows / columns that try it,
// and then one dword per bin of the span (its offset in a channel's block, its row set, the row set of the bin row above,
// its column).  A walking wave takes the span's bins in row-major chunks of 32, lane = (channel, bin): argmax and gradient
// straight from memory, the argmax decoded to a tile pixel (a reciprocal multiplication in fp32, exact for these sizes),
// the reference's tests (same image, same channel, pixel inside the rectangle :161-165, bin among the pixel's candidates
// :191-193) as two shifts into the sets.
// ORDER: a pixel's terms must be added by ascending (RoI, ph, pw) (:146,:191-192).  When no pixel has more than two
// candidate bin rows or columns (every RoI at least a pixel per bin, i.e. all but tiny ones) a pixel gets at most four
// terms from a RoI, one per combination (first / second candidate row) x (first / second candidate column) -- and that
// pair, read as a number 0..3, IS their row-major order.  So a chunk is added in four passes by that number; inside a pass
// no two lanes meet in an accumulator, so the additions are plain LDS read / add / write (ds_add_f32 retires one lane
// per ~3 clocks on gfx950, tools/micro/lds_atomic_bench.hip), lanes with nothing to add working on a dword of their own.
// Otherwise the span's bins are taken one at a time.  Either way every pixel sees the reference's sequence of fp32
// additions: the result is bit-equal, run to run and to the reference.
// (Rounds 1-6a: one global atomic per output element into a zero-filled map, 221 us at the config-2 shape; the first tile
// form -- lane = (channel, bin row), columns in registers, three passes by row parity -- 104.6 us: 385 VALU instructions per
// (RoI, tile, 8 channels) with 12 of 64 lanes active on average.)
constexpr int kTileH = 8, kTileW = 32;       // pixels per tile
constexpr int kTileAcc = kTileH * kTileW + 4;  // accumulator stride of a channel: 16-byte rows, channels 4 banks apart
#ifndef MI_POOL_SCAN
#define MI_POOL_SCAN 512
#endif
#ifndef MI_POOL_SUB
#define MI_POOL_SUB 6
#endif
constexpr int kTileScan = MI_POOL_SCAN;      // RoIs scanned per round
constexpr int kTileSub = MI_POOL_SUB;        // (RoI, tile) entries tabulated at once
constexpr int kTileBins = 128;               // elements of a span's range an entry's table holds (more: one bin at a time)
constexpr int kTileEnt = 12;                 // dwords of an entry
constexpr int kTabRows = 0;                  // [1 + b]: the tile rows whose pixels try bin row ph0 + b; [0]: none
constexpr int kTabCols = 12;                 // [1 + j]: the tile columns whose pixels try bin column pw0 + j; [0]: none
constexpr int kTabBins = kTabCols + 36;      // [q]: element q of the span's range: row set | row set of the bin row above << 8 | j << 16
constexpr int kTabDwords = kTabBins + kTileBins;
constexpr int kTileThreads = 256;
// A wave owns CW channels (8, 4 or 2), a workgroup 4 x CW: 32 channels where the tiles alone fill the chip, fewer on small
// maps -- a tile's list is walked once per wave whatever it owns, so what a short grid lacks in workgroups it gets from
// thinner channel slices (a lane then holds CW / 2 of a chunk's 32 elements instead of four).
// LDS, in dwords: [4 * CW][kTileAcc] accumulators, then
template <int CW>
struct PoolLds {
  static constexpr int kc = 4 * CW;                              // channels per workgroup
  static constexpr int hits = kc * kTileAcc;                     // [kTileScan] RoI indices of the round, ascending
  static constexpr int tab = hits + kTileScan;                   // [kTileSub][kTabDwords]
  static constexpr int ent = tab + kTileSub * kTabDwords;        // [kTileSub][kTileEnt]
  static constexpr int wave_hits = ent + kTileSub * kTileEnt;    // [scan passes][4 waves]
  static constexpr int scratch = wave_hits + 8;                  // [256] a dword per lane: where a lane with nothing to add reads and writes
  static constexpr int dwords = scratch + kTileThreads;
};
static_assert(PoolLds<8>::dwords * 4 * 4 <= 160 * 1024, "four workgroups of 32 channels per CU");
enum { E_R = 0, E_PH0, E_NPH, E_PW0, E_NPW, E_FAST, E_SW, E_SH, E_EW, E_EH };

struct PoolTileRoi {  // what :155-179 derive from a RoI
  int start_w, start_h, end_w, end_h;
  float bin_size_h, bin_size_w;
};
__device__ __forceinline__ PoolTileRoi pool_tile_roi(int start_w, int start_h, int end_w, int end_h, int pooled_height,
                                                     int pooled_width) {
  PoolTileRoi g;
  g.start_w = start_w, g.start_h = start_h, g.end_w = end_w, g.end_h = end_h;
  const int roi_width = (int)fmaxf((float)(end_w - start_w + 1), 1.f);  // :175-176
  const int roi_height = (int)fmaxf((float)(end_h - start_h + 1), 1.f);
  g.bin_size_h = (float)roi_height / (float)pooled_height;  // :178-179
  g.bin_size_w = (float)roi_width / (float)pooled_width;
  return g;
}
// the bins [lo, hi) the reference tries for a pixel at distance d from the rectangle's start (:181-189)
__device__ __forceinline__ void pool_candidates(int d, float bin, int pooled, int& lo, int& hi) {
  lo = (int)floorf((float)d / bin);
  hi = (int)ceilf((float)(d + 1) / bin);
  lo = (int)fminf(fmaxf((float)lo, 0.f), (float)pooled);
  hi = (int)fminf(fmaxf((float)hi, 0.f), (float)pooled);
}

using lds_float_ptr = __attribute__((address_space(3))) float*;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_add(unsigned byte_addr, float v) {
  __hip_atomic_fetch_add((lds_float_ptr)(uintptr_t)byte_addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int CW>
__global__ void __launch_bounds__(kTileThreads)
roi_pool_bwd_tiles(const float* __restrict__ top_diff, const float* __restrict__ rois, const int32_t* __restrict__ argmax_data,
                   float* __restrict__ bottom_diff, int batch, int channels, int height, int width, int num_rois,
                   int pooled_height, int pooled_width, float spatial_scale, int tiles_x, int tiles_y, int cgroups,
                   float inv_width, int vec_ok, int ablate) {
  extern __shared__ __attribute__((aligned(16))) float pool_lds[];
  using L = PoolLds<CW>;
  constexpr int kTileKC = L::kc, kLdsHits = L::hits, kLdsTab = L::tab, kLdsEnt = L::ent, kLdsWaveHits = L::wave_hits, kLdsScratch = L::scratch;
  constexpr int kEpl = CW / 2;  // elements of a 32-element chunk per lane: one 16 / 8 / 4-byte load per array
  float* acc = pool_lds;
  int* ilds = (int*)pool_lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int cg = blockIdx.x % cgroups;  // one XCD's L2 serves the gradients of its channel groups
  int tile = blockIdx.x / cgroups;
  const int tx = tile % tiles_x;
  tile /= tiles_x;
  const int ty = tile % tiles_y, n = tile / tiles_y;
  const int th0 = ty * kTileH, tw0 = tx * kTileW, vh = min(kTileH, height - th0), vw = min(kTileW, width - tw0);
  const int c0 = cg * kTileKC;
  const int bins = pooled_height * pooled_width;
  // the bit sets hold bins < 32; the decode's reciprocal is exact for maps narrower than 16384 (inv_width 0: wider)
  const bool sets_fit = pooled_height <= 32 && pooled_width <= 32 && inv_width != 0.f;

  for (int i = tid; i < kTileKC * kTileAcc / 4; i += kTileThreads) ((float4*)acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  // this lane in the walk: channel cl of the wave's CW, element group k of the chunk's 64 / CW
  const int cl = lane % CW, k = lane / CW;
  const int c = c0 + wave * CW + cl;
  const bool cvalid = c < channels;
  const int tile_base = (((n * channels + (cvalid ? c : 0)) * height) + th0) * width + tw0;  // this lane's plane, the tile's corner
  const unsigned acc_bytes = (unsigned)((wave * CW + cl) * kTileAcc) * 4u;
  const unsigned scratch_dword = (unsigned)(kLdsScratch + tid) * 4u;
  const int chan_bytes = c * bins * 4;
  // kernel arguments are wave-uniform: descriptors over the two output-sized arrays (a lane without an element reads 0 past the end)
  const int out_bytes = (int)((unsigned)num_rois * (unsigned)channels * (unsigned)bins * 4u);
  const __amdgpu_buffer_rsrc_t arg_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(argmax_data), 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t top_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(top_diff), 0, out_bytes, 0x00020000);

  for (int base = 0; base < num_rois; base += kTileScan) {
    // ---- the RoIs of this round whose rectangle meets the tile, in ascending index
    unsigned long long votes[kTileScan / kTileThreads];
#pragma unroll
    for (int m = 0; m < kTileScan / kTileThreads; m++) {
      const int r = base + m * kTileThreads + tid;
      bool hit = false;
      if (r < num_rois) {
        const float* roi = rois + (long long)r * 5;
        const int start_w = (int)roundf(roi[1] * spatial_scale), start_h = (int)roundf(roi[2] * spatial_scale);
        const int end_w = (int)roundf(roi[3] * spatial_scale), end_h = (int)roundf(roi[4] * spatial_scale);
        hit = (int)roi[0] == n && start_w <= end_w && start_h <= end_h && start_w < tw0 + vw && end_w >= tw0 &&
              start_h < th0 + vh && end_h >= th0;
      }
      votes[m] = __ballot(hit);
      if (lane == 0) ilds[kLdsWaveHits + m * 4 + wave] = __popcll(votes[m]);
    }
    __syncthreads();  // (the first round: also the zeroed accumulators)
    int total = 0;
#pragma unroll
    for (int m = 0; m < kTileScan / kTileThreads; m++) {
      int before = total;
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int h = ilds[kLdsWaveHits + m * 4 + v];
        before += v < wave ? h : 0;
        total += h;
      }
      if ((votes[m] >> lane) & 1ull) ilds[kLdsHits + before + __popcll(votes[m] & ((1ull << lane) - 1ull))] = base + m * kTileThreads + tid;
    }
    total = uniform(total);
    if (MI_ABLATE(ablate & 1)) total = 0;

    for (int sub = 0; sub < total; sub += kTileSub) {
      const int nsub = min(kTileSub, total - sub);
      __syncthreads();  // the hit list is written / the previous entries are no longer read
      // ---- tabulate the entries, a wave per entry
      for (int e = wave; e < nsub; e += kTileThreads / 64) {
        const int r = uniform(ilds[kLdsHits + sub + e]);
        const const_float_ptr roi = (const_float_ptr)(uintptr_t)(rois + (long long)r * 5);
        const PoolTileRoi g = pool_tile_roi((int)roundf(roi[1] * spatial_scale), (int)roundf(roi[2] * spatial_scale),
                                            (int)roundf(roi[3] * spatial_scale), (int)roundf(roi[4] * spatial_scale), pooled_height,
                                            pooled_width);
        // lane < 8: tile row, lane 8..39: tile column -> the bins the reference tries for that pixel
        const bool row = lane < kTileH;
        const int pos = row ? lane : lane - kTileH;
        const int p = row ? th0 + pos : tw0 + pos;
        const int s0 = row ? g.start_h : g.start_w, s1 = row ? g.end_h : g.end_w;
        const bool inside = lane < kTileH + kTileW && pos < (row ? vh : vw) && p >= s0 && p <= s1;  // :161-165, and the pixel exists
        int lo = 0, hi = 0;
        if (inside) pool_candidates(p - s0, row ? g.bin_size_h : g.bin_size_w, row ? pooled_height : pooled_width, lo, hi);
        const bool wide = __ballot(hi - lo > 2) != 0;
        // the candidate sets grow with the pixel: the bins over the tile's part of the rectangle are
        // [first of the first pixel, last of the last)
        const int hA = max(g.start_h, th0) - th0, hB = min(g.end_h, th0 + vh - 1) - th0;
        const int wA = max(g.start_w, tw0) - tw0, wB = min(g.end_w, tw0 + vw - 1) - tw0;
        int ph0 = 0, nph = 0, pw0 = 0, npw = 0;
        if (hA <= hB && wA <= wB) {  // (the list holds only RoIs that meet the tile)
          ph0 = __builtin_amdgcn_readlane(lo, hA);
          nph = __builtin_amdgcn_readlane(hi, hB) - ph0;
          pw0 = __builtin_amdgcn_readlane(lo, kTileH + wA);
          npw = __builtin_amdgcn_readlane(hi, kTileH + wB) - pw0;
        }
        if (nph <= 0 || npw <= 0) nph = npw = 0;
        // (no pixel in more than two bins of an axis: at most kTileH + 1 bin rows and kTileW + 1 bin columns over the tile)
        // 1: every pixel in at most two bins of an axis (four ordered passes); 2: tiny RoIs, a pixel in three or more bins of an
        // axis (the same tables, the elements one at a time); 0: no tables (pooled sizes beyond the sets, maps too wide for
        // the reciprocal, spans beyond the table)
        const bool tabled = sets_fit && nph >= 1 && nph <= kTabCols - 2 && npw <= kTileW + 1 && (nph - 1) * pooled_width + npw <= kTileBins;
        const int mode = tabled ? (wide ? 2 : 1) : 0;
        int* tab = ilds + kLdsTab + e * kTabDwords;
        if (tabled) {
          const unsigned upto_hi = hi >= 32 ? 0xffffffffu : (1u << hi) - 1u, upto_lo = lo >= 32 ? 0xffffffffu : (1u << lo) - 1u;
          const unsigned tries = upto_hi & ~upto_lo;  // bit b: this pixel row / column tries bin b
          if (lane == 0) tab[kTabRows] = tab[kTabCols] = 0;
          for (int b = 0; b < nph; b++) {
            const unsigned long long set = __ballot(row && ((tries >> ((ph0 + b) & 31)) & 1u));
            if (lane == 0) tab[kTabRows + 1 + b] = (int)(unsigned)set;
          }
          for (int j = 0; j < npw; j++) {
            const unsigned long long set = __ballot(!row && ((tries >> ((pw0 + j) & 31)) & 1u));
            if (lane == 0) tab[kTabCols + 1 + j] = (int)(unsigned)(set >> kTileH);
          }
          __builtin_amdgcn_wave_barrier();
          // the span as it lies in a channel's block of the output: elements first .. last, row-major; those of the rows'
          // ends that are not in the span get an empty row set
          const int first = ph0 * pooled_width + pw0, count = (nph - 1) * pooled_width + npw;
          for (int q = lane; q < count; q += 64) {
            const int ph = (first + q) / pooled_width, j = first + q - ph * pooled_width - pw0, b = ph - ph0;
            const bool in_span = j >= 0 && j < npw;
            tab[kTabBins + q] = in_span ? tab[kTabRows + 1 + b] | (tab[kTabRows + b] << 8) | (j << 16) : 0;
          }
        }
        if (lane == 0) {
          int* ent = ilds + kLdsEnt + e * kTileEnt;
          ent[E_R] = r;
          ent[E_PH0] = ph0;
          ent[E_NPH] = nph;
          ent[E_PW0] = pw0;
          ent[E_NPW] = npw;
          ent[E_FAST] = mode;
          ent[E_SW] = g.start_w;
          ent[E_SH] = g.start_h;
          ent[E_EW] = g.end_w;
          ent[E_EH] = g.end_h;
        }
      }
      __syncthreads();

      // ---- the walk: every wave, every entry, its own eight channels
      // 32 elements of an entry's range from q0: lane = (channel, kEpl consecutive elements) -- one load per array
      // (per-element dword loads at CW = 8, 8 per chunk instead of 2, measured 99.9 us against 100.8 with the rest equal)
      auto fetch = [&](int e, int q0, bool wanted, int (&am)[kEpl], float (&grad)[kEpl]) {
        const int* ent = ilds + kLdsEnt + e * kTileEnt;
        const int r = uniform(ent[E_R]), first = uniform(ent[E_PH0]) * pooled_width + uniform(ent[E_PW0]);
        const unsigned roi_bytes = (unsigned)r * (unsigned)(channels * bins) * 4u;  // wave-uniform (the arrays stay below 4 GB)
        const int off = wanted && cvalid && uniform(ent[E_FAST]) ? chan_bytes + (first + q0 + kEpl * k) * 4 : -64;  // (beyond the descriptor: 0)
        if constexpr (kEpl == 4) {
          const v4i a = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(arg_srd, off, roi_bytes, 0));
          const v4f g = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(top_srd, off, roi_bytes, 0));
          am[0] = a.x, am[1] = a.y, am[2] = a.z, am[3] = a.w;
          grad[0] = g.x, grad[1] = g.y, grad[2] = g.z, grad[3] = g.w;
        } else if constexpr (kEpl == 2) {
          const v2i a = __builtin_bit_cast(v2i, __builtin_amdgcn_raw_buffer_load_b64(arg_srd, off, roi_bytes, 0));
          const v2f g = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(top_srd, off, roi_bytes, 0));
          am[0] = a.x, am[1] = a.y;
          grad[0] = g.x, grad[1] = g.y;
        } else {
          am[0] = __builtin_amdgcn_raw_buffer_load_b32(arg_srd, off, roi_bytes, 0);
          grad[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(top_srd, off, roi_bytes, 0));
        }
      };
      // a fetched chunk: decode, then the four ordered passes
      auto add_chunk = [&](int e, int q0, int count, int mode, const int (&am)[kEpl], const float (&grad)[kEpl]) {
        const int* tab = ilds + kLdsTab + e * kTabDwords;
        unsigned at[kEpl];
        int pass[kEpl];  // 2 x (the bin row is the pixel's second) + (the bin column is the pixel's second); -1: nothing to add
#pragma unroll
        for (int s = 0; s < kEpl; s++) {
          const int q = q0 + kEpl * k + s;
          const unsigned meta = cvalid && q < count ? (unsigned)tab[kTabBins + min(q, kTileBins - 1)] : 0u;
          const int j = (int)(meta >> 16);
          const unsigned cols = (unsigned)tab[kTabCols + 1 + j], cols_left = (unsigned)tab[kTabCols + j];
          // same image and channel, from the tile's first row on: a small number; its row = floor(rel / width), by the
          // reciprocal: (rel + 0.5) / width is at least 0.5 / width > 2^-15 away from an integer, the two roundings
          // move it by less than 2^-19 (rel < 8 * width < 2^17 where it matters; anything larger lands on rows >= 8)
          const unsigned rel = (unsigned)(am[s] - tile_base);
          const unsigned hl = (unsigned)(((float)rel + 0.5f) * inv_width);
          const unsigned wl = rel - hl * (unsigned)width;
          const unsigned rows = meta & 0xffu, rows_above = (meta >> 8) & 0xffu;
          const bool ok = hl < (unsigned)kTileH && wl < (unsigned)kTileW && (((rows >> hl) & (cols >> wl)) & 1u);
          at[s] = acc_bytes + (hl << 7) + (wl << 2);
          pass[s] = ok ? (int)(((rows_above >> hl) & 1u) * 2u + ((cols_left >> wl) & 1u)) : -1;
        }
        if (MI_ABLATE(ablate & 2)) return;
        if (mode == 2) {
          // a pixel in three or more bins of an axis: no pairing orders them -- the elements in their order, CW lanes
          // (the channels) at a time
          for (int kk = 0; kk < 64 / CW && q0 + kEpl * kk < count; kk++) {
#pragma unroll
            for (int s = 0; s < kEpl; s++) {
              if (k == kk && pass[s] >= 0) lds_add(at[s], grad[s]);
              __builtin_amdgcn_wave_barrier();
            }
          }
          return;
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
          // the terms whose pair is t: no two in one accumulator -- read all, add, write all
          unsigned where[kEpl];
          float sum[kEpl];
#pragma unroll
          for (int s = 0; s < kEpl; s++) {
            where[s] = pass[s] == t ? at[s] : scratch_dword;
            sum[s] = *(lds_float_ptr)(uintptr_t)where[s];
          }
#pragma unroll
          for (int s = 0; s < kEpl; s++) *(lds_float_ptr)(uintptr_t)where[s] = sum[s] + grad[s];
          __builtin_amdgcn_wave_barrier();  // a pixel's additions in program order
        }
      };
      auto take = [&](int e, const int (&am)[kEpl], const float (&grad)[kEpl]) {
        if (e >= nsub || MI_ABLATE(ablate & 4)) return;
        const int* ent = ilds + kLdsEnt + e * kTileEnt;
        const int r = uniform(ent[E_R]), nph = uniform(ent[E_NPH]), npw = uniform(ent[E_NPW]);
        if (nph == 0) return;
        const int mode = uniform(ent[E_FAST]);
        if (mode != 0) {
          const int count = (nph - 1) * pooled_width + npw;
          add_chunk(e, 0, count, mode, am, grad);
          for (int q0 = 32; q0 < count; q0 += 32) {  // ranges of more than 32 elements: further chunks, fetched in place
            int am_x[kEpl];
            float grad_x[kEpl];
            fetch(e, q0, true, am_x, grad_x);
            add_chunk(e, q0, count, mode, am_x, grad_x);
          }
        } else {
          // pooled sizes beyond the sets, spans beyond the table, maps too wide for the reciprocal: one bin at a time, the
          // reference's tests as written
          const int ph0 = uniform(ent[E_PH0]), pw0 = uniform(ent[E_PW0]);
          const PoolTileRoi g = pool_tile_roi(uniform(ent[E_SW]), uniform(ent[E_SH]), uniform(ent[E_EW]), uniform(ent[E_EH]),
                                              pooled_height, pooled_width);
          const long long block = (long long)r * channels * bins;
          for (int b = 0; b < nph * npw; b++) {
            const int ph = ph0 + b / npw, pw = pw0 + b % npw;
            if (cvalid && k == 0) {
              const int off = (c * pooled_height + ph) * pooled_width + pw;
              const unsigned rel = (unsigned)(argmax_data[block + off] - tile_base);
              if (rel < (unsigned)(vh * width)) {
                const int hl = (int)(rel / (unsigned)width), wl = (int)(rel - (unsigned)hl * (unsigned)width);
                const int h = th0 + hl, w = tw0 + wl;
                if (wl < vw && w >= g.start_w && w <= g.end_w && h >= g.start_h && h <= g.end_h) {  // :161-165
                  int phstart, phend, pwstart, pwend;
                  pool_candidates(h - g.start_h, g.bin_size_h, pooled_height, phstart, phend);
                  pool_candidates(w - g.start_w, g.bin_size_w, pooled_width, pwstart, pwend);
                  if (ph >= phstart && ph < phend && pw >= pwstart && pw < pwend)
                    lds_add(acc_bytes + (unsigned)((hl * kTileW + wl) << 2), top_diff[block + off]);
                }
              }
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
      };
      // two register sets: the first chunk of the next entry is in flight under the current one (a fetch past the last
      // entry repeats it: every fetch issues the same loads, so the waits the compiler places stay partial; all six
      // entries fetched up front measured 106 us against 100)
      int am_0[kEpl], am_1[kEpl];
      float grad_0[kEpl], grad_1[kEpl];
      fetch(0, 0, true, am_0, grad_0);
      for (int e = 0; e < nsub; e += 2) {
        fetch(min(e + 1, nsub - 1), 0, true, am_1, grad_1);
        take(e, am_0, grad_0);
        fetch(min(e + 2, nsub - 1), 0, true, am_0, grad_0);
        take(e + 1, am_1, grad_1);
      }
    }
    __syncthreads();  // the hit list is rewritten by the next round / the sums are complete
  }
  if (num_rois <= 0) __syncthreads();

  // ---- the tile leaves as rows of 128 bytes
  const int q = tid & 7, hrow = tid >> 3;  // a lane: four pixels of one row; 64 lanes per channel -> four channels per trip
  for (int cc = hrow / kTileH; cc < kTileKC; cc += kTileThreads / 8 / kTileH) {
    const int ch = c0 + cc, hl = hrow & (kTileH - 1);
    if (ch >= channels || hl >= vh || q * 4 >= vw) continue;
    const float4 v = *(const float4*)(acc + cc * kTileAcc + hl * kTileW + q * 4);
    float* dst = bottom_diff + (((long long)n * channels + ch) * height + th0 + hl) * width + tw0 + q * 4;
    if (q * 4 + 3 < vw && vec_ok) {
      *(float4*)dst = v;
    } else {
      dst[0] = v.x;
      if (q * 4 + 1 < vw) dst[1] = v.y;
      if (q * 4 + 2 < vw) dst[2] = v.z;
      if (q * 4 + 3 < vw) dst[3] = v.w;
    }
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
  const dim3 grid = kPoolSlab ? dim3((unsigned)num_rois * 8u, (unsigned)((tiles + 7) / 8)) : dim3((unsigned)(num_rois * tiles));
  roi_pool_fwd<<<grid, kPoolThreads, (size_t)2 * (pooled_height + pooled_width) * 4, mi::as_stream(stream)>>>(
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
  MI_REQUIRE(pooled_height < 32768 && pooled_width < 32768, "roi_pool: pooled size %d x %d beyond the tile tables", pooled_height, pooled_width);
  if ((long long)batch * channels * height * width == 0) return MI_OK;
  MI_REQUIRE(bottom_grad != nullptr, "roi_pool: null pointer");
  // every element of bottom_grad is written (roi_pooling_kernel.cu:202), also without a single RoI
  const int tiles_x = mi::ceil_div(width, kTileW), tiles_y = mi::ceil_div(height, kTileH);
  // channels per wave: the widest of 8 / 4 / 2 that still makes four workgroups per CU (measured, 512 image-sized RoIs on 50x84:
  // 256 channels 347 / 273 / 221 us, 1024 channels 356 / 293 / 373; the config-2 shape 71.6 / 75.6 / 105.9 -- profiles/r06_pool_crop.txt)
  const long long tiles = (long long)batch * tiles_y * tiles_x, enough = 4LL * mi::compute_units();
  int cw = tiles * mi::ceil_div(channels, 32) >= enough ? 8 : (tiles * mi::ceil_div(channels, 16) >= enough ? 4 : 2);
  const int forced = MI_ABLATE(mi::tuning().ablate >> 8) & 15;  // tuning builds: MI_ROI_ALIGN_ABLATE = 2048 / 1024 / 512 forces 8 / 4 / 2
  if (forced == 8 || forced == 4 || forced == 2) cw = forced;
  const int cgroups = mi::ceil_div(channels, 4 * cw);
  const long long grid = tiles * cgroups;
  MI_REQUIRE(grid < (1LL << 31), "roi_pool: too many tiles");
  MI_REQUIRE((long long)num_rois * channels * pooled_height * pooled_width * 4 < (1LL << 32), "roi_pool: output gradient beyond 4 GB");
  const float inv_width = width < 16384 ? 1.0f / (float)width : 0.f;
  const int vec_ok = (width & 3) == 0 && (reinterpret_cast<uintptr_t>(bottom_grad) & 15) == 0;
#define MI_POOL_BWD(CW)                                                                                                       \
  roi_pool_bwd_tiles<CW><<<(int)grid, kTileThreads, (size_t)PoolLds<CW>::dwords * 4, mi::as_stream(stream)>>>(                \
      top_grad, rois, argmax, bottom_grad, batch, channels, height, width, num_rois, pooled_height, pooled_width, spatial_scale, \
      tiles_x, tiles_y, cgroups, inv_width, vec_ok, mi::tuning().ablate)
  if (cw == 8)
    MI_POOL_BWD(8);
  else if (cw == 4)
    MI_POOL_BWD(4);
  else
    MI_POOL_BWD(2);
#undef MI_POOL_BWD
  return mi::check_launch("roi_pool_bwd");
}
