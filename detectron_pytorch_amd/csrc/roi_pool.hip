// roi_pool.hip -- RoIPool forward / backward for gfx950, C-ABI mi_roi_pool_*.
//
// Arithmetic contract: lib/model/roi_pooling/src/roi_pooling_kernel.cu:24-93 (forward: max over the bin's pixels, strict >,
// scanned row-major so the first maximum wins, int32 flat argmax over the whole input tensor, empty bin -> 0 / -1) and
// :128-203 (backward).
//
// Forward (round 6; until then the reference's one-lane-per-output loop re-typed): one 256-lane workgroup per (RoI,
// 32-channel tile), tile = blockIdx % tiles so that one XCD's L2 serves one channel slab.  The RoI's rectangle is decoded
// once per workgroup (wave-uniform, SGPRs); its rows arrive in LDS by LDS-DMA (buffer_load_dwordx4 ... lds, lanes flattened
// over (row, 16-byte group), one odd-stride plane per channel) in chunks of as many rows as the image holds; lane & 31 =
// channel (32 banks), half-wave = bin row, the wave walks the bin columns together (scalar loop bounds); every lane scans the part of its bins that lies in the chunk and carries
// (max, argmax) in the LDS tile, so a bin taller than a chunk is scanned across chunks in the reference's row-major order.
// The [32][bins] tile of values and of indices then leaves as contiguous runs.  Bit-exact: the comparisons are the
// reference's, on the same values in the same order.
// A window wider than the LDS image (more than 296 columns) is scanned from memory by the same lanes.
// Measured at the config-2 shape (profiles/r06_pool_crop.txt): 86 us against 100 us for the one-lane-per-output kernel it
// replaces; by ablation 26 us of window DMA, 44 us of scan -- a chain of LDS round trips per bin (tile entry, bin columns,
// one per row pair), not VALU issue (per pixel: one compare, two selects) -- 8 us of stores, 14 us of skeleton.  Tried and
// not kept: 512 / 1024-lane workgroups (108-127 us), all seven bins of a bin row in registers per image row (103 us: the
// per-bin column bounds in SGPRs spill).
//
// Backward: the reference launches one thread per INPUT element and loops over all R RoIs and their candidate bins
// comparing argmax == index: O(N*C*H*W*R).  The same sums are produced here by scattering each output gradient through
// its argmax (one fp32 atomic per output element, O(R*C*PH*PW)); the reference's extra conditions -- the argmax pixel
// must lie inside the rounded RoI rectangle [start, end] (:161-165, false for malformed RoIs whose width was forced to
// 1) -- are re-checked so the set of contributing terms is identical.  Only the floating-point addition order differs
// (reference: ascending RoI index).
#include "common.h"
#include "lds_dma.h"

#include <cfloat>

#ifndef MI_POOL_THREADS
#define MI_POOL_THREADS 256
#define MI_POOL_CAP 296
#endif

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
constexpr int kPoolThreads = MI_POOL_THREADS;
constexpr int kPoolSlots = kPoolThreads / kPoolCT;  // half-waves
constexpr int kPoolCap = MI_POOL_CAP;  // window pixels per channel of the LDS image
constexpr int kPoolPlane = kPoolCap | 1;
constexpr int kPoolTileBins = 56;  // bins per channel the LDS tile holds at least (whole 7 x 7 outputs)

// bin p of an axis: [start, end) in map coordinates, clamped to the map (roi_pooling_kernel.cu:57-66)
__device__ __forceinline__ void pool_bin(int p, float bin_size, int roi_start, int size, int& lo, int& hi) {
  lo = (int)floorf((float)p * bin_size);
  hi = (int)ceilf((float)(p + 1) * bin_size);
  lo = (int)fminf(fmaxf((float)(lo + roi_start), 0.f), (float)size);
  hi = (int)fminf(fmaxf((float)(hi + roi_start), 0.f), (float)size);
}

// Rows [hs, he) x kN columns of one bin from the LDS image, two rows per step: all 2 * kN pixels are read before the first
// compare.  Order of the compares = the reference's scan (:77-87): row-major, strict >.
template <int kN>
__device__ __forceinline__ void pool_scan_rows(const float* rowp, int pitch_px, int hs, int he, int rowidx, int width,
                                               float& maxval, int& maxidx) {
  for (int h = hs; h < he; h += 2) {
    float a[kN], b[kN];
    const bool two = h + 1 < he;
#pragma unroll
    for (int j = 0; j < kN; j++) a[j] = rowp[j];
#pragma unroll
    for (int j = 0; j < kN; j++) b[j] = rowp[pitch_px + j];  // (past the bin on its last odd row: read, never compared)
    // the scan is bound by VALU issue: per pixel one compare and two selects -- the winner's POSITION in the row pair is an
    // instruction constant, its flat index is formed once per row pair
    int best = -1;
#pragma unroll
    for (int j = 0; j < kN; j++) {
      const bool up = a[j] > maxval;
      maxval = up ? a[j] : maxval;
      best = up ? j : best;
    }
    if (two) {
#pragma unroll
      for (int j = 0; j < kN; j++) {
        const bool up = b[j] > maxval;
        maxval = up ? b[j] : maxval;
        best = up ? kN + j : best;
      }
    }
    if (best >= 0) maxidx = rowidx + (best >= kN ? width + best - kN : best);
    rowp += 2 * pitch_px;
    rowidx += 2 * width;
  }
}

__global__ void __launch_bounds__(kPoolThreads) __attribute__((amdgpu_waves_per_eu(kPoolThreads >= 1024 ? 8 : 1, 8)))
roi_pool_fwd(const float* __restrict__ bottom_data, const float* __restrict__ rois, float* __restrict__ top_data,
             int32_t* __restrict__ argmax_data, int batch, int channels, int height, int width, int pooled_height,
             int pooled_width, float spatial_scale, int rows_per_group, int ablate_arg) {
  const int ablate = MI_ABLATE(ablate_arg);  // tuning builds: 1 = no window DMA, 2 = no scan, 4 = no store
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tile_bins = rows_per_group * pooled_width, ts = tile_bins | 1;
  float* tval = smem;                                        // [kPoolCT][ts]
  int* targ = reinterpret_cast<int*>(smem + kPoolCT * ts);   // [kPoolCT][ts]
  float* img = smem + 2 * kPoolCT * ts;                      // [kPoolCT][kPoolPlane]
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int cl = tid % kPoolCT, slot = tid / kPoolCT;
  const int tiles = (channels + kPoolCT - 1) / kPoolCT;
  const int r = blockIdx.x / tiles, c0 = (blockIdx.x - r * tiles) * kPoolCT;
  const int bins = pooled_height * pooled_width;
  // ---- the RoI, once per workgroup (five scalar loads) ----
  const const_float_ptr roi = (const_float_ptr)(uintptr_t)(rois + (long long)r * 5);
  const int batch_ind = (int)roi[0];
  const int start_w = (int)roundf(roi[1] * spatial_scale), start_h = (int)roundf(roi[2] * spatial_scale);  // :46-49
  const int end_w = (int)roundf(roi[3] * spatial_scale), end_h = (int)roundf(roi[4] * spatial_scale);
  const int roi_width = (int)fmaxf((float)(end_w - start_w + 1), 1.f);  // :52-53
  const int roi_height = (int)fmaxf((float)(end_h - start_h + 1), 1.f);
  const float bin_size_h = (float)roi_height / (float)pooled_height;  // :54-55
  const float bin_size_w = (float)roi_width / (float)pooled_width;
  const bool no_image = batch_ind < 0 || batch_ind >= batch;
  // window columns: bin starts and ends grow with pw, so the first start and the last end bound them all
  int wlo, whi, t0, t1;
  pool_bin(0, bin_size_w, start_w, width, wlo, t0);
  pool_bin(pooled_width - 1, bin_size_w, start_w, width, t1, whi);
  const int ww = whi - wlo;
  const int pitch_px = (max(ww, 1) + 3) & ~3, gpr = pitch_px >> 2;
  const bool staged = pitch_px <= kPoolCap;
  const int chunk_rows = staged ? kPoolCap / pitch_px : (1 << 30);
  const unsigned gmagic = (1u << 20) / (unsigned)gpr + 1u;
  const long long plane_px = (long long)height * width;
  const int cvalid = min(kPoolCT, channels - c0);  // channels of this tile that exist
  const float* __restrict__ src = bottom_data + ((long long)(no_image ? 0 : batch_ind) * channels + c0) * plane_px;
  constexpr int kChPerWave = kPoolCT / (kPoolThreads / 64);
  // this wave's DMA covers the planes of its kChPerWave channels (those that exist); a lane past them reads zeros
  const int wave_ch = max(0, min(kChPerWave, cvalid - wave * kChPerWave));
  const srd_t srd = make_srd(src + (long long)wave * kChPerWave * plane_px, (unsigned)(wave_ch * plane_px * 4));
  const unsigned plane0 = lds_addr_uniform(img + wave * kChPerWave * kPoolPlane);

  // bin rows / columns of this RoI, once per workgroup: hb[2 ph] .. = [start, end) of bin row ph, wb likewise per bin column
  int* hb = reinterpret_cast<int*>(img + kPoolCT * kPoolPlane);
  int* wb = hb + 2 * pooled_height;
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
  const int bottom_data_offset = (batch_ind * channels + c0 + cl) * height * width;  // :75-76
  for (int pa = 0; pa < pooled_height; pa += rows_per_group) {
    const int pb = min(pooled_height, pa + rows_per_group);
    const int nb = (pb - pa) * pooled_width;
    const int ra = uniform(hb[2 * pa]), rb = uniform(hb[2 * (pb - 1) + 1]);  // rows of this group of bin rows
    // ---- tile: every bin starts empty-or-open (:68-72).  A half-wave owns bin rows slot, slot + 8, ...: the bins a lane
    // initialises are the bins it scans and updates, chunk after chunk ----
    for (int ph = pa + slot; ph < pb; ph += kPoolSlots) {
      const bool row_empty = hb[2 * ph + 1] <= hb[2 * ph] || no_image;
      for (int pw = 0; pw < pooled_width; pw++) {
        const bool is_empty = row_empty || wb[2 * pw + 1] <= wb[2 * pw];
        tval[cl * ts + (ph - pa) * pooled_width + pw] = is_empty ? 0.f : -FLT_MAX;
        targ[cl * ts + (ph - pa) * pooled_width + pw] = -1;
      }
    }
    if (!no_image && ww > 0)
      for (int r0 = ra; r0 < rb; r0 += chunk_rows) {
        const int r1 = min(rb, r0 + chunk_rows);
        if (staged && !(ablate & 1)) {
          __syncthreads();  // the previous chunk's scans are done with the image
          const unsigned groups = (unsigned)(r1 - r0) * (unsigned)gpr;
          for (int kk = 0; kk * 64 < (int)groups; kk++) {
            const unsigned g = (unsigned)(kk * 64 + lane);
            const unsigned q = __umul24(g, gmagic) >> 20;  // g / gpr
            const unsigned gc = g - __umul24(q, (unsigned)gpr);
            const unsigned voff = (((unsigned)r0 + q) * (unsigned)width + (unsigned)wlo + gc * 4u) * 4u;
            if (g < groups) {
#pragma unroll
              for (int c = 0; c < kChPerWave; c++)
                dma_dwordx4(srd, plane0 + (unsigned)(c * kPoolPlane + kk * 256) * 4u, voff, (unsigned)(c * plane_px * 4));
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();  // the chunk has landed
        }
        // Scan.  Bin COLUMNS are walked by the whole wave together (scalar loop bounds, the column offset of a tap is an
        // instruction immediate or an SGPR), bin ROWS belong to half-waves: per pixel one LDS read, one compare, two selects
        // and the index add.
        if (cl < cvalid && !(ablate & 2))
          for (int ph = pa + slot; ph < pb; ph += kPoolSlots) {
            const int hs = max(hb[2 * ph], r0), he = min(hb[2 * ph + 1], r1);
            if (he <= hs) continue;
            for (int pw = 0; pw < pooled_width; pw++) {
              const int ws = uniform(wb[2 * pw]), we = uniform(wb[2 * pw + 1]);
              if (we <= ws) continue;
              const int b = (ph - pa) * pooled_width + pw;
              float maxval = tval[cl * ts + b];
              int maxidx = targ[cl * ts + b];
              const int ncol = we - ws;
              if (staged && ncol <= 8) {
                // two rows of the bin in registers per step (one LDS wait for 2 * ncol pixels instead of one per pixel),
                // compared in the reference's order: row h left to right, then row h + 1
                const float* rowp = img + cl * kPoolPlane + (hs - r0) * pitch_px + (ws - wlo);
                const int rowidx = bottom_data_offset + hs * width + ws;
                switch (ncol) {
                  case 1: pool_scan_rows<1>(rowp, pitch_px, hs, he, rowidx, width, maxval, maxidx); break;
                  case 2: pool_scan_rows<2>(rowp, pitch_px, hs, he, rowidx, width, maxval, maxidx); break;
                  case 3: pool_scan_rows<3>(rowp, pitch_px, hs, he, rowidx, width, maxval, maxidx); break;
                  case 4: pool_scan_rows<4>(rowp, pitch_px, hs, he, rowidx, width, maxval, maxidx); break;
                  case 5: pool_scan_rows<5>(rowp, pitch_px, hs, he, rowidx, width, maxval, maxidx); break;
                  case 6: pool_scan_rows<6>(rowp, pitch_px, hs, he, rowidx, width, maxval, maxidx); break;
                  case 7: pool_scan_rows<7>(rowp, pitch_px, hs, he, rowidx, width, maxval, maxidx); break;
                  default: pool_scan_rows<8>(rowp, pitch_px, hs, he, rowidx, width, maxval, maxidx); break;
                }
              } else if (staged) {
                const float* rowp = img + cl * kPoolPlane + (hs - r0) * pitch_px + (ws - wlo);
                int rowidx = bottom_data_offset + hs * width + ws;
                for (int h = hs; h < he; ++h) {
                  for (int w = 0; w < ncol; ++w) {
                    const float v = rowp[w];
                    if (v > maxval) {  // :83 strict >: the first maximum in row-major order wins
                      maxval = v;
                      maxidx = rowidx + w;
                    }
                  }
                  rowp += pitch_px;
                  rowidx += width;
                }
              } else {
                const float* rowp = src + (long long)cl * plane_px + (long long)hs * width + ws;
                int rowidx = bottom_data_offset + hs * width + ws;
                for (int h = hs; h < he; ++h) {
                  for (int w = 0; w < ncol; ++w) {
                    const float v = rowp[w];
                    if (v > maxval) {
                      maxval = v;
                      maxidx = rowidx + w;
                    }
                  }
                  rowp += width;
                  rowidx += width;
                }
              }
              tval[cl * ts + b] = maxval;
              targ[cl * ts + b] = maxidx;
            }
          }
      }
    __syncthreads();  // the tile is complete
    // ---- [channel][bins of the group] leave as contiguous runs (the whole [32][bins] block when the group is the RoI) ----
    float* __restrict__ dst = top_data + ((long long)r * channels + c0) * bins + pa * pooled_width;
    int32_t* __restrict__ adst = argmax_data != nullptr ? argmax_data + ((long long)r * channels + c0) * bins + pa * pooled_width : nullptr;
    const unsigned nb_magic = (1u << 20) / (unsigned)nb + 1u;
    for (int i = tid; i < cvalid * nb && !(ablate & 4); i += kPoolThreads) {
      const int c = nb <= 128 ? (int)(((unsigned)i * nb_magic) >> 20) : i / nb, b = i - c * nb;
      dst[(long long)c * bins + b] = tval[c * ts + b];
      if (adst != nullptr) adst[(long long)c * bins + b] = targ[c * ts + b];
    }
    __syncthreads();  // before the next group rewrites the tile
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
  // bin rows per LDS tile: the whole output when it has at most kPoolTileBins bins per channel (7 x 7), else as many rows
  // as fit (at least one: a row of pooled_width bins)
  const int rows_per_group = pooled_height * pooled_width <= kPoolTileBins ? pooled_height
                                                                           : (kPoolTileBins / pooled_width > 0 ? kPoolTileBins / pooled_width : 1);
  const size_t lds = (size_t)(2 * kPoolCT * ((rows_per_group * pooled_width) | 1) + kPoolCT * kPoolPlane +
                              2 * (pooled_height + pooled_width)) * 4;
  MI_REQUIRE(lds <= 160 * 1024 - 2048, "roi_pool: pooled_width %d needs an output tile of %zu bytes of LDS", pooled_width, lds);
  MI_REQUIRE((long long)height * width * 4 * kPoolCT < (1LL << 31), "roi_pool: a 32-channel slab of the map exceeds 2 GB");
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&roi_pool_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int tiles = (channels + kPoolCT - 1) / kPoolCT;
  roi_pool_fwd<<<num_rois * tiles, kPoolThreads, lds, mi::as_stream(stream)>>>(
      features, rois, output, argmax, batch, channels, height, width, pooled_height, pooled_width, spatial_scale,
      rows_per_group, mi::tuning().ablate);
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
