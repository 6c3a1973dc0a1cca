// roi_align_fwd_pipe.hip -- RoIAlign forward (Caffe2 semantics, roi_align_kernel.cu:65-121) as ONE persistent,
// software-pipelined workgroup per compute unit, over the records roi_align_prepare (roi_align_records.hip) leaves in
// the caller's workspace.  NCHW and channels-last features share the skeleton.
//
// Why.  roi_align_fwd_records runs a chain per (RoI, 32-channel tile) workgroup: record load -> window LDS-DMA ->
// landing -> bins -> tile -> stores, three chains per CU.  Its ablation (profiles/r02_ablation_bwd_fwd.jsonl) shows the
// phases ADD: 21 us skeleton + 10 us DMA + 12 us arithmetic = 43.5 us -- nothing overlaps, neither inside a workgroup nor
// across the three co-resident ones, which march in lock-step.  Here the phases of consecutive items overlap by
// construction:
//
//   item          = one stage of one RoI (its window fits the LDS image) x the 32-channel tile of this workgroup
//   workgroup     = 13 waves, one per CU (155 KB of LDS), bound to one channel tile (== one XCD when C = 256, so a
//                   tile's slab of the feature maps is served by one L2); items are drawn in sweep order
//   every wave    walks the workgroup's run of the sweep itself (rank, stage; one LDS read of the record header per step)
//   waves 0..3    "loaders" (one per SIMD): in iteration k they issue the window LDS-DMA of item k+2 into image buffer
//                   (k+2) % 3, 8 channels each, then wait until at most those pieces are outstanding (vmcnt retires in
//                   order: item k+1 has landed)
//   waves 4..10   "bin waves": the bins of item k from image buffer k % 3 into output tile k & 1; a half-wave owns an
//                   output column (x entries read once per item), rows two at a time
//   wave 11       "storer": copies output tile (k-1) & 1 to global memory while item k is computed; never waits for a store
//   wave 12       "agent": copies record headers and axis tables record -> LDS two RoIs ahead of the window stream, by
//                   LDS-DMA with counted waits, so that no wave ever waits for a record and the records' latencies
//                   never touch the workers' vmcnt.  roi_align_prepare cuts the sweep into one contiguous, cost-balanced
//                   run per workgroup (windows differ 20x in size).  Measured first and dropped: a ticket counter (the
//                   returning atomic + dependent header load put 2 us on every iteration) and an agent that published
//                   per-item descriptors (a wave issues one instruction per 4 clocks: 250 scalar instructions per item
//                   made the agent, not the data movement, the pace of the pipeline), and 14 symmetric worker waves that
//                   each issued a share of the window and computed two bins (every wave repeated the per-item set-up:
//                   the four SIMDs were VALU-issue bound at 1.3 us per item; per-wave stamps, tools/timeline_pipe.py).
//   one s_barrier per item separates the iterations; two window DMAs are in flight while a third item is computed, so the
//   L2 -> LDS stream does not stop at item boundaries.
//
// Arithmetic, table format, border taps and the reference-order path for RoIs the tables cannot describe are those of
// roi_align_fwd_records (same operation order: the two kernels agree bit for bit on the fast path).
#include "common.h"
#include "roi_align_device.h"
#include "lds_dma.h"
#include "roi_align_record_layout.h"

#include <algorithm>
#include <type_traits>

namespace mi {
namespace {

constexpr int kCT = 32;                  // channels per workgroup
constexpr int kLoaders = 4;              // waves 0..3 (one per SIMD): window LDS-DMA, 8 channels each
constexpr int kBinWaves = 7;             // waves 4..10: the bins
constexpr int kStorer = kLoaders + kBinWaves, kAgent = kStorer + 1;
constexpr int kWaves = kAgent + 1, kThreads = kWaves * 64;
constexpr int kSlots = kBinWaves * 2;    // half-waves of the bin waves
constexpr int kTileBins = 56;            // output bins per channel of one item (roi_align_prepare cuts stages accordingly)
constexpr int kCap = 336;                // window pixels per channel of one image buffer (== the stages' cap)
constexpr int kPlane = kCap | 1;         // NCHW: odd plane stride (words)
constexpr int kImgWords = kCT * kPlane;  // channels-last uses kCap * 32 of them
constexpr int kTileWords = kCT * (kTileBins + 1);
constexpr int kImgBufs = 3;
constexpr int kHdrSlots = 8, kHdrDw = 192;  // record headers + stage lists held in LDS (slot = rank & 7)
constexpr int kTabSlots = 8;             // axis tables held in LDS (slot = rank & 7)
constexpr int kAhead = 2;                // the agent fetches header and tables two RoIs ahead of the window stream

struct TabEntry {
  int off;
  float hw, lw;
  int lo;
};

constexpr size_t pipe_lds_bytes() {
  return (size_t)kHdrSlots * kHdrDw * 4 + (size_t)kTabSlots * 2 * kMaxS * sizeof(TabEntry) +
         (size_t)(2 * kTileWords + kImgBufs * kImgWords) * 4;
}

template <bool kNHWC>
__device__ __forceinline__ void lds_pair(unsigned a, float& v0, float& v1) {
  const lds_cfloat_t q = (lds_cfloat_t)(uintptr_t)a;
  v0 = q[0];
  v1 = q[kNHWC ? kCT : 1];
}

// wait until at most n vector-memory operations of this wave are outstanding (they retire in order)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
  switch (n) {
#define MI_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    MI_VMCNT_CASE(0) MI_VMCNT_CASE(1) MI_VMCNT_CASE(2) MI_VMCNT_CASE(3) MI_VMCNT_CASE(4) MI_VMCNT_CASE(5)
    MI_VMCNT_CASE(6) MI_VMCNT_CASE(7) MI_VMCNT_CASE(8) MI_VMCNT_CASE(9) MI_VMCNT_CASE(10) MI_VMCNT_CASE(11)
    MI_VMCNT_CASE(12) MI_VMCNT_CASE(13) MI_VMCNT_CASE(14) MI_VMCNT_CASE(15)
    default:  // above 15 in steps of 8, rounded down: stricter than asked for, safe
      switch (n >> 3) {
        case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
        case 6: case 7: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      }
      break;
#undef MI_VMCNT_CASE
  }
}

__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// An item = (rank of the RoI along the sweep, stage).  Every wave walks the workgroup's run of the sweep itself: the
// only thing it needs from the record to step is the number of stages.
struct Item {
  int pos, stage;
};
__device__ __forceinline__ void next_item(Item& it, int flags, int nstages) {
  if ((flags & kFlagFast) && it.stage + 1 < nstages) {
    it.stage++;
  } else {
    it.pos++;
    it.stage = 0;
  }
}

// kSR > 0: sampling_ratio == kSR at compile time.  kA > 0: aligned_height == aligned_width == kA at compile time.
template <int kSR, int kA, bool kNHWC>
__global__ void __launch_bounds__(kThreads)
roi_align_fwd_pipe(const LevelTable lv, const float* __restrict__ rois, float* __restrict__ out, int* __restrict__ ws,
                   int num_rois, int batch, int channels, int aligned_height_arg, int aligned_width_arg,
                   int sampling_ratio, int ablate_arg, long long* __restrict__ timeline) {
  const int aligned_height = kA > 0 ? kA : aligned_height_arg, aligned_width = kA > 0 ? kA : aligned_width_arg;
  const int ablate = MI_ABLATE(ablate_arg);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* const hdrs = reinterpret_cast<int*>(smem);
  const unsigned hdrs_addr = lds_addr_uniform(hdrs);
  TabEntry* const tabs = reinterpret_cast<TabEntry*>(hdrs + kHdrSlots * kHdrDw);
  const unsigned tabs_addr = lds_addr_uniform(tabs);
  float* const tiles = reinterpret_cast<float*>(tabs + kTabSlots * 2 * kMaxS);
  // the image buffers are only ever addressed by LDS byte address (DMA destinations, tap reads): integer arithmetic
  // on the base keeps hipcc from routing them through generic pointers
  const unsigned imgs_addr = lds_addr_uniform(tiles + 2 * kTileWords);

  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int bins = aligned_height * aligned_width;
  const int ntiles = channels / kCT;
  const int tile_id = (int)blockIdx.x % ntiles, wg = (int)blockIdx.x / ntiles;
  const int c0 = tile_id * kCT;
  const int* __restrict__ records = ws + kCounterDwords;
  // tuning aid (tools/timeline_pipe.py): shader-clock stamps of iterations 0..23 of a workgroup, 48 slots each: slot w <
  // 16: wave w reaches the barrier; 16 + w: wave w leaves it; 32..: loader 0 (32 top, 33 windows issued), storer (35 top,
  // 36 stored), agent (37 top, 38 fetches issued, 39 waited), first bin wave (40 top, 41 bins done); null in normal operation
  const auto stamp = [&](int it, int slot) {
    if (timeline != nullptr && lane == 0 && it >= 0 && it < 24)
      timeline[((long long)blockIdx.x * 24 + it) * 48 + slot] = (long long)clock64();
  };
  // this workgroup's run of the sweep (roi_align_prepare's chunk table)
  const const_int_ptr chunk_start = (const_int_ptr)(uintptr_t)(ws + kChunkBase);
  const int run_begin = chunk_start[wg], run_end = chunk_start[wg + 1];
  // lane i < kRecHeader: header dword i of the item's record; lanes kRecHeader .. + 3: its stage entry
  const auto item_words = [&](const Item& it) -> int {
    const int idx = lane < kRecHeader ? lane : kRecStages + 4 * it.stage + ((lane - kRecHeader) & 3);
    return hdrs[(it.pos & (kHdrSlots - 1)) * kHdrDw + idx];
  };
  // agent: header + stage list (192 dwords) and the two axis tables (256 dwords) of a record -> LDS slots rank & 7
  const auto fetch = [&](int pos) {
    const srd_t rsrd = make_srd(records + (long long)pos * kRecDwords, (unsigned)kRecDwords * 4u);
    const unsigned hdst = hdrs_addr + (unsigned)((pos & (kHdrSlots - 1)) * kHdrDw) * 4u;
    const unsigned tdst = tabs_addr + (unsigned)((pos & (kTabSlots - 1)) * 2 * kMaxS) * (unsigned)sizeof(TabEntry);
#pragma unroll
    for (int i = 0; i < kHdrDw / 64; i++) dma_dword(rsrd, hdst + (unsigned)i * 256u, (unsigned)(i * 64 + lane) * 4u, 0u);
#pragma unroll
    for (int i = 0; i < 4; i++) dma_dword(rsrd, tdst + (unsigned)i * 256u, (unsigned)(kRecY + i * 64 + lane) * 4u, 0u);
  };
  constexpr int kFetchOps = kHdrDw / 64 + 4;
  if (wave == kAgent) {
    for (int i = 0; i <= kAhead; i++)
      if (run_begin + i < run_end) fetch(run_begin + i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  wg_barrier();

  // issue priority storer / agent > loaders > bin waves: the two single waves are the youngest of their SIMDs and starve
  // behind the bin waves' VALU stream otherwise (measured: 62 -> 55 us per config-2 call in the tuning build)
  if (wave >= kStorer) __builtin_amdgcn_s_setprio(3);
  else if (wave < kLoaders) __builtin_amdgcn_s_setprio(2);
  Item win = {run_begin, 0};   // item k + 2: the window being fetched (workers), the fetch frontier (agent)
  Item cur = {run_begin, 0};   // item k: the bins being computed; everybody follows it to know when to stop
  int st_flags = 0, st_pp = 0, st_roi = 0;  // storer: item k - 1
  int b2 = 0;  // image buffer of item k + 2 (item i lives in buffer i % 3)
  for (int k = -2;; k++) {
    // ---- storer: output tile of item k - 1 -> global memory ----
    if (wave == kStorer && k >= 1) {
      stamp(k, 35);
      if ((st_flags & kFlagFast) && !(ablate & 4)) {
        const int ph0 = st_pp & 0xffff, ph1 = st_pp >> 16;
        const int nb = (ph1 - ph0) * aligned_width, ts = nb | 1;
        float* __restrict__ dst = out + ((long long)st_roi * channels + c0) * bins;
        const float* __restrict__ tile = tiles + ((k - 1) & 1) * kTileWords;
        if (nb == bins && ts == nb && ((kCT * nb) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
          // the whole [32][bins] block is contiguous on both sides: all reads first, then all stores
          const float4* t4 = reinterpret_cast<const float4*>(tile);
          float4* d4 = reinterpret_cast<float4*>(dst);
          const int n4 = kCT * nb / 4;  // <= 448
          float4 v[7];
#pragma unroll
          for (int j = 0; j < 7; j++) v[j] = t4[min(lane + 64 * j, n4 - 1)];
          if (timeline != nullptr) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(k, 42);
          }
#pragma unroll
          for (int j = 0; j < 7; j++)
            if (lane + 64 * j < n4) d4[lane + 64 * j] = v[j];
        } else {
          float* gdst = dst + ph0 * aligned_width;
          const unsigned nb_magic = (1u << 20) / (unsigned)nb + 1u;
          for (int i0 = lane; i0 < kCT * nb; i0 += 64 * 7) {
            float v[7];
            long long o[7];
#pragma unroll
            for (int j = 0; j < 7; j++) {
              const int i = min(i0 + 64 * j, kCT * nb - 1);
              const int c = (int)(((unsigned)i * nb_magic) >> 20), b = i - c * nb;
              o[j] = (long long)c * bins + b;
              v[j] = tile[c * ts + b];
            }
#pragma unroll
            for (int j = 0; j < 7; j++)
              if (i0 + 64 * j < kCT * nb) gdst[o[j]] = v[j];
          }
        }
      }
      stamp(k, 36);
    }
    // ---- everybody: item k ----
    int hc = 0, cur_flags = 0;
    if (k >= 0) {
      if (cur.pos >= run_end) break;
      hc = item_words(cur);
      cur_flags = __builtin_amdgcn_readlane(hc, 0);
      if (wave == kStorer) {
        st_flags = cur_flags;
        st_pp = __builtin_amdgcn_readlane(hc, kRecHeader);
        st_roi = __builtin_amdgcn_readlane(hc, 8);
      }
    }

    if (wave == kAgent) {
      // ---- agent: follow the window stream; when it enters a new RoI, fetch the record two RoIs further on ----
      stamp(k, 37);
      int issued = 0;
      if (win.pos < run_end) {
        const int hw = item_words(win);
        const int before = win.pos;
        next_item(win, __builtin_amdgcn_readlane(hw, 0), __builtin_amdgcn_readlane(hw, 5));
        stamp(k, 43);
        if (win.pos != before && win.pos + kAhead < run_end) {
          fetch(win.pos + kAhead);
          issued = kFetchOps;
        }
      }
      stamp(k, 38);
      // everything issued before this iteration has landed: what was fetched in iteration k - 1 is visible to the other
      // waves after this barrier, an iteration before the window stream can reach it
      wait_vmcnt_le(issued);
      stamp(k, 39);
    } else if (wave < kLoaders) {
      // ---- loaders: window of item k + 2 -> image buffer (k + 2) % 3 by LDS-DMA, 8 channels per wave ----
      if (wave == 0) stamp(k, 32);
      int issued = 0;
      if (win.pos < run_end) {
        const int hw = item_words(win);
        const int flags = __builtin_amdgcn_readlane(hw, 0);
        if ((flags & kFlagFast) && !(ablate & 1)) {
          const unsigned height = (unsigned)__builtin_amdgcn_readlane(hw, 18), width = (unsigned)__builtin_amdgcn_readlane(hw, 19);
          const unsigned wx0 = (unsigned)__builtin_amdgcn_readlane(hw, 2), ww = (unsigned)__builtin_amdgcn_readlane(hw, 3);
          const unsigned magic = (unsigned)__builtin_amdgcn_readlane(hw, 4);
          const unsigned row0 = (unsigned)__builtin_amdgcn_readlane(hw, kRecHeader + 1);
          const unsigned npx = (unsigned)__builtin_amdgcn_readlane(hw, kRecHeader + 2) * ww;
          const uintptr_t base = ((uintptr_t)(unsigned)__builtin_amdgcn_readlane(hw, 17) << 32) |
                                 (unsigned)__builtin_amdgcn_readlane(hw, 16);  // channel 0 of the RoI's image
          const unsigned img = imgs_addr + (unsigned)(b2 * kImgWords) * 4u;
          if constexpr (!kNHWC) {
            // A window row lies in LDS on a pitch of whole 16-byte groups.  Lanes are flattened over the window's (row,
            // group): one piece moves 64 groups = 256 pixels of one channel (buffer_load_dwordx4 ... lds; neither side
            // needs more than dword alignment), a quarter of the instructions of a dword copy -- a wave may have 63 loads
            // in flight, and with dword pieces two items of 8 channels do not fit.  The per-lane source offset is computed
            // once per piece (24-bit multiplies: full rate) and reused for the wave's channels.  A window that touches
            // the map's right edge (its last group would run into the next row, and the column one past the map has to
            // read the border pixel again) is copied pixel by pixel on the same pitch.
            constexpr int kCh = kCT / kLoaders;
            const unsigned plane_bytes = height * width * 4u;
            const srd_t srd = make_srd(reinterpret_cast<const char*>(base) + (size_t)(c0 + wave * kCh) * plane_bytes,
                                       (unsigned)kCh * plane_bytes);
            const unsigned pitch_px = (ww + 3u) & ~3u, nrows = (unsigned)__builtin_amdgcn_readlane(hw, kRecHeader + 2);
            const unsigned dst0 = img + (unsigned)(wave * kCh * kPlane) * 4u;
            if (wave == 0) stamp(k, 44);
            if (wx0 + pitch_px <= width) {
              const unsigned gpr = pitch_px >> 2, groups = nrows * gpr, gmagic = (unsigned)__builtin_amdgcn_readlane(hw, 20);
              const int pieces = (int)((groups + 63u) >> 6);
              issued = kCh * pieces;
              for (int kk = 0; kk < pieces; kk++) {
                const unsigned g = (unsigned)(kk * 64 + lane);
                const unsigned q = __umul24(g, gmagic) >> 20;  // g / gpr
                const unsigned gc = g - __umul24(q, gpr);
                const unsigned voff = (__umul24(min(row0 + q, height - 1u), width) + wx0 + gc * 4u) * 4u;
                if (g < groups) {
#pragma unroll
                  for (int c = 0; c < kCh; c++)
                    dma_dwordx4(srd, dst0 + (unsigned)(c * kPlane + kk * 256) * 4u, voff, (unsigned)c * plane_bytes);
                }
              }
            } else {
              const unsigned npp = nrows * pitch_px, pmagic = (unsigned)__builtin_amdgcn_readlane(hw, 21);
              const int pieces = (int)((npp + 63u) >> 6);
              issued = kCh * pieces;
              for (int kk = 0; kk < pieces; kk++) {
                const unsigned p = (unsigned)(kk * 64 + lane);
                const unsigned q = __umul24(p, pmagic) >> 20;  // p / pitch
                const unsigned col = p - __umul24(q, pitch_px);
                // the window may end one row / column past the map (border samples): those read the last one again
                const unsigned voff = (__umul24(min(row0 + q, height - 1u), width) + min(wx0 + col, width - 1u)) * 4u;
                if (p < npp) {
#pragma unroll
                  for (int c = 0; c < kCh; c++)
                    dma_dword(srd, dst0 + (unsigned)(c * kPlane + kk * 64) * 4u, voff, (unsigned)c * plane_bytes);
                }
              }
            }
          } else {
            // channels-last: a pixel's 32 channels are one 128-byte line; unit = 8 pixels, lane = (pixel, 16-byte piece)
            const srd_t srd = make_srd(reinterpret_cast<const float*>(base) + c0, height * width * (unsigned)channels * 4u);
            const int units = (int)((npx + 7u) >> 3);
            for (int u = wave; u < units; u += kLoaders) {
              const unsigned p = (unsigned)(u * 8 + (lane >> 3));
              const unsigned q = __umul24(p, magic) >> 20;
              const unsigned col = p - __umul24(q, ww);
              const unsigned voff = ((__umul24(min(row0 + q, height - 1u), width) + min(wx0 + col, width - 1u)) * (unsigned)channels +
                                     (unsigned)(lane & 7) * 4u) * 4u;
              if (p < npx) dma_dwordx4(srd, img + (unsigned)(u * 8 * kCT) * 4u, voff, 0u);
              issued++;
            }
          }
        }
        next_item(win, flags, __builtin_amdgcn_readlane(hw, 5));
      }
      if (wave == 0) stamp(k, 33);
      wait_vmcnt_le(uniform(issued));  // everything before the pieces of item k + 2 has landed: item k + 1 is complete
    } else if (wave < kStorer) {
      // ---- bin waves: item k from image buffer k % 3 into output tile k & 1 ----
      if (k >= 0) {
        if (wave == kLoaders) stamp(k, 40);
        const int pp = __builtin_amdgcn_readlane(hc, kRecHeader);
        const int ph0 = pp & 0xffff, ph1 = pp >> 16;
        const int btid = tid - kLoaders * 64;
        if ((cur_flags & kFlagFast) && !(ablate & 2)) {
          const int ww = __builtin_amdgcn_readlane(hc, 3), row0 = __builtin_amdgcn_readlane(hc, kRecHeader + 1);
          const int gh = kSR > 0 ? kSR : __builtin_amdgcn_readlane(hc, 6), gw = kSR > 0 ? kSR : __builtin_amdgcn_readlane(hc, 7);
          const int cl = btid & (kCT - 1), slot = btid >> 5;
          const TabEntry* ty = tabs + (cur.pos & (kTabSlots - 1)) * 2 * kMaxS;
          const TabEntry* tx = ty + kMaxS;
          const unsigned img = imgs_addr + (unsigned)((b2 == 2 ? 0 : b2 + 1) * kImgWords) * 4u;  // item k: buffer (b2 + 1) % 3
          float* tile = tiles + (k & 1) * kTileWords;
          constexpr int kPx = kNHWC ? kCT : 1;  // byte scale of a table offset (tables count bytes of a 1-channel window)
          unsigned img_c = img + (unsigned)(kNHWC ? cl : cl * kPlane) * 4u;
          asm volatile("" : "+v"(img_c));
          // NCHW: rows on a pitch of whole 16-byte groups (see the loaders); channels-last: pixels, 128 bytes each
          const int pitch = kNHWC ? ww * 4 * kPx : ((ww + 3) & ~3) * 4;
          const int base_off = row0 * pitch;
          const int nrow = ph1 - ph0, nb = nrow * aligned_width, ts = nb | 1;
          if constexpr (kSR > 0 && (kA == 7 || kA == 14)) {
            constexpr int kS = kSR;
            // a half-wave owns one output column: pw = slot (14 columns), or (pw, upper / lower half of the stage's bin
            // rows) for 7 columns -- the x entries are read once per item, rows go two at a time (tables, then all tap
            // pairs, then the FMAs: two LDS round trips per pair of rows)
            const int pw = kA == 14 ? slot : (slot >= 7 ? slot - 7 : slot);
            const int first = kA == 14 ? nrow : (nrow + 1) >> 1;
            const int ra = (kA == 14 || slot < 7) ? 0 : first, rb = (kA == 14 || slot >= 7) ? nrow : first;  // rows [ra, rb)
            float hx[kS], lx[kS];
            unsigned xa[kS];
#pragma unroll
            for (int i = 0; i < kS; i++) {
              const TabEntry ex = tx[pw * kS + i];
              hx[i] = ex.hw;
              lx[i] = ex.lw;
              xa[i] = img_c + (unsigned)(ex.off * kPx - base_off);
            }
            auto rows = [&](int r0, auto kn) {
              constexpr int kN = decltype(kn)::value;
              float v[kN][kS][2][kS][2], wy[kN][kS][2];
#pragma unroll
              for (int j = 0; j < kN; j++)
#pragma unroll
                for (int iy = 0; iy < kS; iy++) {
                  const TabEntry ey = ty[(ph0 + r0 + j) * kS + iy];
                  wy[j][iy][0] = ey.hw;
                  wy[j][iy][1] = ey.lw;
                  const int yo = __mul24(ey.lo, pitch);
#pragma unroll
                  for (int ix = 0; ix < kS; ix++) {
                    const unsigned a = xa[ix] + (unsigned)yo;
                    lds_pair<kNHWC>(a, v[j][iy][0][ix][0], v[j][iy][0][ix][1]);
                    lds_pair<kNHWC>(a + (unsigned)pitch, v[j][iy][1][ix][0], v[j][iy][1][ix][1]);
                  }
                }
#pragma unroll
              for (int j = 0; j < kN; j++) {
                float acc = 0.f;
#pragma unroll
                for (int iy = 0; iy < kS; iy++) {
#pragma unroll
                  for (int kx = 0; kx < 2; kx++) {
                    float rsum = hx[0] * v[j][iy][kx][0][0];
                    rsum = __builtin_fmaf(lx[0], v[j][iy][kx][0][1], rsum);
#pragma unroll
                    for (int ix = 1; ix < kS; ix++) {
                      rsum = __builtin_fmaf(hx[ix], v[j][iy][kx][ix][0], rsum);
                      rsum = __builtin_fmaf(lx[ix], v[j][iy][kx][ix][1], rsum);
                    }
                    acc = __builtin_fmaf(wy[j][iy][kx], rsum, acc);
                  }
                }
                tile[cl * ts + (r0 + j) * aligned_width + pw] = acc;
              }
            };
            int r0 = ra;
            for (; r0 + 2 <= rb; r0 += 2) rows(r0, std::integral_constant<int, 2>());
            if (r0 < rb) rows(r0, std::integral_constant<int, 1>());
          } else {
            for (int b = slot; b < nb; b += kSlots) {
              const int phr = b / aligned_width, pw = b - phr * aligned_width, ph = ph0 + phr;
              float acc = 0.f;
              for (int iy = 0; iy < gh; iy++) {
                const TabEntry ey = ty[ph * gh + iy];
                float r0s = 0.f, r1s = 0.f;
                for (int ix = 0; ix < gw; ix++) {
                  const TabEntry ex = tx[pw * gw + ix];
                  const unsigned a = img_c + (unsigned)(ey.lo * pitch + ex.off * kPx - base_off);
                  float a0, a1, b0, b1;
                  lds_pair<kNHWC>(a, a0, a1);
                  lds_pair<kNHWC>(a + (unsigned)pitch, b0, b1);
                  r0s = __builtin_fmaf(ex.hw, a0, r0s);
                  r0s = __builtin_fmaf(ex.lw, a1, r0s);
                  r1s = __builtin_fmaf(ex.hw, b0, r1s);
                  r1s = __builtin_fmaf(ex.lw, b1, r1s);
                }
                acc = __builtin_fmaf(ey.hw, r0s, acc);
                acc = __builtin_fmaf(ey.lw, r1s, acc);
              }
              tile[cl * ts + b] = acc;
            }
          }
        } else if (!(cur_flags & kFlagFast)) {
          // ---- zero output / the reference-order path for this (RoI, channel tile), straight to global memory ----
          const int r = __builtin_amdgcn_readlane(hc, 8), lvl = __builtin_amdgcn_readlane(hc, 11);
          float* __restrict__ dst = out + ((long long)r * channels + c0) * bins;
          if (cur_flags & kFlagZero) {
            for (int i = btid; i < kCT * bins; i += kBinWaves * 64) dst[i] = 0.f;
          } else {
            const int height = lv.height[lvl], width = lv.width[lvl];
            const RoiGeom g = roi_geometry(rois + (long long)r * 5, lv.scale[lvl], aligned_height, aligned_width, sampling_ratio);
            // element strides of (channel, pixel): NCHW or channels-last
            const long long cs = kNHWC ? 1 : (long long)height * width, ps = kNHWC ? channels : 1;
            const float* src = lv.feat[lvl] + (long long)g.batch_ind * channels * height * width + c0 * cs;
            for (int i = btid; i < kCT * bins; i += kBinWaves * 64) {
              const int c = i / bins, bin = i - c * bins;
              const int ph = bin / aligned_width, pw = bin - ph * aligned_width;
              const float* plane = src + (long long)c * cs;
              float output_val = 0.f;
              for (int iy = 0; iy < g.grid_h; iy++) {
                const float y = sample_y(g, ph, iy);
                for (int ix = 0; ix < g.grid_w; ix++) {
                  const float x = sample_x(g, pw, ix);
                  const Taps t = sample_taps(height, width, y, x);
                  float val = 0.f;
                  if (t.y_low >= 0) {
                    const float v1 = plane[(t.y_low * width + t.x_low) * ps], v2 = plane[(t.y_low * width + t.x_high) * ps];
                    const float v3 = plane[(t.y_high * width + t.x_low) * ps], v4 = plane[(t.y_high * width + t.x_high) * ps];
                    val = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(t.w1, v1), __fmul_rn(t.w2, v2)), __fmul_rn(t.w3, v3)),
                                    __fmul_rn(t.w4, v4));
                  }
                  output_val = __fadd_rn(output_val, val);
                }
              }
              dst[i] = output_val / g.count;
            }
          }
        }
        if (wave == kLoaders) stamp(k, 41);
      }
    }
    if (k >= 0) next_item(cur, cur_flags, __builtin_amdgcn_readlane(hc, 5));
    stamp(k, wave);
    wg_barrier();
    stamp(k, 16 + wave);
    b2 = b2 == 2 ? 0 : b2 + 1;
  }
}

int g_num_cus = 0;
int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n <= 0)
      n = 256;
    g_num_cus = n;
  }
  return g_num_cus;
}

long long* g_pipe_timeline = nullptr;

}  // namespace

void roi_align_fwd_pipe_set_timeline(long long* device_buffer) { g_pipe_timeline = device_buffer; }

int roi_align_fwd_pipe_chunks(int channels, int num_rois) {
  // one workgroup per CU (the LDS footprint allows no more), the same number for every channel tile
  const int ntiles = channels / kCT > 0 ? channels / kCT : 1;
  return std::max(1, std::min(std::min(num_cus() / ntiles, kMaxChunks), num_rois));
}

bool roi_align_fwd_pipe_supported(int channels, int height, int width, int num_rois, int aligned_height,
                                  int aligned_width, bool nhwc) {
  const long long slab = nhwc ? (long long)height * width * channels * 4 : (long long)kCT * height * width * 4;
  return channels > 0 && channels % kCT == 0 && aligned_width <= kTileBins &&
         aligned_height > 0 && aligned_width > 0 && aligned_height <= kMaxStages && num_rois <= 8192 && slab < (1LL << 31);
}

// The records of `rois` (stages cut for a 336-pixel image, at most 56 bins each) and the chunk table for
// roi_align_fwd_pipe_chunks() workgroups must already be in `workspace`: launch_roi_align_prepare[_levels] on the same stream.
int launch_roi_align_fwd_pipe_levels(const LevelTable& lv, const float* rois, float* output, void* workspace, int batch,
                                     int channels, int num_rois, int aligned_height, int aligned_width,
                                     int sampling_ratio, bool nhwc, hipStream_t stream) {
  int* ws = static_cast<int*>(workspace);
  const int ntiles = channels / kCT;
  const int grid = roi_align_fwd_pipe_chunks(channels, num_rois) * ntiles;
  const size_t lds = pipe_lds_bytes();
#define MI_LAUNCH_PIPE(SR, A, NHWC)                                                                                   \
  do {                                                                                                                \
    static bool attr_done = false; /* benign race: the attribute is idempotent */                                     \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&roi_align_fwd_pipe<SR, A, NHWC>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    roi_align_fwd_pipe<SR, A, NHWC><<<grid, kThreads, lds, stream>>>(lv, rois, output, ws, num_rois, batch, channels, \
                                                                      aligned_height, aligned_width, sampling_ratio, \
                                                                      tuning().ablate, g_pipe_timeline);              \
  } while (0)
#define MI_LAUNCH_PIPE_L(SR, A)   \
  do {                            \
    if (nhwc)                     \
      MI_LAUNCH_PIPE(SR, A, true);  \
    else                          \
      MI_LAUNCH_PIPE(SR, A, false); \
  } while (0)
  const int a = aligned_height == aligned_width ? aligned_height : 0;
  if (sampling_ratio == 2 && a == 7)
    MI_LAUNCH_PIPE_L(2, 7);
  else if (sampling_ratio == 2 && a == 14)
    MI_LAUNCH_PIPE_L(2, 14);
  else if (sampling_ratio == 2)
    MI_LAUNCH_PIPE_L(2, 0);
  else
    MI_LAUNCH_PIPE_L(0, 0);
#undef MI_LAUNCH_PIPE_L
#undef MI_LAUNCH_PIPE
  return check_launch("roi_align_fwd_pipe");
}

}  // namespace mi
