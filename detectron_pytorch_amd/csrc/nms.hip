// nms.hip -- greedy NMS and IoU matrix for gfx950, fully on-device, C-ABI mi_nms / mi_bbox_overlaps.
//
// Arithmetic contract (fp32, IEEE divide, no FMA contraction: decisions at IoU == thresh must not
// flip):
//   MI_NMS_GE_ORIG_ASC   lib/utils/cython_nms.pyx:37-87  (areas :44, order :45, IoU :76-83, >= :84,
//                        ascending original indices :87)
//   MI_NMS_GT_SORTED_POS lib/model/nms/src/nms_cuda_kernel.cu:31-39 (devIoU), :41-85 (64x64 bitmask
//                        tiles, strict >), :132-144 (greedy OR-reduce), nms_gpu.py:7-12
//   mi_bbox_overlaps     lib/utils/cython_bbox.pyx:32-73 (fp64 intermediates, see oracle.c)
//
// Pipeline (no host synchronisation, no allocation; the reference does 2 cudaMalloc, 4 blocking
// copies and the greedy reduce on the host, nms_cuda_kernel.cu:87-161).  Latency-bound, not HBM: the input is
// 20 n bytes and the n x ceil(n/64) mask stays in L2.
//   1. nms_prepare     rank-sort by score (descending, ties -> higher index first): 64 boxes per workgroup, the four
//                      waves count over a quarter of the range each (scores broadcast from LDS, four per read), then
//                      gather boxes into sorted float4 + area arrays (GT mode: input already sorted, plain gather).
//   2. nms_mask        one wavefront per 64x64 tile of the upper triangle: lane = row box, the 64 column boxes sit in
//                      LDS; bit j of the lane's 64-bit word = IoU(row, col_j) over the threshold.  64 = wave64: one
//                      word per lane, no cross-lane traffic.  Idle lower-triangle blocks emit the TRANSPOSED diagonal
//                      tiles (who overlaps me) for step 3.
//   3. nms_reduce_regs (n <= 4096) four wavefronts walk the 64-box chunks in order; the in-chunk greedy decision is a
//                      fixpoint iteration on the transposed diagonal tile (a handful of AND + compare rounds instead
//                      of ~250 scalar-unit cycles per kept box), kept rows are OR-ed from registers loaded a chunk
//                      ahead, and (GE mode) the kept flags are compacted to ascending int64 original indices in the
//                      same launch.  nms_reduce / nms_compact are the single-wave version for n up to 16384.
// n = 2000 (RPN): 206 us -> 62 us over this round; n = 1000: 100 -> 38 us.
#include "common.h"

namespace {

constexpr int kTile = 64;             // boxes per mask word == wavefront size
constexpr int kMaxWordsPerLane = 4;   // nms_reduce keeps ceil(n/64)/64 words per lane in registers
constexpr int kMaxBoxes = kTile * kTile * kMaxWordsPerLane;  // 16384
constexpr int kCompactThreads = 1024;

struct Workspace {
  float4* boxes;     // [n_pad] sorted (x1,y1,x2,y2)
  float* areas;      // [n_pad]
  int32_t* order;    // [n_pad] sorted position -> original index
  int32_t* flags;    // [n_pad] kept flag per ORIGINAL index (GE mode)
  uint64_t* mask;    // [n, col_blocks]
  uint64_t* diag_t;  // [n_pad] transposed diagonal tiles: bit i of diag_t[j] = box i (i < j, same chunk) overlaps box j
  size_t bytes;
};

__host__ __device__ inline size_t align16(size_t b) { return (b + 15) & ~size_t(15); }

__host__ __device__ inline Workspace carve(void* base, int n) {
  Workspace w;
  const int col_blocks = (n + kTile - 1) / kTile;
  const size_t n_pad = (size_t)col_blocks * kTile;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  w.boxes = reinterpret_cast<float4*>(p + off);
  off += align16(n_pad * sizeof(float4));
  w.areas = reinterpret_cast<float*>(p + off);
  off += align16(n_pad * sizeof(float));
  w.order = reinterpret_cast<int32_t*>(p + off);
  off += align16(n_pad * sizeof(int32_t));
  w.flags = reinterpret_cast<int32_t*>(p + off);
  off += align16(n_pad * sizeof(int32_t));
  w.mask = reinterpret_cast<uint64_t*>(p + off);
  off += align16((size_t)n * col_blocks * sizeof(uint64_t));
  w.diag_t = reinterpret_cast<uint64_t*>(p + off);
  off += align16(n_pad * sizeof(uint64_t));
  w.bytes = off;
  return w;
}

// ---- 1. sort + gather -------------------------------------------------------------------
// Rank sort by counting.  A 256-lane workgroup ranks 64 boxes: lane & 63 = box, the 4 waves each count over a
// quarter of the comparison range (scores staged through LDS 2048 at a time, every lane of a wave reads the same
// word -> broadcast), the partial ranks meet in LDS.  n = 2000: 32 workgroups x 500 comparisons per lane
// (the first version ranked 256 boxes per workgroup over the whole range: 8 workgroups on a 256-CU chip, 55 us).
constexpr int kPrepChunk = 2048;

// Where the rows of one problem live.  mi_nms / mi_nms_batched: the packed [n,5] array (row stride 5, score at +4, no
// threshold).  mi_nms_segmented: the caller's [R,4C] box and [R,C] score matrices read in place (row strides 4C and C),
// with the rows at or below `thresh` dropped: they sort behind every live row and are not counted in *n_live.
template <bool kPacked>
struct DetSource {
  const float* box;
  const float* score;
  long long box_row_, score_row_;  // strides in floats (ignored when kPacked: the constant 5 keeps the address arithmetic
  float thresh;                    // of the latency-critical single-problem path what it was)
  // rows with score <= thresh (or NaN) take no part; thresh -inf: every row takes part
  __device__ __forceinline__ long long box_row() const { return kPacked ? 5 : box_row_; }
  __device__ __forceinline__ long long score_row() const { return kPacked ? 5 : score_row_; }
  __device__ __forceinline__ float score_at(int i) const {
    const float v = score[(long long)i * score_row()];
    // a NaN score (a diverged network) must not break the rank sort: it compares as the lowest score, ties by index, so
    // that the ranks stay a permutation and nothing downstream indexes with an uninitialised slot
    return (v > thresh) ? v : -__builtin_inff();
  }
};

template <bool kSort, bool kCount, bool kPacked>
__device__ __forceinline__ void prepare_body(const DetSource<kPacked> src, int n, float4* __restrict__ boxes,
                                             float* __restrict__ areas, int32_t* __restrict__ order,
                                             int32_t* __restrict__ flags, int chunk, int32_t* __restrict__ n_live) {
  __shared__ __attribute__((aligned(16))) float s_scores[kPrepChunk];
  __shared__ int s_rank[4][kTile];
  __shared__ int s_live[4];
  const int tid = threadIdx.x, bi = tid & (kTile - 1), part = tid >> 6;
  const int i = chunk * kTile + bi;
  const bool live = i < n;
  float x1 = 0, y1 = 0, x2 = 0, y2 = 0, score = 0;
  if (live) {
    const float* d = src.box + (long long)i * src.box_row();
    x1 = d[0];
    y1 = d[1];
    x2 = d[2];
    y2 = d[3];
    score = src.score_at(i);
  }
  int rank = i;
  if (kSort) {
    int cnt = 0, live_rows = 0;
    for (int base = 0; base < n; base += kPrepChunk) {
      const int lim = min(kPrepChunk, n - base);
      const int lim4 = (lim + 15) & ~15;  // padded with NaN (never greater, never equal): four scores per LDS read
      __syncthreads();
      for (int t = tid; t < lim4; t += 256)
        if (t < lim) {
          const float v = src.score_at(base + t);
          s_scores[t] = v;
          if (kCount) live_rows += v > -__builtin_inff();
        } else {
          s_scores[t] = __builtin_nanf("");
        }
      __syncthreads();
      const int per = lim4 / 4;  // a multiple of 4
      const int t0 = part * per, t1 = t0 + per;
      if (live) {
        for (int t = t0; t < t1; t += 4) {
          const float4 s4 = *reinterpret_cast<const float4*>(&s_scores[t]);
          const int jj = base + t;
          // descending score; equal scores: higher original index first
          // (== np.argsort(scores, kind='stable')[::-1], the tie rule fixed in oracle.c)
          cnt += (s4.x > score) || (s4.x == score && jj > i);
          cnt += (s4.y > score) || (s4.y == score && jj + 1 > i);
          cnt += (s4.z > score) || (s4.z == score && jj + 2 > i);
          cnt += (s4.w > score) || (s4.w == score && jj + 3 > i);
        }
      }
    }
    s_rank[part][bi] = cnt;
    if (kCount) {   // every workgroup of the problem sees all scores: the first one publishes the live-row count
#pragma unroll
      for (int d = kTile / 2; d > 0; d >>= 1) live_rows += __shfl_xor(live_rows, d, kTile);
      if (bi == 0) s_live[part] = live_rows;
    }
    __syncthreads();
    rank = s_rank[0][bi] + s_rank[1][bi] + s_rank[2][bi] + s_rank[3][bi];
    if (kCount && chunk == 0 && tid == 0) *n_live = (s_live[0] + s_live[1]) + (s_live[2] + s_live[3]);
  }
  if (live && part == 0) {
    boxes[rank] = make_float4(x1, y1, x2, y2);
    areas[rank] = (x2 - x1 + 1.f) * (y2 - y1 + 1.f);  // cython_nms.pyx:44 / nms_cuda_kernel.cu:36-37
    order[rank] = i;
    flags[i] = 0;
  }
}

__device__ __forceinline__ DetSource<true> packed_source(const float* dets) {
  return DetSource<true>{dets, dets + 4, 5, 5, -__builtin_inff()};
}

template <bool kSort>
__global__ void __launch_bounds__(256)
nms_prepare(const float* __restrict__ dets, int n, float4* __restrict__ boxes,
            float* __restrict__ areas, int32_t* __restrict__ order, int32_t* __restrict__ flags) {
  prepare_body<kSort, false, true>(packed_source(dets), n, boxes, areas, order, flags, blockIdx.x, nullptr);
}

// ---- 2. IoU bitmask tiles ---------------------------------------------------------------
template <bool kGE>
__device__ __forceinline__ bool overlaps(const float4 a, const float area_a, const float4 b,
                                         const float area_b, const float thresh) {
  float xx1, yy1, xx2, yy2;
  if (kGE) {  // cython_nms.pyx:28-32 inline max/min
    xx1 = a.x >= b.x ? a.x : b.x;
    yy1 = a.y >= b.y ? a.y : b.y;
    xx2 = a.z <= b.z ? a.z : b.z;
    yy2 = a.w <= b.w ? a.w : b.w;
  } else {  // nms_cuda_kernel.cu:32-33
    xx1 = fmaxf(a.x, b.x);
    yy1 = fmaxf(a.y, b.y);
    xx2 = fminf(a.z, b.z);
    yy2 = fminf(a.w, b.w);
  }
  float w = (xx2 - xx1) + 1.f;  // pyx:80-81 / cu:34
  float h = (yy2 - yy1) + 1.f;
  w = w >= 0.f ? w : 0.f;
  h = h >= 0.f ? h : 0.f;
  const float inter = w * h;                              // pyx:82
  const float ovr = inter / ((area_a + area_b) - inter);  // pyx:83 / cu:38, IEEE divide
  return kGE ? (ovr >= thresh) : (ovr > thresh);          // pyx:84 / cu:78
}

template <bool kGE>
__device__ __forceinline__ void mask_body(const float4* __restrict__ boxes, const float* __restrict__ areas, int n,
                                          float thresh, uint64_t* __restrict__ mask, uint64_t* __restrict__ diag_t,
                                          int col_start, int row_start, int col_blocks) {
  __shared__ float4 s_box[kTile];
  __shared__ float s_area[kTile];
  const int lane = threadIdx.x;
  // The lower triangle is never read by the reduce (cu:139 starts at nblock).  Its block (row-1, row) computes the
  // TRANSPOSED diagonal tile of chunk `row` instead (chunk 0: block (0,0) does both), which the reduce's fixpoint
  // iteration needs: bit i of diag_t[j] = box i (i < j, same chunk) overlaps box j.  The IoU test is symmetric bit
  // for bit (max/min and the fp32 sum of the two areas commute), so this is the transpose of the diagonal tile.
  // chunk 0 has no block to its left: block (0, 2) takes it when the grid has one, else block (0, 0) does both
  const bool t0_idle = col_blocks >= 3 && col_start == 0 && row_start == 2;
  const bool t0_self = col_blocks < 3 && col_start == 0 && row_start == 0;
  const bool transposed = col_start + 1 == row_start || t0_idle;
  if (col_start < row_start && !transposed) return;
  if (transposed || t0_self) {
    const int chunk = t0_idle ? 0 : row_start;
    const int size = min(n - chunk * kTile, kTile);
    if (lane < size) {
      s_box[lane] = boxes[chunk * kTile + lane];
      s_area[lane] = areas[chunk * kTile + lane];
    }
    __syncthreads();
    if (lane < size) {
      const float4 a = s_box[lane];
      const float area_a = s_area[lane];
      uint64_t tt = 0;
      for (int j = 0; j < lane; j++)
        if (overlaps<kGE>(s_box[j], s_area[j], a, area_a, thresh)) tt |= 1ULL << j;
      diag_t[chunk * kTile + lane] = tt;
    }
    if (transposed) return;
    __syncthreads();
  }
  const int col_size = min(n - col_start * kTile, kTile);
  const int row_size = min(n - row_start * kTile, kTile);
  if (lane < col_size) {
    s_box[lane] = boxes[col_start * kTile + lane];
    s_area[lane] = areas[col_start * kTile + lane];
  }
  __syncthreads();
  if (lane < row_size) {
    const int cur = row_start * kTile + lane;
    const float4 a = boxes[cur];
    const float area_a = areas[cur];
    uint64_t t = 0;
    const int start = (row_start == col_start) ? lane + 1 : 0;  // cu:73-76
    for (int j = start; j < col_size; j++)
      if (overlaps<kGE>(a, area_a, s_box[j], s_area[j], thresh)) t |= 1ULL << j;
    mask[(long long)cur * col_blocks + col_start] = t;
  }
}

template <bool kGE>
__global__ void __launch_bounds__(kTile)
nms_mask(const float4* __restrict__ boxes, const float* __restrict__ areas, int n, float thresh,
         uint64_t* __restrict__ mask, uint64_t* __restrict__ diag_t) {
  mask_body<kGE>(boxes, areas, n, thresh, mask, diag_t, blockIdx.x, blockIdx.y, gridDim.x);
}

// ---- 3. greedy reduce in one wavefront ----------------------------------------------------
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane) {
  const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane);
  const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}

template <int kWords, bool kGE>
__global__ void __launch_bounds__(kTile)
nms_reduce(const uint64_t* __restrict__ mask, int n, const int32_t* __restrict__ order,
           int32_t* __restrict__ flags, int32_t* __restrict__ keep32,
           int32_t* __restrict__ num_keep) {
  const int lane = threadIdx.x;
  const int col_blocks = (n + kTile - 1) / kTile;
  uint64_t remv[kWords];
#pragma unroll
  for (int s = 0; s < kWords; s++) remv[s] = 0;
  int count = 0;
  for (int k = 0; k < col_blocks; k++) {
    // removed-word of this chunk: owned by lane k % 64, slot k / 64
    uint64_t owned = remv[0];
#pragma unroll
    for (int s = 1; s < kWords; s++)
      if (s == k / kTile) owned = remv[s];
    uint64_t cur = readlane64(owned, k % kTile);
    const int row = k * kTile + lane;
    const uint64_t diag = (row < n) ? mask[(long long)row * col_blocks + k] : 0ULL;
    const int live = n - k * kTile;
    const uint64_t valid = live >= kTile ? ~0ULL : ((1ULL << live) - 1ULL);
    // in-chunk greedy pass (cu:132-144 restricted to word k), wave-uniform -> scalar unit
    uint64_t keepbits = 0;
    uint64_t cand = ~cur & valid;
    while (cand) {
      const int i = __builtin_ctzll(cand);
      keepbits |= 1ULL << i;
      cur |= readlane64(diag, i);
      const uint64_t upto = (i == 63) ? ~0ULL : ((2ULL << i) - 1ULL);
      cand = ~cur & valid & ~upto;
    }
    // OR the kept rows into the words this lane owns (only words > k matter from here on)
    uint64_t todo = keepbits;
    while (todo) {
      int idx[4];
      bool on[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        on[u] = todo != 0;
        idx[u] = on[u] ? __builtin_ctzll(todo) : 0;
        if (on[u]) todo &= todo - 1;
      }
#pragma unroll
      for (int s = 0; s < kWords; s++) {
        const int w = s * kTile + lane;
        if (w > k && w < col_blocks) {
          uint64_t m[4];
#pragma unroll
          for (int u = 0; u < 4; u++)
            m[u] = on[u] ? mask[(long long)(k * kTile + idx[u]) * col_blocks + w] : 0ULL;
          remv[s] |= (m[0] | m[1]) | (m[2] | m[3]);
        }
      }
    }
    // emit
    const bool mine = (keepbits >> lane) & 1ULL;
    if (mine) {
      if (kGE) {
        flags[order[row]] = 1;
      } else {
        const int pos = count + __popcll(keepbits & ((1ULL << lane) - 1ULL));
        keep32[pos] = row;
      }
    }
    count += __popcll(keepbits);
  }
  if (!kGE && lane == 0) *num_keep = count;
}

// Reduce for n <= 4096 (col_blocks <= 64, one removed-word per lane), four wavefronts.
// Lane w of every wave owns word w of the removed set; wave q accumulates the OR of the kept mask rows r = q (mod 4)
// of each chunk, so the removed set is the OR of the four waves' partial sets.  Per chunk:
//   1. the four partial words k meet in LDS (one barrier, double-buffered slots) -> cur, wave-uniform everywhere;
//   2. every wave runs the in-chunk greedy decision itself, as a fixpoint iteration on the TRANSPOSED diagonal tile:
//      kept[j] = cand[j] and no kept i < j overlaps j.  Lane j holds who overlaps it, so one round is an AND with the
//      wave-uniform kept set and a compare whose lane mask IS the next kept set.  Box j is final after j+1 rounds
//      whatever the start; the loop ends at the first repeat = (longest suppression chain in the chunk) + 1 rounds.
//      (The first version walked the kept boxes on the scalar unit: ~250 cycles per kept box, 120 us at n = 2000.)
//   3. each wave ORs its 16 rows -- loaded one chunk ahead, register selects, no dependent memory trip.
template <bool kGE>
__device__ __forceinline__ void reduce_regs_body(const uint64_t* __restrict__ mask, const uint64_t* __restrict__ diag_t,
                                                 int n, const int32_t* __restrict__ order, int32_t* __restrict__ flags,
                                                 int32_t* __restrict__ keep32, int64_t* __restrict__ keep64,
                                                 int32_t* __restrict__ num_keep) {
  constexpr int kParts = 4, kRows = kTile / kParts;
  __shared__ uint64_t s_word[2][kParts];
  const int lane = threadIdx.x & (kTile - 1);
  const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col_blocks = (n + kTile - 1) / kTile;
  const bool owner = lane < col_blocks;
  uint64_t nxt[kRows];
  uint64_t nxt_diag;
  int nxt_order = 0;
  auto fetch = [&](int k) {
    const int nrow = n - k * kTile;
#pragma unroll
    for (int i = 0; i < kRows; i++) {
      const int r = part + kParts * i;
      nxt[i] = (owner && r < nrow) ? mask[(long long)(k * kTile + r) * col_blocks + lane] : 0ULL;
    }
    nxt_diag = (lane < nrow) ? diag_t[k * kTile + lane] : 0ULL;
    if (kGE && part == 0) nxt_order = (lane < nrow) ? order[k * kTile + lane] : 0;
  };
  uint64_t remv = 0;  // this wave's partial removed set, word `lane`
  int count = 0;
  fetch(0);
  for (int k = 0; k < col_blocks; k++) {
    uint64_t v[kRows];
#pragma unroll
    for (int i = 0; i < kRows; i++) v[i] = nxt[i];
    const uint64_t diag = nxt_diag;
    const int my_order = nxt_order;
    if (k + 1 < col_blocks) fetch(k + 1);
    const uint64_t mine_k = readlane64(remv, k);
    if (lane == 0) s_word[k & 1][part] = mine_k;
    __syncthreads();
    const uint64_t cur = (s_word[k & 1][0] | s_word[k & 1][1]) | (s_word[k & 1][2] | s_word[k & 1][3]);
    const int row = k * kTile + lane;
    const int live = n - k * kTile;
    const uint64_t valid = live >= kTile ? ~0ULL : ((1ULL << live) - 1ULL);
    const uint64_t cand = ~cur & valid;
    uint64_t keepbits = cand;
    for (;;) {
      const uint64_t next = __ballot((diag & keepbits) == 0ULL) & cand;
      if (next == keepbits) break;
      keepbits = next;
    }
    // OR this wave's kept rows into its partial set (words <= k are dead from here on; harmless)
#pragma unroll
    for (int i = 0; i < kRows; i++)
      if ((keepbits >> (part + kParts * i)) & 1ULL) remv |= v[i];
    // emit (wave 0)
    if (part == 0) {
      const bool mine = (keepbits >> lane) & 1ULL;
      if (mine) {
        if (kGE) {
          flags[my_order] = 1;
        } else {
          const int pos = count + __popcll(keepbits & ((1ULL << lane) - 1ULL));
          keep32[pos] = row;
        }
      }
      count += __popcll(keepbits);
    }
  }
  if (!kGE || keep64 == nullptr) {   // GE without an index list: the caller wants the per-row kept flags only
    if (threadIdx.x == 0) *num_keep = count;
    return;
  }
  // ---- (GE mode) flags by original index -> ascending int64 indices, in the same launch: the flags were written by
  // wave 0 of this workgroup; a workgroup-scope fence + barrier orders them for the other waves of the CU ----
  __threadfence_block();
  __syncthreads();
  __shared__ int s_wave_total[4];
  const int tid = threadIdx.x;
  const int per = (n + 255) / 256;
  const int begin = min(tid * per, n), end = min(begin + per, n);
  int local = 0;
  for (int i = begin; i < end; i++) local += flags[i] != 0;
  int incl = local;
#pragma unroll
  for (int d = 1; d < kTile; d <<= 1) {
    const int up = __shfl_up(incl, d, kTile);
    if (lane >= d) incl += up;
  }
  if (lane == kTile - 1) s_wave_total[part] = incl;
  __syncthreads();
  int wave_base = 0, total = 0;
  for (int w = 0; w < 4; w++) {
    const int v = s_wave_total[w];
    if (w < part) wave_base += v;
    total += v;
  }
  int pos = wave_base + incl - local;
  for (int i = begin; i < end; i++)
    if (flags[i] != 0) keep64[pos++] = i;
  if (tid == 0) *num_keep = total;
}

template <bool kGE>
__global__ void __launch_bounds__(256)
nms_reduce_regs(const uint64_t* __restrict__ mask, const uint64_t* __restrict__ diag_t, int n,
                const int32_t* __restrict__ order, int32_t* __restrict__ flags, int32_t* __restrict__ keep32,
                int64_t* __restrict__ keep64, int32_t* __restrict__ num_keep) {
  reduce_regs_body<kGE>(mask, diag_t, n, order, flags, keep32, keep64, num_keep);
}

// ---- batched entry points: independent problems (the RPN runs one NMS per FPN level and image) in ONE launch of each
// stage.  One problem's reduce occupies a single CU and its three launches are a dependent chain, so P problems issued
// one after the other leave a 256-CU chip idle; here the P reduces run side by side. ----
constexpr int kMaxBatch = 32;
struct BatchTable {
  int count;
  int n[kMaxBatch];
  int chunk_start[kMaxBatch + 1];  // prefix sum of ceil(n / 64)
  const float* dets[kMaxBatch];
  void* keep[kMaxBatch];
  int32_t* num_keep[kMaxBatch];
  Workspace ws[kMaxBatch];
};

template <bool kSort>
__global__ void __launch_bounds__(256) nms_prepare_batched(const BatchTable t) {
  int p = 0;
  while (p + 1 < t.count && (int)blockIdx.x >= t.chunk_start[p + 1]) p++;
  prepare_body<kSort, false, true>(packed_source(t.dets[p]), t.n[p], t.ws[p].boxes, t.ws[p].areas, t.ws[p].order, t.ws[p].flags,
                             blockIdx.x - t.chunk_start[p], nullptr);
}

template <bool kGE>
__global__ void __launch_bounds__(kTile) nms_mask_batched(const BatchTable t, float thresh) {
  const int p = blockIdx.z;
  const int n = t.n[p];
  const int col_blocks = (n + kTile - 1) / kTile;
  if ((int)blockIdx.x >= col_blocks || (int)blockIdx.y >= col_blocks) return;
  mask_body<kGE>(t.ws[p].boxes, t.ws[p].areas, n, thresh, t.ws[p].mask, t.ws[p].diag_t, blockIdx.x, blockIdx.y,
                 col_blocks);
}

template <bool kGE>
__global__ void __launch_bounds__(256) nms_reduce_batched(const BatchTable t) {
  const int p = blockIdx.x;
  if (t.n[p] == 0) {
    if (threadIdx.x == 0) *t.num_keep[p] = 0;
    return;
  }
  reduce_regs_body<kGE>(t.ws[p].mask, t.ws[p].diag_t, t.n[p], t.ws[p].order, t.ws[p].flags,
                        static_cast<int32_t*>(t.keep[p]), static_cast<int64_t*>(t.keep[p]), t.num_keep[p]);
}

// ---- segmented entry point: the per-class NMS of the test-time post-processing (core/test.py:748-771) with the class
// sizes never leaving the device.  Segment s = class s + 1: its candidate rows are ALL R RoIs, read in place from the
// score / box matrices; the rows at or below the score threshold are dropped by the sort (they rank last) and the number
// of live rows -- which only the device knows -- sizes the mask and the reduce of that segment. ----
struct SegmentArgs {
  const float* boxes;
  const float* scores;
  long long box_seg, box_row, score_seg, score_row;
  int rows;
  float score_thresh;
  char* workspace;          // num_segments x seg_bytes, each carved like a single problem of `rows` boxes
  size_t seg_bytes;
  int32_t* n_live;          // [num_segments], in the workspace
  int32_t* kept;            // [num_segments, rows] caller's flags
  int32_t* num_keep;        // [num_segments]
  float* masked;            // [num_segments, rows] or nullptr: the score where the row survives, -inf elsewhere
};

__global__ void __launch_bounds__(256) nms_prepare_segmented(const SegmentArgs a) {
  const int s = blockIdx.y;
  const Workspace ws = carve(a.workspace + (size_t)s * a.seg_bytes, a.rows);
  const DetSource<false> src{a.boxes + s * a.box_seg, a.scores + s * a.score_seg, a.box_row, a.score_row, a.score_thresh};
  prepare_body<true, true, false>(src, a.rows, ws.boxes, ws.areas, ws.order, a.kept + (long long)s * a.rows, blockIdx.x,
                           a.n_live + s);
}

// kMaskGrid workgroups per segment walk the segment's col_blocks x col_blocks tile pairs (the segment's live size is a
// device value: a grid of the worst case, (rows / 64)^2 pairs x 80 classes = 20 480 workgroups of which a detection's
// ~100 live ones did anything, cost 39 us; a class above the score threshold rarely has more than a few tile pairs)
constexpr int kMaskGrid = 16;
__global__ void __launch_bounds__(kTile) nms_mask_segmented(const SegmentArgs a, float thresh) {
  const int s = blockIdx.y;
  const int n = a.n_live[s];
  const int col_blocks = (n + kTile - 1) / kTile;
  const Workspace ws = carve(a.workspace + (size_t)s * a.seg_bytes, a.rows);
  for (int t = blockIdx.x; t < col_blocks * col_blocks; t += kMaskGrid) {
    const int row = t / col_blocks, col = t - row * col_blocks;
    mask_body<true>(ws.boxes, ws.areas, n, thresh, ws.mask, ws.diag_t, col, row, col_blocks);
    __syncthreads();  // the tile's LDS boxes are reused by the next pair
  }
}

__global__ void __launch_bounds__(256) nms_reduce_segmented(const SegmentArgs a) {
  const int s = blockIdx.x;
  const int n = a.n_live[s];
  int32_t* kept = a.kept + (long long)s * a.rows;
  if (n == 0) {
    if (threadIdx.x == 0) a.num_keep[s] = 0;
  } else {
    const Workspace ws = carve(a.workspace + (size_t)s * a.seg_bytes, a.rows);
    reduce_regs_body<true>(ws.mask, ws.diag_t, n, ws.order, kept, nullptr, nullptr, a.num_keep + s);
  }
  if (a.masked == nullptr) return;
  // the flags were written by wave 0 of this workgroup: workgroup-scope fence + barrier, then the whole workgroup writes
  // the segment's row of the masked score matrix (what the detections_per_im cut of core/test.py:776-785 ranks)
  __threadfence_block();
  __syncthreads();
  const float* score = a.scores + s * a.score_seg;
  float* out = a.masked + (long long)s * a.rows;
  for (int i = threadIdx.x; i < a.rows; i += 256)
    out[i] = kept[i] != 0 ? score[(long long)i * a.score_row] : -__builtin_inff();
}

// ---- the detections_per_im cut and the final gather (core/test.py:776-790) in one workgroup.  Input: the masked score
// matrix of mi_nms_segmented and its `cap` best entries in descending order (mi_topk_batched).  image_thresh = the
// D-th best; every surviving row at or above it stays (ties included: `total` counts them all, `count` = how many of
// them fit into the cap rows).  Output rows in the reference's order: class-major, RoI-ascending inside a class. ----
constexpr int kSelectThreads = 1024;
__global__ void __launch_bounds__(kSelectThreads)
detection_select(const float* __restrict__ scores, const float* __restrict__ boxes, const float* __restrict__ masked,
                 const float* __restrict__ top_vals, const long long* __restrict__ top_idx, int rows, int classes,
                 int cap, int detections_per_im, float* __restrict__ dets, int32_t* __restrict__ cls,
                 long long* __restrict__ sizes) {
  __shared__ long long s_flat[kSelectThreads];
  __shared__ int s_red[kSelectThreads / 64];
  __shared__ int s_count;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nseg = classes - 1;
  const long long m = (long long)nseg * rows;
  const float ninf = -__builtin_inff();
  const float thresh = (detections_per_im >= 1 && detections_per_im <= cap) ? top_vals[detections_per_im - 1] : ninf;
  for (int j = tid; j < nseg; j += kSelectThreads) sizes[2 + j] = 0;
  if (tid == 0) s_count = 0;
  int mine = 0;
  for (long long i = tid; i < m; i += kSelectThreads) {
    const float v = masked[i];
    mine += (v >= thresh) && (v > ninf);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
  if (lane == 0) s_red[wave] = mine;
  const bool sel = tid < cap && top_vals[tid] >= thresh && top_vals[tid] > ninf;
  const long long flat = sel ? top_idx[tid] : m + tid;   // unselected rows sort behind every real one, in input order
  s_flat[tid] = tid < cap ? flat : m + tid;
  __threadfence();                                       // the zeroed class counters, before the atomics below
  __syncthreads();
  if (tid == 0) {
    int total = 0;
    for (int w = 0; w < kSelectThreads / 64; w++) total += s_red[w];
    sizes[1] = total;
  }
  if (tid < cap) {
    int rank = 0;
    for (int j = 0; j < cap; j++) rank += s_flat[j] < flat;
    float* o = dets + (long long)rank * 5;
    if (sel) {
      const int c0 = (int)(flat / rows), roi = (int)(flat - (long long)c0 * rows);
      const float* b = boxes + ((long long)roi * classes + c0 + 1) * 4;
      o[0] = b[0];
      o[1] = b[1];
      o[2] = b[2];
      o[3] = b[3];
      o[4] = scores[(long long)roi * classes + c0 + 1];
      cls[rank] = c0 + 1;
      atomicAdd(reinterpret_cast<unsigned long long*>(&sizes[2 + c0]), 1ULL);
      atomicAdd(&s_count, 1);
    } else {
      o[0] = o[1] = o[2] = o[3] = o[4] = 0.f;
      cls[rank] = 0;
    }
  }
  __syncthreads();
  if (tid == 0) sizes[0] = s_count;
}

// ---- 4. flags -> ascending original indices (n > 4096 path) -------------------------------------------------
__global__ void __launch_bounds__(kCompactThreads)
nms_compact(const int32_t* __restrict__ flags, int n, int64_t* __restrict__ keep64,
            int32_t* __restrict__ num_keep) {
  __shared__ int s_wave[kCompactThreads / kTile];
  const int tid = threadIdx.x;
  const int per = (n + kCompactThreads - 1) / kCompactThreads;
  const int begin = min(tid * per, n), end = min(begin + per, n);
  int local = 0;
  for (int i = begin; i < end; i++) local += flags[i] != 0;
  // inclusive scan inside the wavefront
  int incl = local;
#pragma unroll
  for (int d = 1; d < kTile; d <<= 1) {
    const int up = __shfl_up(incl, d, kTile);
    if ((tid & (kTile - 1)) >= d) incl += up;
  }
  if ((tid & (kTile - 1)) == kTile - 1) s_wave[tid / kTile] = incl;
  __syncthreads();
  int wave_base = 0, total = 0;
  for (int w = 0; w < kCompactThreads / kTile; w++) {
    const int v = s_wave[w];
    if (w < tid / kTile) wave_base += v;
    total += v;
  }
  int pos = wave_base + incl - local;
  for (int i = begin; i < end; i++)
    if (flags[i] != 0) keep64[pos++] = i;
  if (tid == 0) *num_keep = total;
}

__global__ void nms_write_zero(int32_t* num_keep) { *num_keep = 0; }

// ---- IoU matrix -------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
bbox_overlaps_kernel(const float* __restrict__ boxes, int N, const float* __restrict__ query, int K,
                     float* __restrict__ overlaps) {
  const long long total = (long long)N * K;
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
       index += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(index % K);
    const int n = (int)(index / K);
    const float* q = query + (long long)k * 4;
    const float* b = boxes + (long long)n * 4;
    // cython_bbox.pyx:52-72 with the fp64 intermediates of the Cython-generated C (oracle.c)
    const float box_area = (float)(((double)(q[2] - q[0]) + 1.0) * ((double)(q[3] - q[1]) + 1.0));
    float result = 0.f;
    const float iw =
        (float)((double)((b[2] <= q[2] ? b[2] : q[2]) - (b[0] >= q[0] ? b[0] : q[0])) + 1.0);
    if (iw > 0) {
      const float ih =
          (float)((double)((b[3] <= q[3] ? b[3] : q[3]) - (b[1] >= q[1] ? b[1] : q[1])) + 1.0);
      if (ih > 0) {
        const float ua =
            (float)(((((double)(b[2] - b[0]) + 1.0) * ((double)(b[3] - b[1]) + 1.0)) + (double)box_area) -
                    (double)(iw * ih));
        result = iw * ih / ua;
      }
    }
    overlaps[index] = result;
  }
}

template <bool kGE>
int launch_reduce(int words, const Workspace& ws, int n, int32_t* keep32, int64_t* keep64, int32_t* num_keep,
                  hipStream_t s) {
  switch (words) {
    case 1: {
      nms_reduce_regs<kGE><<<1, 256, 0, s>>>(ws.mask, ws.diag_t, n, ws.order, ws.flags, keep32, keep64, num_keep);
      break;
    }
    case 2:
      nms_reduce<2, kGE><<<1, kTile, 0, s>>>(ws.mask, n, ws.order, ws.flags, keep32, num_keep);
      break;
    default:
      nms_reduce<kMaxWordsPerLane, kGE><<<1, kTile, 0, s>>>(ws.mask, n, ws.order, ws.flags, keep32,
                                                            num_keep);
      break;
  }
  return mi::check_launch("nms_reduce");
}

}  // namespace

extern "C" size_t mi_nms_workspace_bytes(int n) {
  if (n <= 0) return 16;
  return carve(nullptr, n).bytes;
}

extern "C" int mi_nms(const float* dets, int n, float thresh, int mode, void* keep,
                      int32_t* num_keep, void* workspace, size_t workspace_bytes,
                      mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(n >= 0, "nms: negative box count");
  MI_REQUIRE(mode == MI_NMS_GE_ORIG_ASC || mode == MI_NMS_GT_SORTED_POS, "nms: unknown mode %d", mode);
  MI_REQUIRE(num_keep != nullptr, "nms: null num_keep");
  hipStream_t s = mi::as_stream(stream);
  if (n == 0) {
    nms_write_zero<<<1, 1, 0, s>>>(num_keep);
    return mi::check_launch("nms_write_zero");
  }
  MI_REQUIRE(dets != nullptr && keep != nullptr && workspace != nullptr, "nms: null pointer");
  if (n > kMaxBoxes) {
    mi::set_error("nms: n = %d exceeds the %d boxes the single-wavefront reduce supports", n, kMaxBoxes);
    return MI_ERR_UNSUPPORTED;
  }
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "nms: workspace must be 16-byte aligned");
  Workspace ws = carve(workspace, n);
  if (workspace_bytes < ws.bytes) {
    mi::set_error("nms: workspace %zu bytes < required %zu", workspace_bytes, ws.bytes);
    return MI_ERR_WORKSPACE;
  }
  const int col_blocks = (n + kTile - 1) / kTile;
  const int words = (col_blocks + kTile - 1) / kTile;
  const bool ge = mode == MI_NMS_GE_ORIG_ASC;
  int rc;
  if (ge)
    nms_prepare<true><<<(n + kTile - 1) / kTile, 256, 0, s>>>(dets, n, ws.boxes, ws.areas, ws.order, ws.flags);
  else
    nms_prepare<false><<<(n + kTile - 1) / kTile, 256, 0, s>>>(dets, n, ws.boxes, ws.areas, ws.order, ws.flags);
  if ((rc = mi::check_launch("nms_prepare")) != MI_OK) return rc;
  dim3 grid(col_blocks, col_blocks);
  if (ge)
    nms_mask<true><<<grid, kTile, 0, s>>>(ws.boxes, ws.areas, n, thresh, ws.mask, ws.diag_t);
  else
    nms_mask<false><<<grid, kTile, 0, s>>>(ws.boxes, ws.areas, n, thresh, ws.mask, ws.diag_t);
  if ((rc = mi::check_launch("nms_mask")) != MI_OK) return rc;
  if (ge) {
    if ((rc = launch_reduce<true>(words, ws, n, nullptr, static_cast<int64_t*>(keep), num_keep, s)) != MI_OK) return rc;
    if (words == 1) return MI_OK;  // the four-wave reduce compacts in the same launch
    nms_compact<<<1, kCompactThreads, 0, s>>>(ws.flags, n, static_cast<int64_t*>(keep), num_keep);
    return mi::check_launch("nms_compact");
  }
  return launch_reduce<false>(words, ws, n, static_cast<int32_t*>(keep), nullptr, num_keep, s);
}

extern "C" size_t mi_nms_batched_workspace_bytes(int num_problems, const int* n) {
  size_t total = 0;
  for (int p = 0; p < num_problems; p++) total += (n[p] > 0 ? carve(nullptr, n[p]).bytes : 16);
  return total + 16;
}

extern "C" int mi_nms_batched(int num_problems, const float* const* dets, const int* n, float thresh, int mode,
                              void* const* keep, int32_t* const* num_keep, void* workspace,
                              size_t workspace_bytes, mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(num_problems >= 0, "nms_batched: negative problem count");
  MI_REQUIRE(mode == MI_NMS_GE_ORIG_ASC || mode == MI_NMS_GT_SORTED_POS, "nms_batched: unknown mode %d", mode);
  if (num_problems == 0) return MI_OK;
  MI_REQUIRE(dets != nullptr && n != nullptr && keep != nullptr && num_keep != nullptr && workspace != nullptr,
             "nms_batched: null pointer");
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "nms_batched: workspace must be 16-byte aligned");
  if (workspace_bytes < mi_nms_batched_workspace_bytes(num_problems, n)) {
    mi::set_error("nms_batched: workspace %zu bytes < required %zu", workspace_bytes,
                  mi_nms_batched_workspace_bytes(num_problems, n));
    return MI_ERR_WORKSPACE;
  }
  for (int p = 0; p < num_problems; p++) {
    MI_REQUIRE(n[p] >= 0, "nms_batched: negative box count");
    if (n[p] > kTile * kTile) {
      mi::set_error("nms_batched: problem %d has %d boxes; the batched path takes at most %d (use mi_nms)", p, n[p],
                    kTile * kTile);
      return MI_ERR_UNSUPPORTED;
    }
    MI_REQUIRE(num_keep[p] != nullptr && (n[p] == 0 || (dets[p] != nullptr && keep[p] != nullptr)),
               "nms_batched: null pointer in problem %d", p);
  }
  hipStream_t s = mi::as_stream(stream);
  const bool ge = mode == MI_NMS_GE_ORIG_ASC;
  char* base = static_cast<char*>(workspace);
  for (int first = 0; first < num_problems; first += kMaxBatch) {
    BatchTable t;
    t.count = num_problems - first < kMaxBatch ? num_problems - first : kMaxBatch;
    int max_cb = 0;
    t.chunk_start[0] = 0;
    for (int q = 0; q < t.count; q++) {
      const int p = first + q;
      t.n[q] = n[p];
      t.dets[q] = dets[p];
      t.keep[q] = keep[p];
      t.num_keep[q] = num_keep[p];
      const int cb = (n[p] + kTile - 1) / kTile;
      t.chunk_start[q + 1] = t.chunk_start[q] + cb;
      max_cb = cb > max_cb ? cb : max_cb;
      t.ws[q] = carve(base, n[p] > 0 ? n[p] : 1);
      base += n[p] > 0 ? t.ws[q].bytes : 16;
    }
    int rc;
    if (t.chunk_start[t.count] > 0) {
      if (ge)
        nms_prepare_batched<true><<<t.chunk_start[t.count], 256, 0, s>>>(t);
      else
        nms_prepare_batched<false><<<t.chunk_start[t.count], 256, 0, s>>>(t);
      if ((rc = mi::check_launch("nms_prepare_batched")) != MI_OK) return rc;
      dim3 grid(max_cb, max_cb, t.count);
      if (ge)
        nms_mask_batched<true><<<grid, kTile, 0, s>>>(t, thresh);
      else
        nms_mask_batched<false><<<grid, kTile, 0, s>>>(t, thresh);
      if ((rc = mi::check_launch("nms_mask_batched")) != MI_OK) return rc;
    }
    if (ge)
      nms_reduce_batched<true><<<t.count, 256, 0, s>>>(t);
    else
      nms_reduce_batched<false><<<t.count, 256, 0, s>>>(t);
    if ((rc = mi::check_launch("nms_reduce_batched")) != MI_OK) return rc;
  }
  return MI_OK;
}

extern "C" size_t mi_nms_segmented_workspace_bytes(int num_segments, int rows) {
  if (num_segments <= 0 || rows <= 0) return 16;
  return (size_t)num_segments * carve(nullptr, rows).bytes + align16((size_t)num_segments * sizeof(int32_t));
}

extern "C" int mi_nms_segmented(const float* boxes, long long box_segment_stride, long long box_row_stride,
                                const float* scores, long long score_segment_stride, long long score_row_stride,
                                int num_segments, int rows, float score_thresh, float nms_thresh, int32_t* kept,
                                int32_t* num_keep, float* masked_scores, void* workspace, size_t workspace_bytes,
                                mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(num_segments >= 0 && rows >= 0, "nms_segmented: negative size");
  if (num_segments == 0) return MI_OK;
  MI_REQUIRE(num_keep != nullptr, "nms_segmented: null num_keep");
  hipStream_t s = mi::as_stream(stream);
  if (rows == 0) {
    if (hipMemsetAsync(num_keep, 0, (size_t)num_segments * sizeof(int32_t), s) != hipSuccess)
      return mi::check_launch("nms_segmented: zero fill");
    return MI_OK;
  }
  MI_REQUIRE(boxes != nullptr && scores != nullptr && kept != nullptr && workspace != nullptr,
             "nms_segmented: null pointer");
  if (rows > kTile * kTile) {
    mi::set_error("nms_segmented: %d rows per segment; the four-wave reduce takes at most %d", rows, kTile * kTile);
    return MI_ERR_UNSUPPORTED;
  }
  MI_REQUIRE(num_segments <= 65535, "nms_segmented: at most 65535 segments");
  MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "nms_segmented: workspace must be 16-byte aligned");
  const size_t need = mi_nms_segmented_workspace_bytes(num_segments, rows);
  if (workspace_bytes < need) {
    mi::set_error("nms_segmented: workspace %zu bytes < required %zu", workspace_bytes, need);
    return MI_ERR_WORKSPACE;
  }
  SegmentArgs a;
  a.boxes = boxes;
  a.scores = scores;
  a.box_seg = box_segment_stride;
  a.box_row = box_row_stride;
  a.score_seg = score_segment_stride;
  a.score_row = score_row_stride;
  a.rows = rows;
  a.score_thresh = score_thresh;
  a.workspace = static_cast<char*>(workspace);
  a.seg_bytes = carve(nullptr, rows).bytes;
  a.n_live = reinterpret_cast<int32_t*>(a.workspace + (size_t)num_segments * a.seg_bytes);
  a.kept = kept;
  a.num_keep = num_keep;
  a.masked = masked_scores;
  const int cb = (rows + kTile - 1) / kTile;
  int rc;
  nms_prepare_segmented<<<dim3(cb, num_segments), 256, 0, s>>>(a);
  if ((rc = mi::check_launch("nms_prepare_segmented")) != MI_OK) return rc;
  nms_mask_segmented<<<dim3(std::min(cb * cb, kMaskGrid), num_segments), kTile, 0, s>>>(a, nms_thresh);
  if ((rc = mi::check_launch("nms_mask_segmented")) != MI_OK) return rc;
  nms_reduce_segmented<<<num_segments, 256, 0, s>>>(a);
  return mi::check_launch("nms_reduce_segmented");
}

extern "C" int mi_detection_select(const float* scores, const float* boxes, const float* masked_scores,
                                   const float* top_values, const int64_t* top_indices, int rows, int num_classes,
                                   int cap, int detections_per_im, float* dets, int32_t* cls, int64_t* sizes,
                                   mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(rows > 0 && num_classes >= 2 && cap > 0, "detection_select: bad size");
  if (cap > kSelectThreads) {
    mi::set_error("detection_select: cap = %d rows; one workgroup ranks at most %d", cap, kSelectThreads);
    return MI_ERR_UNSUPPORTED;
  }
  MI_REQUIRE((long long)(num_classes - 1) * rows >= cap, "detection_select: cap exceeds the number of candidates");
  MI_REQUIRE(scores != nullptr && boxes != nullptr && masked_scores != nullptr && top_values != nullptr &&
                 top_indices != nullptr && dets != nullptr && cls != nullptr && sizes != nullptr,
             "detection_select: null pointer");
  detection_select<<<1, kSelectThreads, 0, mi::as_stream(stream)>>>(
      scores, boxes, masked_scores, top_values, reinterpret_cast<const long long*>(top_indices), rows, num_classes, cap,
      detections_per_im, dets, cls, reinterpret_cast<long long*>(sizes));
  return mi::check_launch("detection_select");
}

extern "C" int mi_bbox_overlaps(const float* boxes, int num_boxes, const float* query, int num_query,
                                float* overlaps, mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(num_boxes >= 0 && num_query >= 0, "bbox_overlaps: negative size");
  const long long total = (long long)num_boxes * num_query;
  if (total == 0) return MI_OK;
  MI_REQUIRE(boxes != nullptr && query != nullptr && overlaps != nullptr, "bbox_overlaps: null pointer");
  const int block = 256;
  bbox_overlaps_kernel<<<mi::grid_for(total, block), block, 0, mi::as_stream(stream)>>>(
      boxes, num_boxes, query, num_query, overlaps);
  return mi::check_launch("bbox_overlaps");
}
