// topk.hip -- sorted top-k of fp32 arrays for gfx950, many independent problems per launch (C-ABI mi_topk_batched).
//
// The callers of the NMS kernels select before and after it: the k best RPN scores of every (level, image)
// (lib/modeling/generate_proposals.py:131-142, np.argpartition + argsort on the host in the reference), the
// post_nms_topN best of the collected levels (collect_and_distribute_fpn_rpn_proposals.py:83-98, np.argsort) and the
// detections_per_im best class scores of an image (core/test.py:776-785, np.sort).  torch.topk answers each of them with a
// dozen sort / merge launches (rocprofv3: ~320 us per test image for the six selections of the RPN path); here one
// workgroup per problem does the whole selection out of LDS:
//   1. radix select of the k-th largest key, 12 + 12 + 8 bits: three passes over the values (the array of the largest
//      FPN level, 800 KB, stays in L2 after the first), LDS histogram, wave-aggregated atomics for the hot bins (RPN scores
//      of a fresh network all share their leading bits: plain LDS atomics would serialise on one address);
//   2. if the k-th key is tied with more elements than fit, two more passes select among the ties by index (lowest
//      first), so the selected SET is unique and the result deterministic;
//   3. one pass gathers the selected (key, index) pairs into LDS (slots from a wave-aggregated counter);
//   4. bitonic sort of the <= 4096 pairs in LDS: descending value, ties by ascending index; NaN sorts last.
// Latency-bound single-CU work per problem; the problems of a call run side by side.
#include "common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxK = 4096;
constexpr int kMaxProblems = 16;
constexpr int kBins = 4096;

struct Problem {
  const float* values;
  float* out_values;
  long long* out_indices;
  int n, k;
};
struct Table {
  int count;
  Problem p[kMaxProblems];
};

// monotone map float -> uint32 (larger float = larger key); NaN -> 0, below -inf
__device__ __forceinline__ uint32_t key_of(float v) {
  const uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0u;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// histogram increment; the first two distinct bins of the wavefront are counted with one atomic each (a concentrated
// distribution puts all 64 lanes on one LDS address, which the hardware serialises)
__device__ __forceinline__ void hist_add(uint32_t* hist, uint32_t bin, bool active, int lane) {
  uint64_t todo = __ballot(active);
#pragma unroll
  for (int it = 0; it < 2; it++) {
    if (todo == 0) break;
    const int leader = __builtin_ctzll(todo);
    const uint32_t b0 = __builtin_amdgcn_readlane(bin, leader);
    const bool same = active && bin == b0;
    const uint64_t m = __ballot(same);
    if (lane == leader) atomicAdd(&hist[b0], (uint32_t)__popcll(m));
    if (same) active = false;
    todo &= ~m;
  }
  if (active) atomicAdd(&hist[bin], 1u);
}

struct Found {
  uint32_t bin;     // the bin holding the kk-th largest element
  uint32_t above;   // elements in higher bins
  uint32_t inside;  // elements in that bin
};

// Which of `bins` (<= 4096, a multiple of 64 or less than 64 * per) histogram bins holds the kk-th largest element?
// Block-wide; s_wtot / s_found are LDS scratch.  All threads return the same answer.
__device__ __forceinline__ Found find_bin(const uint32_t* hist, int bins, uint32_t kk, uint32_t* s_wtot, Found* s_found) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = bins >= kThreads ? bins / kThreads : 1;
  const int first = tid * per;
  uint32_t c = 0;
  if (first < bins)
    for (int b = 0; b < per; b++) c += hist[first + b];
  uint32_t incl = c;  // inclusive suffix sum over the lanes of the wave (lane 63 = highest bins)
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_down(incl, d, 64);
    if (lane + d < 64) incl += y;
  }
  if (lane == 0) s_wtot[wave] = incl;
  __syncthreads();
  uint32_t above_w = 0;
  for (int w = wave + 1; w < kWaves; w++) above_w += s_wtot[w];
  const uint32_t hi = above_w + incl - c;  // elements in bins above this thread's
  if (hi < kk && kk <= hi + c) {           // exactly one thread
    uint32_t above = hi;
    for (int b = per - 1; b >= 0; b--) {
      const uint32_t h = hist[first + b];
      if (kk <= above + h) {
        s_found->bin = (uint32_t)(first + b);
        s_found->above = above;
        s_found->inside = h;
        break;
      }
      above += h;
    }
  }
  __syncthreads();
  return *s_found;
}

__global__ void __launch_bounds__(kThreads) topk_select_sort(const Table t) {
  __shared__ uint32_t s_hist[kBins];
  __shared__ unsigned long long s_sel[kMaxK];
  __shared__ uint32_t s_wtot[kWaves];
  __shared__ Found s_found;
  __shared__ uint32_t s_count;
  const Problem p = t.p[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = p.n, k = p.k;
  if (k <= 0) return;
  // every wavefront walks a contiguous segment, 64 consecutive values per load instruction
  const int per_wave = ((n + kWaves - 1) / kWaves + 63) & ~63;
  const int seg0 = min(wave * per_wave, n), seg1 = min(seg0 + per_wave, n);

  // ---- 1. radix select on the value keys --------------------------------------------------------------------------
  uint32_t prefix = 0, mask = 0, kk = (uint32_t)k, ties = 0;
  const int shifts[3] = {20, 8, 0};
  const int widths[3] = {12, 12, 8};
#pragma unroll
  for (int pass = 0; pass < 3; pass++) {
    const int shift = shifts[pass], bins = 1 << widths[pass];
    for (int b = tid; b < bins; b += kThreads) s_hist[b] = 0;
    __syncthreads();
    for (int base = seg0; base < seg1; base += 256) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = base + u * 64 + lane;
        v[u] = i < seg1 ? p.values[i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = base + u * 64 + lane;
        const uint32_t key = key_of(v[u]);
        hist_add(s_hist, (key >> shift) & (uint32_t)(bins - 1), i < seg1 && (key & mask) == prefix, lane);
      }
    }
    __syncthreads();
    const Found f = find_bin(s_hist, bins, kk, s_wtot, &s_found);
    prefix |= f.bin << shift;
    mask |= (uint32_t)(bins - 1) << shift;
    kk -= f.above;
    ties = f.inside;
    __syncthreads();
  }
  const uint32_t kth = prefix;  // key of the k-th largest value; kk of its `ties` occurrences are wanted
  // ---- 2. more ties than wanted: the kk lowest indices among them (select on the inverted index, 12 + 12 bits) ----
  uint32_t inv_floor = 0;       // take a tie when (~index & 0xffffff) >= inv_floor
  if (ties > kk) {
    uint32_t iprefix = 0, imask = 0;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int shift = pass == 0 ? 12 : 0;
      for (int b = tid; b < kBins; b += kThreads) s_hist[b] = 0;
      __syncthreads();
      for (int base = seg0; base < seg1; base += 256) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = base + u * 64 + lane;
          v[u] = i < seg1 ? p.values[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = base + u * 64 + lane;
          const uint32_t inv = ~(uint32_t)i & 0xffffffu;
          hist_add(s_hist, (inv >> shift) & 0xfffu, i < seg1 && key_of(v[u]) == kth && (inv & imask) == iprefix, lane);
        }
      }
      __syncthreads();
      const Found f = find_bin(s_hist, kBins, kk, s_wtot, &s_found);
      iprefix |= f.bin << shift;
      imask |= 0xfffu << shift;
      kk -= f.above;
      __syncthreads();
    }
    inv_floor = iprefix;
  }
  // ---- 3. gather the selected pairs into LDS ----------------------------------------------------------------------
  int m = 1;
  while (m < k) m <<= 1;
  for (int i = tid; i < m; i += kThreads) s_sel[i] = 0ULL;  // padding sorts last
  if (tid == 0) s_count = 0;
  __syncthreads();
  for (int base = seg0; base < seg1; base += 256) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = base + u * 64 + lane;
      v[u] = i < seg1 ? p.values[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = base + u * 64 + lane;
      const uint32_t key = key_of(v[u]);
      const uint32_t inv = ~(uint32_t)i;
      const bool take = i < seg1 && (key > kth || (key == kth && (inv & 0xffffffu) >= inv_floor));
      const uint64_t mm = __ballot(take);
      if (mm != 0) {
        const int leader = __builtin_ctzll(mm);
        uint32_t slot0 = 0;
        if (lane == leader) slot0 = atomicAdd(&s_count, (uint32_t)__popcll(mm));
        slot0 = __shfl(slot0, leader, 64);
        if (take) {
          const uint32_t slot = slot0 + (uint32_t)__popcll(mm & ((1ULL << lane) - 1ULL));
          if (slot < (uint32_t)k) s_sel[slot] = ((unsigned long long)key << 32) | inv;
        }
      }
    }
  }
  // ---- 4. bitonic sort, descending (value descending, index ascending) ----------------------------------------------
  for (int size = 2; size <= m; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int q = tid; q < (m >> 1); q += kThreads) {
        const int i = ((q / stride) * stride << 1) + (q % stride), j = i + stride;
        const bool desc = (i & size) == 0;
        const unsigned long long a = s_sel[i], b = s_sel[j];
        if ((a < b) == desc) {
          s_sel[i] = b;
          s_sel[j] = a;
        }
      }
    }
  __syncthreads();
  for (int r = tid; r < k; r += kThreads) {
    const uint32_t idx = ~(uint32_t)(s_sel[r] & 0xffffffffULL);
    p.out_indices[r] = (long long)idx;
    p.out_values[r] = p.values[idx];
  }
}

}  // namespace

extern "C" int mi_topk_batched(int num_problems, const float* const* values, const int* n, const int* k,
                               float* const* out_values, int64_t* const* out_indices, mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(num_problems >= 0, "topk_batched: negative problem count");
  if (num_problems == 0) return MI_OK;
  MI_REQUIRE(values != nullptr && n != nullptr && k != nullptr && out_values != nullptr && out_indices != nullptr,
             "topk_batched: null pointer");
  for (int q = 0; q < num_problems; q++) {
    MI_REQUIRE(n[q] >= 0 && k[q] >= 0 && k[q] <= n[q], "topk_batched: problem %d: need 0 <= k <= n", q);
    MI_REQUIRE(n[q] <= (1 << 24), "topk_batched: problem %d has %d values, at most %d are supported", q, n[q], 1 << 24);
    if (k[q] > kMaxK) {
      mi::set_error("topk_batched: problem %d asks for k = %d; the LDS sort holds at most %d", q, k[q], kMaxK);
      return MI_ERR_UNSUPPORTED;
    }
    MI_REQUIRE(k[q] == 0 || (values[q] != nullptr && out_values[q] != nullptr && out_indices[q] != nullptr),
               "topk_batched: null pointer in problem %d", q);
  }
  hipStream_t s = mi::as_stream(stream);
  for (int first = 0; first < num_problems; first += kMaxProblems) {
    Table t;
    t.count = num_problems - first < kMaxProblems ? num_problems - first : kMaxProblems;
    for (int q = 0; q < t.count; q++) {
      t.p[q].values = values[first + q];
      t.p[q].out_values = out_values[first + q];
      t.p[q].out_indices = reinterpret_cast<long long*>(out_indices[first + q]);
      t.p[q].n = n[first + q];
      t.p[q].k = k[first + q];
    }
    topk_select_sort<<<t.count, kThreads, 0, s>>>(t);
    int rc = mi::check_launch("topk_select_sort");
    if (rc != MI_OK) return rc;
  }
  return MI_OK;
}
