// topk.hip -- sorted top-k of fp32 arrays for gfx950, many independent problems per launch (C-ABI mi_topk_batched).
//
// The callers of the NMS kernels select before and after it: the k best RPN scores of every (level, image)
// (lib/modeling/generate_proposals.py:131-142, np.argpartition + argsort on the host in the reference), the
// post_nms_topN best of the collected levels (collect_and_distribute_fpn_rpn_proposals.py:83-98, np.argsort) and the
// detections_per_im best class scores of an image (core/test.py:776-785, np.sort).  torch.topk answers each of them with a
// dozen sort / merge launches (rocprofv3: ~320 us per test image for the six selections of the RPN path); here one
// workgroup per problem does the whole selection out of LDS:
//   1. radix select on the leading 24 bits of the k-th largest key, 12 + 12: two passes over the values (16 bytes per lane
//      and load; the array of the largest FPN level, 800 KB, stays in L2 after the first), LDS histogram, wave-aggregated
//      atomics for the hot bins (RPN scores of a fresh network all share their leading bits: plain LDS atomics would
//      serialise on one address);
//   2. one pass gathers every (key, index) pair at or above that 24-bit prefix into LDS (slots from a wave-aggregated
//      counter) -- normally a few more than k;
//   3. bitonic sort of the <= 4096 pairs in LDS: descending value, ties by ascending index; NaN sorts last; the first k
//      are the answer.
//   Only when more than 4096 values share the prefix (a flood of equal values: the -inf fill of a masked array) the last
//   8 bits are selected too, and if the k-th value itself is tied, two more passes pick the lowest indices among the ties,
//   so that the selected SET is unique and the result deterministic.
// Latency-bound single-CU work per problem; the problems of a call run side by side.
#include "common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxK = 4096;
constexpr int kMaxProblems = 64;
constexpr int kSplitAbove = 32768;   // a problem with more values than this is cut into chunks of about kChunk ...
constexpr int kChunk = 24576;        // ... one workgroup each, and a second launch merges the chunks' winners
constexpr int kMaxChunks = 16;
constexpr int kBins = 4096;

struct Problem {
  const float* values;
  float* out_values;
  long long* out_indices;
  const long long* remap;   // merge stage of a split problem: position -> original index (nullptr otherwise)
  long long index_base;     // first stage of a split problem: index of values[0] in the whole array
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

// histogram increment; when every active lane of the wavefront hits the same bin (a concentrated distribution: the
// hardware would serialise 64 atomics on one LDS address) the wavefront counts itself with one atomic
__device__ __forceinline__ void hist_add(uint32_t* hist, uint32_t bin, bool active, int lane) {
  const uint64_t act = __ballot(active);
  if (act == 0) return;
  const int leader = __builtin_ctzll(act);
  const uint32_t b0 = __builtin_amdgcn_readlane(bin, leader);
  const uint64_t same = __ballot(active && bin == b0);
  if (same == act) {
    if (lane == leader) atomicAdd(&hist[b0], (uint32_t)__popcll(act));
  } else if (active) {
    atomicAdd(&hist[bin], 1u);
  }
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

// Walk the values of one problem: every wavefront owns a contiguous range, a lane loads two float4 per trip (the pointer
// is aligned down to 16 bytes; the up to three values in front of the array and behind it are masked out, they share a
// 16-byte granule with real elements).  `body(i, v, live)` runs in wave-uniform control flow -- it may ballot.
template <typename Body>
__device__ __forceinline__ void for_each_value(const float* values, int n, Body body) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mis = (int)((reinterpret_cast<uintptr_t>(values) >> 2) & 3);
  const float4* base = reinterpret_cast<const float4*>(values - mis);
  const int total_vec = (n + mis + 3) >> 2;
  const int per_wave = ((total_vec + kWaves - 1) / kWaves + 127) & ~127;
  const int seg0 = min(wave * per_wave, total_vec), seg1 = min(seg0 + per_wave, total_vec);
  for (int vb = seg0; vb < seg1; vb += 128) {
    const int j0 = vb + lane, j1 = vb + 64 + lane;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (j0 < seg1) a = base[j0];
    if (j1 < seg1) b = base[j1];
    const int i0 = 4 * j0 - mis, i1 = 4 * j1 - mis;
    const bool in0 = j0 < seg1, in1 = j1 < seg1;
    body(i0 + 0, a.x, in0 && i0 + 0 >= 0 && i0 + 0 < n);
    body(i0 + 1, a.y, in0 && i0 + 1 >= 0 && i0 + 1 < n);
    body(i0 + 2, a.z, in0 && i0 + 2 >= 0 && i0 + 2 < n);
    body(i0 + 3, a.w, in0 && i0 + 3 >= 0 && i0 + 3 < n);
    body(i1 + 0, b.x, in1 && i1 + 0 >= 0 && i1 + 0 < n);
    body(i1 + 1, b.y, in1 && i1 + 1 >= 0 && i1 + 1 < n);
    body(i1 + 2, b.z, in1 && i1 + 2 >= 0 && i1 + 2 < n);
    body(i1 + 3, b.w, in1 && i1 + 3 >= 0 && i1 + 3 < n);
  }
}

__global__ void __launch_bounds__(kThreads) topk_select_sort(const Table t) {
  __shared__ uint32_t s_hist[kBins];
  __shared__ unsigned long long s_sel[kMaxK];
  __shared__ uint32_t s_wtot[kWaves];
  __shared__ Found s_found;
  __shared__ uint32_t s_count;
  const Problem p = t.p[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = p.n, k = p.k;
  if (k <= 0) return;

  // ---- 1. radix select on the value keys: 12 + 12 bits, and the last 8 only when they are needed -------------------
  uint32_t prefix = 0, mask = 0, kk = (uint32_t)k, ties = 0;
  uint32_t inv_floor = 0;       // ties are taken when (~index & 0xffffff) >= inv_floor
  auto digit_pass = [&](int shift, int bins) {
    for (int b = tid; b < bins; b += kThreads) s_hist[b] = 0;
    __syncthreads();
    const uint32_t pre = prefix, msk = mask;
    for_each_value(p.values, n, [&](int, float v, bool live) {
      const uint32_t key = key_of(v);
      hist_add(s_hist, (key >> shift) & (uint32_t)(bins - 1), live && (key & msk) == pre, lane);
    });
    __syncthreads();
    const Found f = find_bin(s_hist, bins, kk, s_wtot, &s_found);
    prefix |= f.bin << shift;
    mask |= (uint32_t)(bins - 1) << shift;
    kk -= f.above;
    ties = f.inside;
    __syncthreads();
  };
  digit_pass(20, 1 << 12);
  digit_pass(8, 1 << 12);
  // after 24 bits: (k - kk) values lie above the prefix, `ties` share it.  If all of them fit into the LDS sort, the
  // sort settles the rest (the common case); otherwise (a flood of equal values, e.g. the -inf fill of a masked array)
  // the last 8 bits are selected too, and then -- if the k-th value itself is tied -- the lowest indices among the ties.
  uint32_t floor_key = prefix;  // gather every value whose key is >= floor_key (and, for ties of it, passes inv_floor)
  bool exact = false;           // floor_key is the full key of the k-th value
  if ((uint32_t)k - kk + ties > (uint32_t)kMaxK) {
    digit_pass(0, 1 << 8);
    floor_key = prefix;
    exact = true;
    if (ties > kk) {
      uint32_t iprefix = 0, imask = 0;
#pragma unroll
      for (int pass = 0; pass < 2; pass++) {
        const int shift = pass == 0 ? 12 : 0;
        for (int b = tid; b < kBins; b += kThreads) s_hist[b] = 0;
        __syncthreads();
        const uint32_t kth = prefix;
        for_each_value(p.values, n, [&](int i, float v, bool live) {
          const uint32_t inv = ~(uint32_t)i & 0xffffffu;
          hist_add(s_hist, (inv >> shift) & 0xfffu, live && key_of(v) == kth && (inv & imask) == iprefix, lane);
        });
        __syncthreads();
        const Found f = find_bin(s_hist, kBins, kk, s_wtot, &s_found);
        iprefix |= f.bin << shift;
        imask |= 0xfffu << shift;
        kk -= f.above;
        __syncthreads();
      }
      inv_floor = iprefix;
    }
  }
  const uint32_t cmp_mask = exact ? 0xffffffffu : 0xffffff00u;
  const uint32_t gathered = exact ? (uint32_t)k : (uint32_t)k - kk + ties;   // how many pairs the gather finds
  // ---- 2. gather the selected pairs into LDS ----------------------------------------------------------------------
  int m = 1;
  while (m < (int)gathered) m <<= 1;
  for (int i = tid; i < m; i += kThreads) s_sel[i] = 0ULL;  // padding sorts last
  if (tid == 0) s_count = 0;
  __syncthreads();
  for_each_value(p.values, n, [&](int i, float v, bool live) {
    const uint32_t key = key_of(v);
    const uint32_t inv = ~(uint32_t)i;
    const uint32_t hi = key & cmp_mask;
    const bool take = live && (hi > floor_key || (hi == floor_key && (!exact || (inv & 0xffffffu) >= inv_floor)));
    const uint64_t mm = __ballot(take);
    if (mm != 0) {
      const int leader = __builtin_ctzll(mm);
      uint32_t slot0 = 0;
      if (lane == leader) slot0 = atomicAdd(&s_count, (uint32_t)__popcll(mm));
      slot0 = __shfl(slot0, leader, 64);
      if (take) {
        const uint32_t slot = slot0 + (uint32_t)__popcll(mm & ((1ULL << lane) - 1ULL));
        if (slot < gathered) s_sel[slot] = ((unsigned long long)key << 32) | inv;
      }
    }
  });
  // ---- 3. bitonic sort, descending (value descending, index ascending) ----------------------------------------------
  // Sub-stages whose partner distance is below 128 stay inside a 128-element block; a wavefront owns whole blocks and
  // runs those sub-stages back to back without workgroup barriers (LDS operations of one wavefront complete in order).
  // Only the sub-stages with a distance of 128 and more -- 10 of the 66 of a 2048-element sort -- need the workgroup.
  const int wave = tid >> 6;
  const int block = m < 128 ? m : 128, half = block >> 1;
  auto exchange = [&](int q, int stride, int size) {
    const int i = ((q & ~(stride - 1)) << 1) | (q & (stride - 1)), j = i + stride;
    const bool desc = (i & size) == 0;
    const unsigned long long x = s_sel[i], y = s_sel[j];
    if ((x < y) == desc) {
      s_sel[i] = y;
      s_sel[j] = x;
    }
  };
  __syncthreads();
  for (int size = 2; size <= m; size <<= 1) {
    int stride = size >> 1;
    for (; stride >= 128; stride >>= 1) {
      for (int q = tid; q < (m >> 1); q += kThreads) exchange(q, stride, size);
      __syncthreads();
    }
    for (int blk = wave; blk * block < m; blk += kWaves) {
      const int q0 = blk * half;   // pairs [q0, q0 + half) live in elements [blk * block, (blk + 1) * block)
      for (int st = stride; st > 0; st >>= 1) {
        if (lane < half) exchange(q0 + lane, st, size);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
  }
  for (int r = tid; r < k; r += kThreads) {
    const uint32_t idx = ~(uint32_t)(s_sel[r] & 0xffffffffULL);
    p.out_indices[r] = p.remap != nullptr ? p.remap[idx] : (long long)idx + p.index_base;
    p.out_values[r] = p.values[idx];
  }
}

}  // namespace

namespace {

// top-k is decomposable: top-k(whole) = top-k(union of the chunks' top-k).  One workgroup walks ~24k values in the time
// the launch overhead costs anyway; the largest FPN level (201 600 scores) becomes 9 chunks side by side and a merge of
// 9 k candidates.  Chunks are in index order and each chunk's winners are in (value desc, index asc) order, so "lower
// position first" among the merge stage's ties is "lower index first" of the whole array.
struct Split {
  int chunks, chunk_n;
};
Split plan(int n, int k) {
  Split s{1, n};
  if (n <= kSplitAbove || k == 0) return s;
  int chunks = (n + kChunk - 1) / kChunk;
  if (chunks > kMaxChunks) chunks = kMaxChunks;
  while (chunks > 1 && (long long)chunks * k > (1 << 24)) chunks--;
  if (chunks < 2) return s;
  s.chunks = chunks;
  s.chunk_n = (((n + chunks - 1) / chunks) + 63) & ~63;
  s.chunks = (n + s.chunk_n - 1) / s.chunk_n;
  return s;
}
int chunk_k(const Split& sp, int n, int k, int c) {
  const int len = (c + 1) * sp.chunk_n <= n ? sp.chunk_n : n - c * sp.chunk_n;
  return k < len ? k : len;
}
size_t align16(size_t b) { return (b + 15) & ~size_t(15); }

int launch_table(const Table& t, hipStream_t s) {
  topk_select_sort<<<t.count, kThreads, 0, s>>>(t);
  return mi::check_launch("topk_select_sort");
}

}  // namespace

extern "C" size_t mi_topk_batched_workspace_bytes(int num_problems, const int* n, const int* k) {
  size_t total = 16;
  if (n == nullptr || k == nullptr) return total;
  for (int q = 0; q < num_problems; q++) {
    const Split sp = plan(n[q], k[q]);
    if (sp.chunks < 2) continue;
    size_t cand = 0;
    for (int c = 0; c < sp.chunks; c++) cand += (size_t)chunk_k(sp, n[q], k[q], c);
    total += align16(cand * sizeof(float)) + align16(cand * sizeof(long long));
  }
  return total;
}

extern "C" int mi_topk_batched(int num_problems, const float* const* values, const int* n, const int* k,
                               float* const* out_values, int64_t* const* out_indices, void* workspace,
                               size_t workspace_bytes, mi_stream_t stream) {
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
  const size_t need = mi_topk_batched_workspace_bytes(num_problems, n, k);
  if (need > 16) {
    MI_REQUIRE(workspace != nullptr && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
               "topk_batched: a 16-byte aligned workspace is needed for problems of more than %d values", kSplitAbove);
    if (workspace_bytes < need) {
      mi::set_error("topk_batched: workspace %zu bytes < required %zu", workspace_bytes, need);
      return MI_ERR_WORKSPACE;
    }
  }
  hipStream_t s = mi::as_stream(stream);
  char* ws = static_cast<char*>(workspace);
  Table first, merge;
  first.count = merge.count = 0;
  int rc;
  auto push = [&](Table& t, const Problem& p) -> int {
    t.p[t.count++] = p;
    if (t.count == kMaxProblems) {
      const int r = launch_table(t, s);
      t.count = 0;
      return r;
    }
    return MI_OK;
  };
  // first-stage tables are launched as they fill up, the merge problems after all of them (stream order)
  for (int q = 0; q < num_problems; q++) {
    if (k[q] == 0) continue;
    const Split sp = plan(n[q], k[q]);
    long long* out_idx = reinterpret_cast<long long*>(out_indices[q]);
    if (sp.chunks < 2) {
      if ((rc = push(first, Problem{values[q], out_values[q], out_idx, nullptr, 0, n[q], k[q]})) != MI_OK) return rc;
      continue;
    }
    size_t cand = 0;
    for (int c = 0; c < sp.chunks; c++) cand += (size_t)chunk_k(sp, n[q], k[q], c);
    float* cand_vals = reinterpret_cast<float*>(ws);
    ws += align16(cand * sizeof(float));
    long long* cand_idx = reinterpret_cast<long long*>(ws);
    ws += align16(cand * sizeof(long long));
    size_t off = 0;
    for (int c = 0; c < sp.chunks; c++) {
      const int kc = chunk_k(sp, n[q], k[q], c);
      const int len = (c + 1) * sp.chunk_n <= n[q] ? sp.chunk_n : n[q] - c * sp.chunk_n;
      if ((rc = push(first, Problem{values[q] + (size_t)c * sp.chunk_n, cand_vals + off, cand_idx + off, nullptr,
                                    (long long)c * sp.chunk_n, len, kc})) != MI_OK)
        return rc;
      off += kc;
    }
    // merges are collected and launched after every first-stage table of this call has been issued
    if (merge.count == kMaxProblems) {
      if (first.count > 0) {
        if ((rc = launch_table(first, s)) != MI_OK) return rc;
        first.count = 0;
      }
      if ((rc = launch_table(merge, s)) != MI_OK) return rc;
      merge.count = 0;
    }
    merge.p[merge.count++] = Problem{cand_vals, out_values[q], out_idx, cand_idx, 0, (int)cand, k[q]};
  }
  if (first.count > 0 && (rc = launch_table(first, s)) != MI_OK) return rc;
  if (merge.count > 0 && (rc = launch_table(merge, s)) != MI_OK) return rc;
  return MI_OK;
}
