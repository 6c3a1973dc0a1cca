// soft_nms.hip -- Soft-NMS (hard / linear / gaussian re-scoring), C-ABI mi_soft_nms, for gfx950.
//
// Follows utils.cython_nms.soft_nms (lib/utils/cython_nms.pyx:98-203) in result AND in layout of the result: the
// reference works in place on one array -- pick the best remaining box, swap it to the front, re-score everything
// behind it, and drop a box whose score falls below `threshold` by overwriting it with the current last box (which is
// then re-scored in its new place).  The order of the surviving rows therefore depends on that swap discipline, and
// the drop-in returns exactly the rows boxes[:N], inds[:N] the reference returns.
//
// Arithmetic: as the C that Cython generates (cython_nms.c:3882-3967 of the reference): variables are fp32, but the
// literal 1 inside the float expressions is the double constant 1.0, so  area, iw, ih, ua  and  1 - ov  are evaluated
// in double and rounded to fp32 on assignment;  ov = (iw * ih) / ua  in fp32;  the gaussian weight is exp in double of
// the fp32 value -(ov * ov) / sigma.  The library is compiled with -ffp-contract=off.
//
// One workgroup per problem, all state in LDS (28 B per box; at most 4096 boxes).  The outer loop over picked boxes is
// inherently sequential; each step is parallel over the remaining boxes:
//   1. arg-max of the scores in [i, N), first maximum wins (the reference scans with a strict '<', :131-135)
//   2. swap rows i <-> maxpos (:138-151)
//   3. re-score rows (i, N) against row i (:162-190); a row fails when its new score < threshold (:194)
//   4. if any row failed: the reference's "overwrite with the last row, shrink, re-examine" loop is a two-pointer
//      compaction -- with N' = N - #failed, the k-th failed row below N' (ascending) receives the k-th surviving row
//      at or above N' (descending).  Both ranks come from 64-bit ballot masks and one wave-wide prefix sum.
#include "common.h"

namespace {

using namespace mi;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxBoxes = 4096;
constexpr int kMaxMasks = kMaxBoxes / 64;  // 64: the per-chunk counts fit one wavefront

__device__ __forceinline__ float f32min(float a, float b) { return a <= b ? a : b; }  // cython_nms.pyx:31-32
__device__ __forceinline__ float f32max(float a, float b) { return a >= b ? a : b; }  // :28-29

struct Lds {
  float *x1, *y1, *x2, *y2, *sc;
  int* ind;
  int* src;                    // [n] rows that move into the holes, by hole rank
  unsigned long long* fmask;   // [kMaxMasks] failed-row masks of the current step (relative to row i + 1)
  int* pre_hole;               // [kMaxMasks + 1] exclusive prefix of holes per chunk, total at [kMaxMasks]
  int* pre_tail;               // [kMaxMasks + 1] same for the surviving rows of the tail
  float* red_s;                // [kWaves] arg-max partials
  int* red_p;                  // [kWaves]
  int* wfail;                  // [kWaves] failed rows per wave
  __device__ Lds(unsigned char* base, int n) {
    fmask = reinterpret_cast<unsigned long long*>(base);
    float* f = reinterpret_cast<float*>(fmask + kMaxMasks);
    x1 = f;
    y1 = x1 + n;
    x2 = y1 + n;
    y2 = x2 + n;
    sc = y2 + n;
    ind = reinterpret_cast<int*>(sc + n);
    src = ind + n;
    pre_hole = src + n;
    pre_tail = pre_hole + kMaxMasks + 1;
    red_s = reinterpret_cast<float*>(pre_tail + kMaxMasks + 1);
    red_p = reinterpret_cast<int*>(red_s + kWaves);
    wfail = red_p + kWaves;
  }
  static size_t bytes(int n) {
    return sizeof(unsigned long long) * kMaxMasks + sizeof(float) * 7 * (size_t)n + sizeof(int) * (2 * (kMaxMasks + 1) + 3 * kWaves);
  }
};

// bits of chunk m (rows 64 m .. 64 m + 63, relative) that lie below the relative row `limit`
__device__ __forceinline__ unsigned long long below(int limit, int m) {
  const int k = limit - 64 * m;
  return k <= 0 ? 0ull : (k >= 64 ? ~0ull : ((1ull << k) - 1ull));
}

// One workgroup per problem.  offsets == nullptr: a single problem of `n` rows.  Otherwise problem p = blockIdx.x owns
// rows [offsets[p], offsets[p + 1]) of dets / out_dets / out_inds (indices are relative to the problem's first row) and
// `n` is only the capacity the LDS image was sized for.
__global__ void __launch_bounds__(kThreads)
soft_nms_kernel(const float* __restrict__ dets, int n, const int* __restrict__ offsets, float sigma, float Nt,
                float threshold, int method, float* __restrict__ out_dets, long long* __restrict__ out_inds,
                int* __restrict__ num_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int cap = n;
  if (offsets != nullptr) {
    const int first = offsets[blockIdx.x];
    n = offsets[blockIdx.x + 1] - first;
    dets += (long long)first * 5;
    out_dets += (long long)first * 5;
    out_inds += first;
    num_out += blockIdx.x;
    if (n > cap) n = cap;  // never reached through mi_soft_nms_segmented's contract (capacity >= longest segment)
  }
  Lds s(smem, cap);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int p = tid; p < n; p += kThreads) {  // boxes_in.copy(), inds = arange(N) (:108,:116)
    s.x1[p] = dets[p * 5 + 0];
    s.y1[p] = dets[p * 5 + 1];
    s.x2[p] = dets[p * 5 + 2];
    s.y2[p] = dets[p * 5 + 3];
    s.sc[p] = dets[p * 5 + 4];
    s.ind[p] = p;
  }
  __syncthreads();
  int N = n;
  for (int i = 0; i < N; i++) {
    // ---- 1. first maximum of sc[i .. N) ----
    float best = -INFINITY;
    int bpos = 0x7fffffff;
    for (int p = i + tid; p < N; p += kThreads) {
      const float v = s.sc[p];
      if (bpos == 0x7fffffff || best < v) {  // ascending scan inside the thread: strict '<' keeps the first maximum
        best = v;
        bpos = p;
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const float ov = __shfl_xor(best, d);
      const int op = __shfl_xor(bpos, d);
      const bool take = op != 0x7fffffff && (bpos == 0x7fffffff || best < ov || (best == ov && op < bpos));
      if (take) {
        best = ov;
        bpos = op;
      }
    }
    if (lane == 0) {
      s.red_s[wave] = best;
      s.red_p[wave] = bpos;
    }
    __syncthreads();
    best = s.red_s[0];
    bpos = s.red_p[0];
#pragma unroll
    for (int w = 1; w < kWaves; w++) {
      const float ov = s.red_s[w];
      const int op = s.red_p[w];
      if (op != 0x7fffffff && (bpos == 0x7fffffff || best < ov || (best == ov && op < bpos))) {
        best = ov;
        bpos = op;
      }
    }
    // A NaN score never wins a '<' (as in the reference); if everything compared false the scan kept the first row.
    const int maxpos = bpos;
    // ---- 2. swap rows i and maxpos ----
    if (tid == 0 && maxpos != i) {
      float t;
      int ti;
      t = s.x1[i]; s.x1[i] = s.x1[maxpos]; s.x1[maxpos] = t;
      t = s.y1[i]; s.y1[i] = s.y1[maxpos]; s.y1[maxpos] = t;
      t = s.x2[i]; s.x2[i] = s.x2[maxpos]; s.x2[maxpos] = t;
      t = s.y2[i]; s.y2[i] = s.y2[maxpos]; s.y2[maxpos] = t;
      t = s.sc[i]; s.sc[i] = s.sc[maxpos]; s.sc[maxpos] = t;
      ti = s.ind[i]; s.ind[i] = s.ind[maxpos]; s.ind[maxpos] = ti;
    }
    __syncthreads();
    // ---- 3. re-score rows (i, N) ----
    const float tx1 = s.x1[i], ty1 = s.y1[i], tx2 = s.x2[i], ty2 = s.y2[i];
    const int L = N - (i + 1);  // rows behind the picked one; relative row q <-> absolute row i + 1 + q
    int my_fails = 0;
    for (int q0 = 0; q0 < L; q0 += kThreads) {
      const int q = q0 + tid, p = i + 1 + q;
      bool fail = false;
      if (q < L) {
        const float x1 = s.x1[p], y1 = s.y1[p], x2 = s.x2[p], y2 = s.y2[p];
        const float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));  // :169
        const float iw = (float)((double)(f32min(tx2, x2) - f32max(tx1, x1)) + 1.0);        // :170
        if (iw > 0) {
          const float ih = (float)((double)(f32min(ty2, y2) - f32max(ty1, y1)) + 1.0);  // :172
          if (ih > 0) {
            const float ua = (float)(((((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0)) + (double)area) -
                                     (double)(iw * ih));  // :174
            const float ov = iw * ih / ua;                // :175
            float weight;
            if (method == 1)
              weight = ov > Nt ? (float)(1.0 - (double)ov) : 1.f;  // :177-181
            else if (method == 2)
              weight = (float)exp((double)(-(ov * ov) / sigma));  // :182-183
            else
              weight = ov > Nt ? 0.f : 1.f;  // :184-188
            const float ns = weight * s.sc[p];  // :190
            s.sc[p] = ns;
            fail = ns < threshold;  // :194
          }
        }
      }
      const unsigned long long m = __ballot(fail);
      if (lane == 0) s.fmask[(q0 >> 6) + wave] = m;
      my_fails += __popcll(m);
    }
    if (lane == 0) s.wfail[wave] = my_fails;
    __syncthreads();
    int F = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) F += s.wfail[w];
    if (F == 0) continue;  // uniform: every thread read the same counts; the next step's barriers order the reuse
    // ---- 4. two-pointer compaction of rows (i, N) ----
    const int Lk = L - F;                   // surviving rows; new N = i + 1 + Lk
    const int nmask = (L + kThreads - 1) / kThreads * kWaves;  // chunks written above (the last ones may be empty)
    if (wave == 0) {
      int holes = 0, tails = 0;
      if (lane < nmask) {
        const unsigned long long fm = s.fmask[lane];
        holes = __popcll(fm & below(Lk, lane));
        tails = __popcll(~fm & ~below(Lk, lane) & below(L, lane));
      }
      int ph = holes, pt = tails;  // inclusive prefix sums over the lanes
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int a = __shfl_up(ph, d), b = __shfl_up(pt, d);
        if (lane >= d) {
          ph += a;
          pt += b;
        }
      }
      s.pre_hole[lane] = ph - holes;
      s.pre_tail[lane] = pt - tails;
      if (lane == 63) {
        s.pre_hole[kMaxMasks] = ph;
        s.pre_tail[kMaxMasks] = pt;
      }
    }
    __syncthreads();
    const int H = s.pre_hole[kMaxMasks];  // == pre_tail total: as many holes below N' as survivors at or above it
    for (int q0 = 0; q0 < L; q0 += kThreads) {
      const int q = q0 + tid;
      if (q < L && q >= Lk) {
        const int m = q >> 6, bit = q & 63;
        const unsigned long long tm = ~s.fmask[m] & ~below(Lk, m) & below(L, m);
        if ((tm >> bit) & 1ull) {
          const int asc = s.pre_tail[m] + __popcll(tm & ((1ull << bit) - 1ull));
          s.src[H - 1 - asc] = i + 1 + q;  // the k-th hole takes the k-th survivor counted from the end
        }
      }
    }
    __syncthreads();
    for (int q0 = 0; q0 < Lk; q0 += kThreads) {
      const int q = q0 + tid;
      if (q < Lk) {
        const int m = q >> 6, bit = q & 63;
        const unsigned long long hm = s.fmask[m] & below(Lk, m);
        if ((hm >> bit) & 1ull) {
          const int k = s.pre_hole[m] + __popcll(hm & ((1ull << bit) - 1ull));
          const int from = s.src[k], to = i + 1 + q;
          s.x1[to] = s.x1[from];
          s.y1[to] = s.y1[from];
          s.x2[to] = s.x2[from];
          s.y2[to] = s.y2[from];
          s.sc[to] = s.sc[from];
          s.ind[to] = s.ind[from];
        }
      }
    }
    __syncthreads();
    N = i + 1 + Lk;
  }
  __syncthreads();
  for (int p = tid; p < N; p += kThreads) {  // boxes[:N], inds[:N] (:203)
    out_dets[p * 5 + 0] = s.x1[p];
    out_dets[p * 5 + 1] = s.y1[p];
    out_dets[p * 5 + 2] = s.x2[p];
    out_dets[p * 5 + 3] = s.y2[p];
    out_dets[p * 5 + 4] = s.sc[p];
    out_inds[p] = s.ind[p];
  }
  if (tid == 0) *num_out = N;
}

__global__ void soft_nms_write_zero(int* num_out) { *num_out = 0; }

}  // namespace

extern "C" int mi_soft_nms(const float* dets, int n, float sigma, float overlap_thresh, float score_thresh, int method,
                           float* out_dets, int64_t* out_inds, int32_t* num_out, mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(n >= 0, "soft_nms: negative box count");
  MI_REQUIRE(n <= kMaxBoxes, "soft_nms: %d boxes, at most %d are supported", n, kMaxBoxes);
  MI_REQUIRE(method >= 0 && method <= 2, "soft_nms: unknown method %d (0 hard, 1 linear, 2 gaussian)", method);
  MI_REQUIRE(num_out != nullptr, "soft_nms: null num_out");
  hipStream_t s = mi::as_stream(stream);
  if (n == 0) {
    soft_nms_write_zero<<<1, 1, 0, s>>>(num_out);
    return mi::check_launch("soft_nms_write_zero");
  }
  MI_REQUIRE(dets != nullptr && out_dets != nullptr && out_inds != nullptr, "soft_nms: null pointer");
  const size_t lds = Lds::bytes(n);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&soft_nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  soft_nms_kernel<<<1, kThreads, lds, s>>>(dets, n, nullptr, sigma, overlap_thresh, score_thresh, method, out_dets,
                                           reinterpret_cast<long long*>(out_inds), num_out);
  return mi::check_launch("soft_nms_kernel");
}

extern "C" int mi_soft_nms_segmented(const float* dets, const int32_t* offsets, int num_segments, int max_segment,
                                     float sigma, float overlap_thresh, float score_thresh, int method,
                                     float* out_dets, int64_t* out_inds, int32_t* num_out, mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(num_segments >= 0 && max_segment >= 0, "soft_nms_segmented: negative size");
  MI_REQUIRE(max_segment <= kMaxBoxes, "soft_nms_segmented: segments of up to %d boxes, at most %d are supported",
             max_segment, kMaxBoxes);
  MI_REQUIRE(method >= 0 && method <= 2, "soft_nms_segmented: unknown method %d (0 hard, 1 linear, 2 gaussian)", method);
  if (num_segments == 0) return MI_OK;
  MI_REQUIRE(offsets != nullptr && num_out != nullptr, "soft_nms_segmented: null pointer");
  MI_REQUIRE(max_segment == 0 || (dets != nullptr && out_dets != nullptr && out_inds != nullptr),
             "soft_nms_segmented: null pointer");
  const int cap = max_segment > 0 ? max_segment : 1;
  const size_t lds = Lds::bytes(cap);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&soft_nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  soft_nms_kernel<<<num_segments, kThreads, lds, mi::as_stream(stream)>>>(
      dets, cap, reinterpret_cast<const int*>(offsets), sigma, overlap_thresh, score_thresh, method, out_dets,
      reinterpret_cast<long long*>(out_inds), num_out);
  return mi::check_launch("soft_nms_kernel");
}
