// proposals.hip -- RPN proposal decode for gfx950 (C-ABI mi_rpn_decode_proposals): steps 1-3 of
// GenerateProposalsOp.proposals_for_one_image (lib/modeling/generate_proposals.py:105-153) for the pre-NMS top-k
// anchors of every image of a level, in one launch:
//   anchor of the k-th best score (generated on the fly: base anchor + cell shift, :66-88) -> bbox_transform
//   (lib/utils/boxes.py:156-196, weights 1) -> clip to the image (:138-153) -> min-size / centre filter (:170-182).
// Arithmetic types are those numpy >= 2 gives the reference's expressions (see oracle/proposals.py): fp32 throughout,
// except the width / height branch, which np.minimum(dw, cfg.BBOX_XFORM_CLIP) promotes to fp64 (the clip constant is an
// np.float64 scalar): exp, the product with the anchor size and the final +- 0.5 * size are evaluated in double and
// rounded to fp32 once, on the store.  Compiled with -ffp-contract=off.
// Boxes the filter rejects are written as a far-away degenerate box and flagged in `valid`: they ride through the NMS
// without touching any real box (IoU 0) and are dropped afterwards, which spares a compaction pass (and its host sync)
// between the decode and the NMS.
#include "common.h"

namespace {

constexpr int kMaxAnchors = 16;
struct AnchorTable {
  int count;
  double a[kMaxAnchors][4];
};

__global__ void __launch_bounds__(256)
rpn_decode_kernel(const float* __restrict__ bbox_pred, const float* __restrict__ topk_scores,
                  const long long* __restrict__ topk_idx, const float* __restrict__ im_info, const AnchorTable anchors,
                  int num_images, int num_anchors, int height, int width, int k, double feat_stride, float min_size,
                  double xform_clip, float* __restrict__ dets, int* __restrict__ valid) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)num_images * k) return;
  const int n = (int)(t / k);
  const long long idx = topk_idx[t];  // flat index into the image's [A, H, W] score map
  const int plane = height * width;
  const int a = (int)(idx / plane), rem = (int)(idx - (long long)a * plane);
  const int h = rem / width, w = rem - h * width;
  // shifted anchor in double (generate_proposals.py:69-88), cast to the deltas' dtype (boxes.py:164)
  const double sx = (double)w * feat_stride, sy = (double)h * feat_stride;
  const float b0 = (float)(anchors.a[a][0] + sx), b1 = (float)(anchors.a[a][1] + sy);
  const float b2 = (float)(anchors.a[a][2] + sx), b3 = (float)(anchors.a[a][3] + sy);
  const float* d = bbox_pred + (((long long)n * 4 * num_anchors + 4 * a) * height + h) * width + w;
  const float dx = d[0] / 1.0f, dy = d[plane] / 1.0f;                       // :172-175, weights (1, 1, 1, 1)
  const double dw = fmin((double)d[2 * (long long)plane], xform_clip);       // :178-179 (float64 from here)
  const double dh = fmin((double)d[3 * (long long)plane], xform_clip);
  const float widths = b2 - b0 + 1.0f, heights = b3 - b1 + 1.0f;            // :166-169
  const float ctr_x = b0 + 0.5f * widths, ctr_y = b1 + 0.5f * heights;
  const float pred_ctr_x = dx * widths + ctr_x, pred_ctr_y = dy * heights + ctr_y;  // :181-182 (fp32, no FMA)
  const double pred_w = exp(dw) * (double)widths, pred_h = exp(dh) * (double)heights;  // :183-184
  float x1 = (float)((double)pred_ctr_x - 0.5 * pred_w);                    // :188-194, rounded on the store
  float y1 = (float)((double)pred_ctr_y - 0.5 * pred_h);
  float x2 = (float)((double)pred_ctr_x + 0.5 * pred_w - 1.0);
  float y2 = (float)((double)pred_ctr_y + 0.5 * pred_h - 1.0);
  const float im_h = im_info[n * 3 + 0], im_w = im_info[n * 3 + 1], im_scale = im_info[n * 3 + 2];
  x1 = fmaxf(fminf(x1, im_w - 1.f), 0.f);                                   // clip_tiled_boxes, boxes.py:146-152
  y1 = fmaxf(fminf(y1, im_h - 1.f), 0.f);
  x2 = fmaxf(fminf(x2, im_w - 1.f), 0.f);
  y2 = fmaxf(fminf(y2, im_h - 1.f), 0.f);
  const float ms = min_size * im_scale;                                      // _filter_boxes, :170-182
  const float ws = x2 - x1 + 1.f, hs = y2 - y1 + 1.f;
  const float x_ctr = x1 + ws / 2.f, y_ctr = y1 + hs / 2.f;
  const bool ok = ws >= ms && hs >= ms && x_ctr < im_w && y_ctr < im_h;
  float* o = dets + t * 5;
  const float far = -1.0e6f;
  o[0] = ok ? x1 : far;
  o[1] = ok ? y1 : far;
  o[2] = ok ? x2 : far;
  o[3] = ok ? y2 : far;
  o[4] = topk_scores[t];
  valid[t] = ok ? 1 : 0;
}

// ---- collect: steps 6-8 of proposals_for_one_image (generate_proposals.py:155-161) for every (level, image) problem
// and the concatenation of collect_and_distribute_fpn_rpn_proposals.py:83-90, in one launch.  A problem's candidates
// are its k decoded boxes in descending score order; a candidate is TAKEN when the NMS kept it, the size filter passed
// it, and it is among the first post_nms_topN such boxes.  Every candidate gets a row of the flat arrays (its score, or
// -inf when it is not taken; (image, x1, y1, x2, y2)), so the global top-k that follows has fixed shapes. ----
constexpr int kMaxCollect = 16;
constexpr int kCollectThreads = 1024;
constexpr int kMaxCandidates = 4096;   // per problem == the NMS batch limit
struct CollectProblem {
  const float* dets;          // [k, 5]
  const int* valid;           // [k]
  const long long* keep;      // ascending kept positions (nullptr: no NMS ran, everything is kept)
  const int* num_keep;
  int k, image, offset;
};
struct CollectTable {
  int count, post_nms_topn;
  CollectProblem p[kMaxCollect];
};

__global__ void __launch_bounds__(kCollectThreads)
rpn_collect_candidates(const CollectTable t, float* __restrict__ cand_scores, float* __restrict__ cand_rois) {
  __shared__ int s_take[kMaxCandidates];
  __shared__ int s_wave[kCollectThreads / 64];
  const CollectProblem p = t.p[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = tid; j < p.k; j += kCollectThreads) s_take[j] = p.keep == nullptr ? (p.valid[j] != 0) : 0;
  __syncthreads();
  if (p.keep != nullptr) {
    const int nk = min(*p.num_keep, p.k);
    for (int j = tid; j < nk; j += kCollectThreads) {
      const long long pos = p.keep[j];
      if (pos >= 0 && pos < p.k && p.valid[pos] != 0) s_take[pos] = 1;
    }
  }
  __syncthreads();
  // rank of every taken candidate in score order: contiguous chunks per thread, block-wide exclusive scan
  const int per = (p.k + kCollectThreads - 1) / kCollectThreads;
  const int begin = min(tid * per, p.k), end = min(begin + per, p.k);
  int local = 0;
  for (int j = begin; j < end; j++) local += s_take[j];
  int incl = local;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int rank = incl - local;
  for (int w = 0; w < wave; w++) rank += s_wave[w];
  for (int j = begin; j < end; j++) {
    bool taken = s_take[j] != 0;
    if (taken) {
      rank++;
      taken = t.post_nms_topn <= 0 || rank <= t.post_nms_topn;
    }
    const float* d = p.dets + (long long)j * 5;
    const long long row = p.offset + j;
    cand_scores[row] = taken ? d[4] : -__builtin_inff();
    float* o = cand_rois + row * 5;
    o[0] = (float)p.image;
    o[1] = d[0];
    o[2] = d[1];
    o[3] = d[2];
    o[4] = d[3];
  }
}

// The rows the global top-k picked, as the RoI blob the heads consume: (image, x1, y1, x2, y2), whether the row is a
// proposal at all (fewer candidates than rows: score -inf), and its FPN level (utils/fpn.py:11-28, fp32 in the
// reference's operation order: floor(lvl0 + log2(sqrt(area) / s0 + 1e-6)) clamped to [k_min, k_max]).
__global__ void __launch_bounds__(256)
rpn_collect_finish(const float* __restrict__ top_scores, const long long* __restrict__ top_idx,
                   const float* __restrict__ cand_rois, int rows, int mark_invalid, int k_min, int k_max, float s0,
                   float lvl0, float* __restrict__ rois, unsigned char* __restrict__ valid, int* __restrict__ levels) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* c = cand_rois + top_idx[r] * 5;
  const bool ok = top_scores[r] > -__builtin_inff();
  const float x1 = c[1], y1 = c[2], x2 = c[3], y2 = c[4];
  float* o = rois + (long long)r * 5;
  o[0] = (ok || !mark_invalid) ? c[0] : -1.f;
  o[1] = x1;
  o[2] = y1;
  o[3] = x2;
  o[4] = y2;
  valid[r] = ok ? 1 : 0;
  const float w = x2 - x1 + 1.f, h = y2 - y1 + 1.f;
  float area = w * h;
  area = area < 0.f ? 0.f : area;             // areas[neg_idx] = 0 (utils/boxes.py:113-121 via fpn.py:18)
  const float s = sqrtf(area);
  float lvl = floorf(lvl0 + log2f(s / s0 + 1e-6f));
  lvl = fminf(fmaxf(lvl, (float)k_min), (float)k_max);
  levels[r] = (int)lvl;
}

}  // namespace

extern "C" int mi_rpn_collect_candidates(int num_problems, const float* const* dets, const int32_t* const* valid,
                                         const int64_t* const* keep, const int32_t* const* num_keep, const int* k,
                                         const int* image, int post_nms_topn, float* cand_scores, float* cand_rois,
                                         mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(num_problems >= 0, "rpn_collect_candidates: negative problem count");
  if (num_problems == 0) return MI_OK;
  MI_REQUIRE(dets != nullptr && valid != nullptr && k != nullptr && image != nullptr && cand_scores != nullptr &&
                 cand_rois != nullptr,
             "rpn_collect_candidates: null pointer");
  hipStream_t s = mi::as_stream(stream);
  int offset = 0;
  for (int first = 0; first < num_problems; first += kMaxCollect) {
    CollectTable t;
    t.count = num_problems - first < kMaxCollect ? num_problems - first : kMaxCollect;
    t.post_nms_topn = post_nms_topn;
    for (int q = 0; q < t.count; q++) {
      const int i = first + q;
      MI_REQUIRE(k[i] >= 0 && k[i] <= kMaxCandidates, "rpn_collect_candidates: problem %d has %d candidates (max %d)", i,
                 k[i], kMaxCandidates);
      MI_REQUIRE(k[i] == 0 || (dets[i] != nullptr && valid[i] != nullptr), "rpn_collect_candidates: null pointer");
      const bool with_nms = keep != nullptr && keep[i] != nullptr;
      MI_REQUIRE(!with_nms || (num_keep != nullptr && num_keep[i] != nullptr), "rpn_collect_candidates: null num_keep");
      t.p[q].dets = dets[i];
      t.p[q].valid = valid[i];
      t.p[q].keep = with_nms ? reinterpret_cast<const long long*>(keep[i]) : nullptr;
      t.p[q].num_keep = with_nms ? num_keep[i] : nullptr;
      t.p[q].k = k[i];
      t.p[q].image = image[i];
      t.p[q].offset = offset;
      offset += k[i];
    }
    rpn_collect_candidates<<<t.count, kCollectThreads, 0, s>>>(t, cand_scores, cand_rois);
    int rc = mi::check_launch("rpn_collect_candidates");
    if (rc != MI_OK) return rc;
  }
  return MI_OK;
}

extern "C" int mi_rpn_collect_finish(const float* top_scores, const int64_t* top_indices, const float* cand_rois,
                                     int rows, int mark_invalid, int k_min, int k_max, float canonical_scale,
                                     float canonical_level, float* rois, uint8_t* valid, int32_t* levels,
                                     mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(rows >= 0 && k_min <= k_max, "rpn_collect_finish: bad size");
  if (rows == 0) return MI_OK;
  MI_REQUIRE(top_scores != nullptr && top_indices != nullptr && cand_rois != nullptr && rois != nullptr &&
                 valid != nullptr && levels != nullptr,
             "rpn_collect_finish: null pointer");
  rpn_collect_finish<<<mi::ceil_div(rows, 256), 256, 0, mi::as_stream(stream)>>>(
      top_scores, reinterpret_cast<const long long*>(top_indices), cand_rois, rows, mark_invalid, k_min, k_max,
      canonical_scale, canonical_level, rois, valid, levels);
  return mi::check_launch("rpn_collect_finish");
}

// Row i of a pyramid's RoI blob in dataloader order sits at position restore[i] of the level-major concatenation
// (utils/fpn.py:31-58 builds rois_idx_restore_int32 that way): its level is the one whose rows span that position.
struct LevelSpans {
  int count;
  int end[8];    // exclusive prefix ends of the levels' row counts, level-major
  int value[8];  // what to write for a row of that span
};
template <typename Index>
__global__ void __launch_bounds__(256)
fpn_level_of_restore(const Index* __restrict__ restore, int rows, const LevelSpans spans, int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  const long long pos = (long long)restore[i];
  int v = spans.value[spans.count - 1];
  for (int k = spans.count - 1; k >= 0; k--)
    if (pos < spans.end[k]) v = spans.value[k];
  out[i] = v;
}

extern "C" int mi_fpn_level_index_from_restore(const void* restore, int restore_is_int64, int rows, int num_spans,
                                               const int* span_rows_host, const int* span_value_host, int32_t* out,
                                               mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(rows >= 0 && num_spans >= 1 && num_spans <= 8, "fpn_level_index_from_restore: 1..8 spans");
  if (rows == 0) return MI_OK;
  MI_REQUIRE(restore != nullptr && span_rows_host != nullptr && span_value_host != nullptr && out != nullptr,
             "fpn_level_index_from_restore: null pointer");
  LevelSpans sp;
  sp.count = num_spans;
  int run = 0;
  for (int k = 0; k < num_spans; k++) {
    MI_REQUIRE(span_rows_host[k] >= 0, "fpn_level_index_from_restore: negative span");
    run += span_rows_host[k];
    sp.end[k] = run;
    sp.value[k] = span_value_host[k];
  }
  if (restore_is_int64)
    fpn_level_of_restore<long long><<<mi::ceil_div(rows, 256), 256, 0, mi::as_stream(stream)>>>(
        static_cast<const long long*>(restore), rows, sp, out);
  else
    fpn_level_of_restore<int><<<mi::ceil_div(rows, 256), 256, 0, mi::as_stream(stream)>>>(static_cast<const int*>(restore),
                                                                                          rows, sp, out);
  return mi::check_launch("fpn_level_of_restore");
}

extern "C" int mi_rpn_decode_proposals(const float* bbox_pred, const float* topk_scores, const int64_t* topk_idx,
                                       const float* im_info, const double* base_anchors_host, int num_images,
                                       int num_anchors, int height, int width, int k, double feat_stride,
                                       float min_size, double xform_clip, float* dets, int32_t* valid,
                                       mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(num_images >= 0 && num_anchors > 0 && height > 0 && width > 0 && k >= 0, "rpn_decode: bad size");
  MI_REQUIRE(num_anchors <= kMaxAnchors, "rpn_decode: %d anchors per cell, at most %d are supported", num_anchors,
             kMaxAnchors);
  const long long total = (long long)num_images * k;
  if (total == 0) return MI_OK;
  MI_REQUIRE(bbox_pred != nullptr && topk_scores != nullptr && topk_idx != nullptr && im_info != nullptr &&
                 base_anchors_host != nullptr && dets != nullptr && valid != nullptr,
             "rpn_decode: null pointer");
  AnchorTable t;
  t.count = num_anchors;
  for (int a = 0; a < num_anchors; a++)
    for (int c = 0; c < 4; c++) t.a[a][c] = base_anchors_host[a * 4 + c];
  rpn_decode_kernel<<<mi::ceil_div(total, 256), 256, 0, mi::as_stream(stream)>>>(
      bbox_pred, topk_scores, reinterpret_cast<const long long*>(topk_idx), im_info, t, num_images, num_anchors, height,
      width, k, feat_stride, min_size, xform_clip, dets, reinterpret_cast<int*>(valid));
  return mi::check_launch("rpn_decode_kernel");
}
