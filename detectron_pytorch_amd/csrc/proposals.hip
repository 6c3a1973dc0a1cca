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

}  // namespace

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
