// box_voting.hip -- bounding-box voting (lib/utils/boxes.py:268-317; call site lib/core/test.py:766-773) on the device,
// all classes of an image in one launch.
//
// For every detection that survived NMS ("top"), the candidates of its class ("all": every row above the score threshold,
// before NMS) whose IoU with it is >= thresh vote for its coordinates: score-weighted average of their boxes.  Optionally
// the score is replaced by a statistic of the voters' scores (TEST.BBOX_VOTE.SCORING_METHOD).
//
// One wavefront per top row; lanes stride over the class's segment of `all`.  The IoU is utils.cython_bbox.bbox_overlaps
// bit for bit (the fp64 intermediates of the Cython-generated C, as in nms.hip: bbox_overlaps_kernel), so the voter SET
// equals the reference's; the averages are accumulated in fp64 and rounded once, where numpy sums fp32 pairwise -- the
// parity contract is 1e-5 relative, asserted 2e-6 (tests/test_ops_gpu.py).  HBM traffic is 20 bytes per candidate per
// top row of its class, L2-resident: latency-bound, ~10 us per image.
#include "common.h"

namespace {

enum : int { kId = 0, kTempAvg = 1, kAvg = 2, kIouAvg = 3, kGeneralizedAvg = 4, kQuasiSum = 5 };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

__global__ void __launch_bounds__(256)
box_voting_kernel(const float* __restrict__ top, const int* __restrict__ top_seg, const float* __restrict__ all,
                  const int* __restrict__ all_off, int num_top, float thresh, int method, float beta,
                  float* __restrict__ out) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= num_top) return;
  const float* b = top + (long long)k * 5;
  const float b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
  const int seg = top_seg[k];
  const int lo = all_off[seg], hi = all_off[seg + 1];
  double sw = 0., sx1 = 0., sy1 = 0., sx2 = 0., sy2 = 0., stat = 0., sov = 0., cnt = 0.;
  for (int i = lo + lane; i < hi; i += 64) {
    const float* q = all + (long long)i * 5;
    // cython_bbox.pyx:52-72, boxes = top row, query = candidate (the argument order of boxes.py:279)
    const float box_area = (float)(((double)(q[2] - q[0]) + 1.0) * ((double)(q[3] - q[1]) + 1.0));
    float ov = 0.f;
    const float iw = (float)((double)((b2 <= q[2] ? b2 : q[2]) - (b0 >= q[0] ? b0 : q[0])) + 1.0);
    if (iw > 0) {
      const float ih = (float)((double)((b3 <= q[3] ? b3 : q[3]) - (b1 >= q[1] ? b1 : q[1])) + 1.0);
      if (ih > 0) {
        const float ua =
            (float)(((((double)(b2 - b0) + 1.0) * ((double)(b3 - b1) + 1.0)) + (double)box_area) - (double)(iw * ih));
        ov = iw * ih / ua;
      }
    }
    if (!(ov >= thresh)) continue;
    const float w = q[4];
    sw += (double)w;
    sx1 += (double)w * (double)q[0];
    sy1 += (double)w * (double)q[1];
    sx2 += (double)w * (double)q[2];
    sy2 += (double)w * (double)q[3];
    cnt += 1.;
    if (method == kTempAvg) {
      // boxes.py:292-297: P = (w, 1 - w), smoothed with temperature beta, P(class) averaged
      const float pa = w, pb = 1.0f - w;
      const float pm = pa > pb ? pa : pb;
      const float ea = expf(logf(pa / pm) / beta), eb = expf(logf(pb / pm) / beta);
      stat += (double)(ea / (ea + eb));
    } else if (method == kIouAvg) {
      stat += (double)w * (double)ov;
      sov += (double)ov;
    } else if (method == kGeneralizedAvg) {
      stat += (double)powf(w, beta);
    }
  }
  sw = wave_sum(sw);
  sx1 = wave_sum(sx1);
  sy1 = wave_sum(sy1);
  sx2 = wave_sum(sx2);
  sy2 = wave_sum(sy2);
  cnt = wave_sum(cnt);
  if (method == kTempAvg || method == kIouAvg || method == kGeneralizedAvg) stat = wave_sum(stat);
  if (method == kIouAvg) sov = wave_sum(sov);
  if (lane != 0) return;
  float* o = out + (long long)k * 5;
  float score = b[4];
  if (cnt > 0. && sw != 0.) {  // a top row is among the candidates of its class, so it always votes for itself
    o[0] = (float)(sx1 / sw);
    o[1] = (float)(sy1 / sw);
    o[2] = (float)(sx2 / sw);
    o[3] = (float)(sy2 / sw);
    if (method == kTempAvg) score = (float)(stat / cnt);
    else if (method == kAvg) score = (float)(sw / cnt);
    else if (method == kIouAvg) score = (float)(stat / sov);
    else if (method == kGeneralizedAvg) score = powf((float)(stat / cnt), 1.0f / beta);
    else if (method == kQuasiSum) score = (float)(sw / pow(cnt, (double)beta));
  } else {
    o[0] = b0;
    o[1] = b1;
    o[2] = b2;
    o[3] = b3;
  }
  o[4] = score;
}

}  // namespace

extern "C" int mi_box_voting(const float* top_dets, const int32_t* top_segments, int num_top, const float* all_dets,
                             const int32_t* all_offsets, int num_segments, float thresh, int scoring_method, float beta,
                             float* out, mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(num_top >= 0 && num_segments >= 0, "box_voting: negative size");
  MI_REQUIRE(scoring_method >= kId && scoring_method <= kQuasiSum, "box_voting: unknown scoring method %d", scoring_method);
  if (num_top == 0) return MI_OK;
  MI_REQUIRE(top_dets != nullptr && top_segments != nullptr && all_dets != nullptr && all_offsets != nullptr &&
                 out != nullptr, "box_voting: null pointer");
  box_voting_kernel<<<(num_top + 3) / 4, 256, 0, mi::as_stream(stream)>>>(top_dets, top_segments, all_dets, all_offsets,
                                                                          num_top, thresh, scoring_method, beta, out);
  return mi::check_launch("box_voting");
}
