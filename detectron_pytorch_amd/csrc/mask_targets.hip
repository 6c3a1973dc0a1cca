// mask_targets.hip -- the polygon side of the Mask R-CNN training targets on the device: every foreground RoI's M x M
// binary target rasterised from the polygons of the ground-truth instance it was matched to, all RoIs of a step in one
// launch.  Replaces the per-RoI host loop of lib/roi_data/mask_rcnn.py:66-76 over
// lib/utils/segms.py:93-119 polys_to_mask_wrt_box, i.e. pycocotools 2.0 mask_util.frPyObjects + mask_util.decode
// (common/maskApi.c rleFrPoly, rleDecode -- a third-party package that is not part of the reference tree; its published
// procedure is what is computed here, oracle/oracle.c oracle_poly_to_mask states it step by step).
//
// rleFrPoly is sequential as written (a chain of boundary points, a sort, a run-length merge).  What it computes is not:
//   * a boundary point depends only on its edge and its step along it, and a crossing only on two consecutive points --
//     within an edge, or the last point of one edge and the first of the next;
//   * the sorted, zero-run-merged run lengths say "the pixel at column-major position i is set iff an odd number of
//     crossings lie at positions <= i": crossings TOGGLE, two at one position cancel.
// So: one workgroup per RoI; per polygon, the up-sampled vertices go to LDS (a lane per vertex), then a lane per edge
// walks its steps (edges of more than 16 steps are queued and walked by a wavefront each) and XORs crossings into an
// LDS array of M*M + 1 toggles; a parity prefix scan over that array is the polygon's mask; the polygons of the
// instance are OR-ed (segms.py:117-118 sums and thresholds).  All coordinate arithmetic in the precision and order of
// the originals: float32 for the move into the RoI's frame (numpy, segms.py:104-112), then double with C casts.
// Bit-exact against the oracle (tests/test_ops_gpu.py).  Work is tiny (a few thousand boundary points per RoI): the
// launch is latency-bound, tens of microseconds for a step's 256 foreground RoIs.
#include "common.h"

namespace {

constexpr int kMaxM = 64;  // LDS: (M*M + 1) toggles + M*M accumulated mask words
constexpr int kThreads = 256;
constexpr int kChunk = 1024;     // edges of a polygon handled per pass (their up-sampled vertices sit in LDS)
constexpr int kShortEdge = 16;  // steps one lane walks by itself; longer edges are walked by a wavefront

struct Edge {
  int xs, ys, dx, dy, flip, n;  // start AFTER the flip; n = points on the edge
  double slope;
};

// vertex j of the polygon in the RoI's frame, up-sampled: segms.py:108-111 in float32, then maskApi.c's (int)(5 v + .5)
__device__ __forceinline__ void vertex(const float* __restrict__ pts, int j, float bx, float by, float fm, float w, float h,
                                       int& x, int& y) {
  const float px = __fdiv_rn(__fmul_rn(__fsub_rn(pts[2 * j], bx), fm), w);
  const float py = __fdiv_rn(__fmul_rn(__fsub_rn(pts[2 * j + 1], by), fm), h);
  x = (int)__dadd_rn(__dmul_rn(5.0, (double)px), .5);
  y = (int)__dadd_rn(__dmul_rn(5.0, (double)py), .5);
}

__device__ __forceinline__ Edge make_edge(int xs, int ys, int xe, int ye) {
  Edge e;
  e.dx = abs(xe - xs);
  e.dy = abs(ys - ye);
  e.flip = (e.dx >= e.dy && xs > xe) || (e.dx < e.dy && ys > ye);
  if (e.flip) {
    int t = xs; xs = xe; xe = t;
    t = ys; ys = ye; ye = t;
  }
  e.xs = xs;
  e.ys = ys;
  // 0 / 0 for an edge of one point: its other coordinate is never consumed (both neighbours share its x)
  e.slope = e.dx >= e.dy ? __ddiv_rn((double)(ye - ys), (double)e.dx) : __ddiv_rn((double)(xe - xs), (double)e.dy);
  e.n = (e.dx >= e.dy ? e.dx : e.dy) + 1;
  return e;
}

// point d (0 .. n-1, in emission order: from the edge's first vertex to its second)
__device__ __forceinline__ void point(const Edge& e, int d, int& u, int& v) {
  if (e.dx >= e.dy) {
    const int t = e.flip ? e.dx - d : d;
    u = t + e.xs;
    v = (int)__dadd_rn(__dadd_rn((double)e.ys, __dmul_rn(e.slope, (double)t)), .5);
  } else {
    const int t = e.flip ? e.dy - d : d;
    v = t + e.ys;
    u = (int)__dadd_rn(__dadd_rn((double)e.xs, __dmul_rn(e.slope, (double)t)), .5);
  }
}

// a crossing between consecutive points (up, vp) -> (u, v): toggles one column-major position (maskApi.c, step 3)
__device__ __forceinline__ void crossing(int up, int vp, int u, int v, int m, unsigned* toggles) {
  if (u == up) return;
  double xd = (double)(u < up ? u : u - 1);
  xd = __dsub_rn(__ddiv_rn(__dadd_rn(xd, .5), 5.0), .5);
  if (floor(xd) != xd || xd < 0 || xd > (double)(m - 1)) return;
  double yd = (double)(v < vp ? v : vp);
  yd = __dsub_rn(__ddiv_rn(__dadd_rn(yd, .5), 5.0), .5);
  if (yd < 0) yd = 0;
  else if (yd > (double)m) yd = (double)m;
  yd = ceil(yd);
  atomicXor(&toggles[(int)xd * m + (int)yd], 1u);
}

// crossings of edge `e` between its points [d0, d1), walked by one lane; (up, vp) = the point in front of d0
__device__ __forceinline__ void walk(const Edge& e, int d0, int d1, int up, int vp, int m, unsigned* toggles) {
  for (int d = d0; d < d1; d++) {
    int u, v;
    point(e, d, u, v);
    crossing(up, vp, u, v, m, toggles);
    up = u;
    vp = v;
  }
}

__global__ void __launch_bounds__(kThreads)
polys_to_masks_kernel(const float* __restrict__ poly_xy, const int* __restrict__ poly_start,
                      const int* __restrict__ inst_start, const int* __restrict__ roi_inst,
                      const float* __restrict__ rois, int* __restrict__ masks, int num_instances, int m) {
  __shared__ unsigned toggles[kMaxM * kMaxM + 1];
  __shared__ unsigned acc[kMaxM * kMaxM];
  __shared__ int vx[kChunk + 2], vy[kChunk + 2];  // up-sampled vertices c0 - 1 .. c0 + kChunk of the polygon
  __shared__ int long_edges[kThreads];
  __shared__ int num_long;
  __shared__ unsigned wave_parity[kThreads / 64];
  const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int mm = m * m;
  int* out = masks + (long long)r * mm;
  const int inst = roi_inst[r];
  if (inst < 0 || inst >= num_instances) {  // a padding row: no instance, all zeros
    for (int i = tid; i < mm; i += kThreads) out[i] = 0;
    return;
  }
  const float bx = rois[r * 4], by = rois[r * 4 + 1];
  const float w = fmaxf(__fsub_rn(rois[r * 4 + 2], bx), 1.f), h = fmaxf(__fsub_rn(rois[r * 4 + 3], by), 1.f);  // segms.py:99-103
  const float fm = (float)m;
  for (int i = tid; i < mm; i += kThreads) acc[i] = 0u;
  const int per = (mm + kThreads - 1) / kThreads;
  for (int p = inst_start[inst]; p < inst_start[inst + 1]; p++) {
    const float* pts = poly_xy + 2LL * poly_start[p];
    const int k = poly_start[p + 1] - poly_start[p];
    for (int i = tid; i <= mm; i += kThreads) toggles[i] = 0u;
    for (int c0 = 0; c0 < k; c0 += kChunk) {
      const int cnt = k - c0 < kChunk ? k - c0 : kChunk;  // edges c0 .. c0 + cnt - 1 of the closed outline
      if (tid == 0) num_long = 0;
      for (int i = tid; i < cnt + 2; i += kThreads) {     // slot i = vertex c0 - 1 + i (vertex k = vertex 0)
        const int j = c0 - 1 + i;
        if (j >= 0) vertex(pts, j < k ? j : 0, bx, by, fm, w, h, vx[i], vy[i]);
      }
      __syncthreads();
      // a lane per edge: most edges are a few steps long (a 28 x 28 target is 140 samples wide); the long ones queue up
      for (int i = tid; i < cnt; i += kThreads) {
        const Edge edge = make_edge(vx[i + 1], vy[i + 1], vx[i + 2], vy[i + 2]);
        if (edge.n > kShortEdge) {
          const int q = atomicAdd(&num_long, 1);
          if (q < kThreads) {
            long_edges[q] = i;
            continue;
          }
        }
        int u0, v0;
        point(edge, 0, u0, v0);
        if (c0 + i > 0) {  // the point in front of the edge's first: the last point of the edge before (none at the chain's start)
          const Edge before = make_edge(vx[i], vy[i], vx[i + 1], vy[i + 1]);
          int up, vp;
          point(before, before.n - 1, up, vp);
          crossing(up, vp, u0, v0, m, toggles);
        }
        walk(edge, 1, edge.n, u0, v0, m, toggles);
      }
      __syncthreads();
      const int nl = num_long < kThreads ? num_long : kThreads;
      for (int q = wave; q < nl; q += kThreads / 64) {  // long edges: a wavefront each, lanes stride over the steps
        const int i = long_edges[q];
        const Edge edge = make_edge(vx[i + 1], vy[i + 1], vx[i + 2], vy[i + 2]);
        for (int d = lane; d < edge.n; d += 64) {
          int u, v, up, vp;
          point(edge, d, u, v);
          if (d > 0) {
            point(edge, d - 1, up, vp);
          } else {
            if (c0 + i == 0) continue;
            const Edge before = make_edge(vx[i], vy[i], vx[i + 1], vy[i + 1]);
            point(before, before.n - 1, up, vp);
          }
          crossing(up, vp, u, v, m, toggles);
        }
      }
      __syncthreads();
    }
    // parity prefix over the column-major positions: every lane owns `per` consecutive ones
    const int lo = tid * per, hi = lo + per < mm ? lo + per : mm;
    unsigned mine = 0u;
    for (int i = lo; i < hi; i++) mine ^= toggles[i];
    const unsigned long long bal = __ballot(mine & 1u);
    if (lane == 0) wave_parity[wave] = (unsigned)__popcll(bal) & 1u;
    __syncthreads();
    unsigned run = (unsigned)__popcll(bal & ((1ull << lane) - 1ull)) & 1u;
    for (int q = 0; q < wave; q++) run ^= wave_parity[q];
    for (int i = lo; i < hi; i++) {
      run ^= toggles[i] & 1u;
      acc[i] |= run;
    }
    __syncthreads();
  }
  __syncthreads();
  // column-major (x * m + y) -> the blob's row-major y * m + x (mask_rcnn.py:76 reshapes the [M, M] image)
  for (int i = tid; i < mm; i += kThreads) {
    const int y = i / m, x = i - y * m;
    out[i] = (int)acc[x * m + y];
  }
}

}  // namespace

extern "C" int mi_polys_to_masks_wrt_boxes(const float* poly_xy, const int32_t* poly_start, const int32_t* inst_start,
                                           const int32_t* roi_inst, const float* rois, int32_t* masks, int num_rois,
                                           int num_instances, int m, mi_stream_t stream) {
  mi::begin_call();
  MI_REQUIRE(num_rois >= 0 && num_instances >= 0, "polys_to_masks_wrt_boxes: negative size");
  MI_REQUIRE(m >= 1, "polys_to_masks_wrt_boxes: resolution %d", m);
  if (m > kMaxM) {
    mi::set_error("polys_to_masks_wrt_boxes: resolution %d > %d", m, kMaxM);
    return MI_ERR_UNSUPPORTED;
  }
  if (num_rois == 0) return MI_OK;
  MI_REQUIRE(roi_inst != nullptr && rois != nullptr && masks != nullptr, "polys_to_masks_wrt_boxes: null pointer");
  MI_REQUIRE(num_instances == 0 || (poly_xy != nullptr && poly_start != nullptr && inst_start != nullptr),
             "polys_to_masks_wrt_boxes: null polygon arrays");
  polys_to_masks_kernel<<<num_rois, kThreads, 0, mi::as_stream(stream)>>>(poly_xy, poly_start, inst_start, roi_inst, rois,
                                                                          masks, num_instances, m);
  return mi::check_launch("polys_to_masks_wrt_boxes");
}
