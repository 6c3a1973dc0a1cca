// affine_channel.hip -- the frozen-BatchNorm chain of the ResNet bottleneck as ONE streaming pass for gfx950.
//
// Reference: lib/nn/modules/affine.py:5-17 (AffineChannel2d: x * w[c] + b[c], applied after every backbone convolution),
// followed in lib/modeling/ResNet.py:246-293 by ReLU (in place) or by "+= residual" and ReLU.  In PyTorch that is three to
// four element-wise kernels per convolution, each a full read + write of the activation (137.6 MB for a res2 output of
// two 800x1344 images): rocprofv3 puts them at ~20 % of the GPU time of a training step, next to MIOpen's fp32
// convolutions.  SURVEY.md section 2b lists the chain as a fusion candidate outside the hot path; it is built because it
// is what stands between the operators of this library and the images/s the metric is quoted in.
//
//   forward    y = relu?( x * w[c] + b[c] (+ r) )       one read of x (and r), one write of y
//   backward   dx = dy * [y > 0]? * w[c],  dr = dy * [y > 0]?   one read of dy and y, one or two writes
//
// fp32, NCHW (channel = plane index) or channels-last (channel = fastest index); the operation order is PyTorch's
// (multiply, add bias, add residual, clamp: no FMA contraction), so the result is bit-identical to the unfused chain.
// Memory-bound: lanes move float4 (16 B) -- 1 KB per wave instruction, fully coalesced.  AffineChannel parameters are
// frozen in every reference configuration (ResNet.py:76-77), so there is no gradient for w and b.
#include "common.h"

// multiply, THEN add: the results must equal the unfused torch chain bit for bit (hipcc contracts to FMA by default)
#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;
constexpr int kVecPerLane = 4;  // float4 per lane and loop trip: 16 KB per workgroup trip

// NCHW: flat float4 index; the four elements of a lane belong to channel (i / plane) % C unless they straddle the end of
// a plane (only possible when H*W is not a multiple of 4, e.g. the 25x42 map of res5), in which case each element looks
// its channel up itself.  One integer division per 16 bytes: hidden under the memory traffic.
struct Chan4 {
  float w[4], b[4];
};
__device__ __forceinline__ Chan4 channels_of(long long e0, long long plane, int channels, const float* __restrict__ w,
                                             const float* __restrict__ b) {
  Chan4 r;
  const long long q = e0 / plane;
  const long long rem = e0 - q * plane;
  const int c = (int)(q % channels);
  if (rem + 3 < plane) {
    const float wc = w[c], bc = b != nullptr ? b[c] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      r.w[j] = wc;
      r.b[j] = bc;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int cj = (int)(((e0 + j) / plane) % channels);
      r.w[j] = w[cj];
      r.b[j] = b != nullptr ? b[cj] : 0.f;
    }
  }
  return r;
}

template <bool kRelu, bool kRes>
__global__ void __launch_bounds__(kThreads)
affine_fwd_nchw(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                const float* __restrict__ r, float* __restrict__ y, int channels, long long plane, long long total) {
  const long long vecs = total >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* r4 = reinterpret_cast<const float4*>(r);
  float4* y4 = reinterpret_cast<float4*>(y);
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < vecs; i += (long long)gridDim.x * kThreads) {
    const Chan4 ch = channels_of(i << 2, plane, channels, w, b);
    float4 v = x4[i];
    v.x = v.x * ch.w[0] + ch.b[0];
    v.y = v.y * ch.w[1] + ch.b[1];
    v.z = v.z * ch.w[2] + ch.b[2];
    v.w = v.w * ch.w[3] + ch.b[3];
    if (kRes) {
      const float4 q = r4[i];
      v.x += q.x;
      v.y += q.y;
      v.z += q.z;
      v.w += q.w;
    }
    if (kRelu) {
      v.x = v.x <= 0.f ? 0.f : v.x;
      v.y = v.y <= 0.f ? 0.f : v.y;
      v.z = v.z <= 0.f ? 0.f : v.z;
      v.w = v.w <= 0.f ? 0.f : v.w;
    }
    y4[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (total & 3)) {   // the last one to three elements of the tensor
    const long long e = (vecs << 2) + threadIdx.x;
    const int c = (int)((e / plane) % channels);
    float v = x[e] * w[c] + b[c];
    if (kRes) v += r[e];
    if (kRelu) v = v <= 0.f ? 0.f : v;
    y[e] = v;
  }
}

template <bool kRelu, bool kRes>
__global__ void __launch_bounds__(kThreads)
affine_bwd_nchw(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ w,
                float* __restrict__ dx, float* __restrict__ dr, int channels, long long plane, long long total) {
  const long long vecs = total >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(dy);
  const float4* y4 = reinterpret_cast<const float4*>(y);
  float4* dx4 = reinterpret_cast<float4*>(dx);
  float4* dr4 = reinterpret_cast<float4*>(dr);
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < vecs; i += (long long)gridDim.x * kThreads) {
    const Chan4 ch = channels_of(i << 2, plane, channels, w, nullptr);
    float4 g = g4[i];
    if (kRelu) {
      const float4 o = y4[i];
      g.x = o.x <= 0.f ? 0.f : g.x;   // threshold_backward: zero where y <= 0 (NaN passes)
      g.y = o.y <= 0.f ? 0.f : g.y;
      g.z = o.z <= 0.f ? 0.f : g.z;
      g.w = o.w <= 0.f ? 0.f : g.w;
    }
    if (kRes) dr4[i] = g;
    g.x *= ch.w[0];
    g.y *= ch.w[1];
    g.z *= ch.w[2];
    g.w *= ch.w[3];
    dx4[i] = g;
  }
  if (blockIdx.x == 0 && threadIdx.x < (total & 3)) {
    const long long e = (vecs << 2) + threadIdx.x;
    float g = dy[e];
    if (kRelu) g = y[e] <= 0.f ? 0.f : g;
    if (kRes) dr[e] = g;
    dx[e] = g * w[(int)((e / plane) % channels)];
  }
}

// channels-last: element i belongs to channel i % C; C % 4 == 0 is required, a lane's float4 = 4 consecutive channels
template <bool kRelu, bool kRes>
__global__ void __launch_bounds__(kThreads)
affine_fwd_nhwc(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                const float* __restrict__ r, float* __restrict__ y, int channels, long long total) {
  const long long vecs = total >> 2;
  const int cvec = channels >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* r4 = reinterpret_cast<const float4*>(r);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float4* y4 = reinterpret_cast<float4*>(y);
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < vecs; i += (long long)gridDim.x * kThreads) {
    const int cv = (int)(i % cvec);
    const float4 wc = w4[cv], bc = b4[cv];
    float4 v = x4[i];
    v.x = v.x * wc.x + bc.x;
    v.y = v.y * wc.y + bc.y;
    v.z = v.z * wc.z + bc.z;
    v.w = v.w * wc.w + bc.w;
    if (kRes) {
      const float4 q = r4[i];
      v.x += q.x;
      v.y += q.y;
      v.z += q.z;
      v.w += q.w;
    }
    if (kRelu) {
      v.x = v.x <= 0.f ? 0.f : v.x;
      v.y = v.y <= 0.f ? 0.f : v.y;
      v.z = v.z <= 0.f ? 0.f : v.z;
      v.w = v.w <= 0.f ? 0.f : v.w;
    }
    y4[i] = v;
  }
}

template <bool kRelu, bool kRes>
__global__ void __launch_bounds__(kThreads)
affine_bwd_nhwc(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ w,
                float* __restrict__ dx, float* __restrict__ dr, int channels, long long total) {
  const long long vecs = total >> 2;
  const int cvec = channels >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(dy);
  const float4* y4 = reinterpret_cast<const float4*>(y);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float4* dx4 = reinterpret_cast<float4*>(dx);
  float4* dr4 = reinterpret_cast<float4*>(dr);
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < vecs; i += (long long)gridDim.x * kThreads) {
    const float4 wc = w4[(int)(i % cvec)];
    float4 g = g4[i];
    if (kRelu) {
      const float4 o = y4[i];
      g.x = o.x <= 0.f ? 0.f : g.x;
      g.y = o.y <= 0.f ? 0.f : g.y;
      g.z = o.z <= 0.f ? 0.f : g.z;
      g.w = o.w <= 0.f ? 0.f : g.w;
    }
    if (kRes) dr4[i] = g;
    g.x *= wc.x;
    g.y *= wc.y;
    g.z *= wc.z;
    g.w *= wc.w;
    dx4[i] = g;
  }
}

int check(const void* a, const void* w, int batch, int channels, int height, int width, int layout, const char* who) {
  MI_REQUIRE(batch >= 0 && channels > 0 && height >= 0 && width >= 0, "%s: bad size", who);
  MI_REQUIRE(layout == MI_LAYOUT_NCHW || layout == MI_LAYOUT_NHWC, "%s: unknown layout %d", who, layout);
  MI_REQUIRE(layout == MI_LAYOUT_NCHW || channels % 4 == 0, "%s: channels-last needs channels %% 4 == 0", who);
  const long long total = (long long)batch * channels * height * width;
  if (total > 0) MI_REQUIRE(a != nullptr && w != nullptr, "%s: null pointer", who);
  MI_REQUIRE((reinterpret_cast<uintptr_t>(a) & 15) == 0, "%s: tensors must be 16-byte aligned", who);
  return MI_OK;
}

}  // namespace

#define MI_AFFINE_DISPATCH(KERNEL, GRID, ...)                                                       \
  do {                                                                                              \
    if (relu && residual_used)                                                                      \
      KERNEL<true, true><<<GRID, kThreads, 0, s>>>(__VA_ARGS__);                                    \
    else if (relu)                                                                                  \
      KERNEL<true, false><<<GRID, kThreads, 0, s>>>(__VA_ARGS__);                                   \
    else if (residual_used)                                                                         \
      KERNEL<false, true><<<GRID, kThreads, 0, s>>>(__VA_ARGS__);                                   \
    else                                                                                            \
      KERNEL<false, false><<<GRID, kThreads, 0, s>>>(__VA_ARGS__);                                  \
  } while (0)

extern "C" int mi_affine_channel_forward(const float* x, const float* weight, const float* bias, const float* residual,
                                         float* y, int batch, int channels, int height, int width, int relu,
                                         int layout, mi_stream_t stream) {
  mi::begin_call();
  int rc = check(x, weight, batch, channels, height, width, layout, "affine_channel_forward");
  if (rc != MI_OK) return rc;
  const long long plane = (long long)height * width, total = plane * batch * channels;
  if (total == 0) return MI_OK;
  MI_REQUIRE(bias != nullptr && y != nullptr, "affine_channel_forward: null pointer");
  hipStream_t s = mi::as_stream(stream);
  const bool residual_used = residual != nullptr;
  const long long want = ((total >> 2) + kThreads * kVecPerLane - 1) / (kThreads * kVecPerLane);
  const int grid = (int)(want > 256 * 32 ? 256 * 32 : (want < 1 ? 1 : want));   // 256 CUs x 32: grid-stride beyond
  if (layout == MI_LAYOUT_NCHW)
    MI_AFFINE_DISPATCH(affine_fwd_nchw, grid, x, weight, bias, residual, y, channels, plane, total);
  else
    MI_AFFINE_DISPATCH(affine_fwd_nhwc, grid, x, weight, bias, residual, y, channels, total);
  return mi::check_launch("affine_channel_forward");
}

extern "C" int mi_affine_channel_backward(const float* grad_y, const float* y, const float* weight, float* grad_x,
                                          float* grad_residual, int batch, int channels, int height, int width,
                                          int relu, int layout, mi_stream_t stream) {
  mi::begin_call();
  int rc = check(grad_y, weight, batch, channels, height, width, layout, "affine_channel_backward");
  if (rc != MI_OK) return rc;
  const long long plane = (long long)height * width, total = plane * batch * channels;
  if (total == 0) return MI_OK;
  MI_REQUIRE(grad_x != nullptr && (!relu || y != nullptr), "affine_channel_backward: null pointer");
  hipStream_t s = mi::as_stream(stream);
  const bool residual_used = grad_residual != nullptr;
  const long long want = ((total >> 2) + kThreads * kVecPerLane - 1) / (kThreads * kVecPerLane);
  const int grid = (int)(want > 256 * 32 ? 256 * 32 : (want < 1 ? 1 : want));
  if (layout == MI_LAYOUT_NCHW)
    MI_AFFINE_DISPATCH(affine_bwd_nchw, grid, grad_y, y, weight, grad_x, grad_residual, channels, plane, total);
  else
    MI_AFFINE_DISPATCH(affine_bwd_nhwc, grid, grad_y, y, weight, grad_x, grad_residual, channels, total);
  return mi::check_launch("affine_channel_backward");
}
