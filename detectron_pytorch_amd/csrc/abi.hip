// abi.hip -- version / error-string entry points of the C-ABI (include/mi_detectron_ops.h).
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace mi {
namespace {
thread_local char g_error[512] = {0};
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
void clear_error() { g_error[0] = 0; }

namespace {
int env_int(const char* name, int fallback) {
  const char* v = std::getenv(name);
  return v != nullptr ? std::atoi(v) : fallback;
}
Tuning read_tuning() {
  Tuning t = {};
  const char* impl = std::getenv("MI_ROI_ALIGN_IMPL");
  t.force_direct = impl != nullptr && std::strcmp(impl, "direct") == 0;
  t.use_pipe = impl != nullptr && std::strcmp(impl, "pipe") == 0;
  t.no_ws = std::getenv("MI_ROI_ALIGN_NO_WS") != nullptr;
  t.cap_px = env_int("MI_ROI_ALIGN_CAP", 336);
  const int th = env_int("MI_ROI_ALIGN_BWD_TH", 16);
  t.bwd_tile_rows = (th == 8 || th == 32) ? th : 16;
  const int sl = env_int("MI_ROI_ALIGN_BWD_SLICE", 32);
  t.bwd_slice = 0;  // a power of two in [2, 256], or 0 (no plan)
  for (int p2 = 2; p2 <= 256 && p2 <= sl; p2 *= 2) t.bwd_slice = p2;
  t.nhwc_vec = env_int("MI_ROI_ALIGN_NHWC_V", 0);
  t.nhwc_pb = env_int("MI_ROI_ALIGN_NHWC_PB", 0);
  const int om = env_int("MI_ROI_ALIGN_NHWC_ORDER_MUL", 1);
  t.nhwc_order_mul = om > 0 ? om : 1;
  t.nhwc_zigzag = env_int("MI_ROI_ALIGN_NHWC_ZIGZAG", 1);
  t.ablate = MI_ABLATE(env_int("MI_ROI_ALIGN_ABLATE", 0));
  return t;
}
}  // namespace

namespace {
Tuning g_tuning;
std::once_flag g_tuning_once;
}  // namespace

const Tuning& tuning() {
  std::call_once(g_tuning_once, [] { g_tuning = read_tuning(); });
  return g_tuning;
}
void reload_tuning() {
  (void)tuning();
  g_tuning = read_tuning();
}
}  // namespace mi

// Measurement aid, not part of include/mi_detectron_ops.h: the streaming ceiling of the box the roofline fractions are also
// quoted against (bench.py: roofline.copy_ceiling) -- a grid-stride copy with 16 bytes per lane and four independent
// loads in flight per lane, the form /opt/skills/guides/MI355X_MICROARCH.md measures at ~6.3 TB/s (read + write bytes).
namespace {
__global__ void __launch_bounds__(256) copy_float4(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a;
    dst[i + stride] = b;
    dst[i + 2 * stride] = c;
    dst[i + 3 * stride] = d;
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}
}  // namespace
extern "C" int mi_dbg_copy_float4(const void* src, void* dst, size_t bytes, mi_stream_t stream) {
  mi::begin_call();
  if (bytes == 0) return MI_OK;
  MI_REQUIRE(src != nullptr && dst != nullptr && bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(dst) & 15) == 0,
             "mi_dbg_copy_float4: 16-byte aligned buffers of a multiple of 16 bytes");
  const size_t n4 = bytes / 16;
  const size_t want = (n4 + 256 * 4 - 1) / (256 * 4);
  copy_float4<<<(unsigned)(want < 256 * 32 ? (want ? want : 1) : 256 * 32), 256, 0, mi::as_stream(stream)>>>(
      static_cast<const float4*>(src), static_cast<float4*>(dst), n4);
  return mi::check_launch("copy_float4");
}

// Test / tuning aid, not part of include/mi_detectron_ops.h: re-read the MI_ROI_ALIGN_* environment into the tuning
// struct.  The only writer of that struct after its one-time initialisation; call it with no launch in flight on any
// thread (the tests do, between cases).
extern "C" void mi_dbg_reload_tuning(void) { mi::reload_tuning(); }

extern "C" int mi_abi_version(void) { return MI_ABI_VERSION; }
extern "C" const char* mi_last_error(void) { return mi::g_error; }
