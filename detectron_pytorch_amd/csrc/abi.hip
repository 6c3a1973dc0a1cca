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
  t.fwd_split = env_int("MI_ROI_ALIGN_FWD_SPLIT", 0);
  t.fwd_full_wait = env_int("MI_ROI_ALIGN_FWD_FULL_WAIT", 0);
  t.slab = env_int("MI_ROI_ALIGN_SLAB", 1);
  t.ablate = MI_ABLATE(env_int("MI_ROI_ALIGN_ABLATE", 0));
  t.copy_variant = env_int("MI_COPY_VARIANT", 411);
  return t;
}
}  // namespace

namespace {
Tuning g_tuning;
std::once_flag g_tuning_once;
}  // namespace

// compute units of the current device (resident grids, the copy-ceiling kernel); asked once
int compute_units() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    (void)hipGetLastError();
    return cus;
  }();
  return n;
}

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
// quoted against (bench.py: roofline.copy_ceiling) -- a grid-stride copy with 16 bytes per lane, the form
// /opt/skills/guides/MI355X_MICROARCH.md measures at ~6.3 TB/s (read + write bytes).
namespace {
typedef float v4f_t __attribute__((ext_vector_type(4)));
template <int kU, bool kNT>
__global__ void __launch_bounds__(256) copy_float4(const v4f_t* __restrict__ src, v4f_t* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (kU - 1) * stride < n4; i += kU * stride) {
    v4f_t v[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) v[u] = kNT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      if (kNT)
        __builtin_nontemporal_store(v[u], dst + i + u * stride);
      else
        dst[i + u * stride] = v[u];
    }
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
  // MI_COPY_VARIANT = blocks_per_cu * 100 + unroll * 10 + nontemporal (tools/copy_sweep.py; read with the other tuning
  // variables, i.e. once, or again through mi_dbg_reload_tuning).  Default: 4 workgroups per CU, one 16-byte load in flight
  // per lane and loop trip, non-temporal loads and stores -- 6.48 TB/s on a 256 MiB buffer, the best of the sweep (2..64
  // workgroups per CU x unroll 1 / 4 / 8 x nt: 4.1-6.5 TB/s; torch's copy_: 5.40)
  const int v = mi::tuning().copy_variant;
  const int per_cu = v / 100 > 0 ? v / 100 : 8, unroll = (v / 10) % 10, nt = v % 10;
  size_t blocks = (size_t)mi::compute_units() * per_cu;
  const size_t want = (n4 + 255) / 256;
  if (blocks > want) blocks = want ? want : 1;
  const v4f_t* s4 = static_cast<const v4f_t*>(src);
  v4f_t* d4 = static_cast<v4f_t*>(dst);
  hipStream_t st = mi::as_stream(stream);
  if (unroll >= 8 && nt)
    copy_float4<8, true><<<(unsigned)blocks, 256, 0, st>>>(s4, d4, n4);
  else if (unroll >= 8)
    copy_float4<8, false><<<(unsigned)blocks, 256, 0, st>>>(s4, d4, n4);
  else if (unroll >= 4 && nt)
    copy_float4<4, true><<<(unsigned)blocks, 256, 0, st>>>(s4, d4, n4);
  else if (unroll >= 4)
    copy_float4<4, false><<<(unsigned)blocks, 256, 0, st>>>(s4, d4, n4);
  else if (nt)
    copy_float4<1, true><<<(unsigned)blocks, 256, 0, st>>>(s4, d4, n4);
  else
    copy_float4<1, false><<<(unsigned)blocks, 256, 0, st>>>(s4, d4, n4);
  return mi::check_launch("copy_float4");
}

// Test / tuning aid, not part of include/mi_detectron_ops.h: re-read the MI_ROI_ALIGN_* environment into the tuning
// struct.  The only writer of that struct after its one-time initialisation; call it with no launch in flight on any
// thread (the tests do, between cases).
extern "C" void mi_dbg_reload_tuning(void) { mi::reload_tuning(); }

extern "C" int mi_abi_version(void) { return MI_ABI_VERSION; }
extern "C" const char* mi_last_error(void) { return mi::g_error; }
