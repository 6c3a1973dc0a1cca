// common.h -- shared host-side helpers of libmi_detectron_ops.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "mi_detectron_ops.h"

namespace mi {

constexpr int kWave = 64;  // CDNA wavefront

// Per-host-thread last-error text behind mi_last_error().
void set_error(const char* fmt, ...);
void clear_error();

inline hipStream_t as_stream(mi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Reports a failed launch as MI_ERR_LAUNCH (the reference printed and exit(-1)'d,
// roi_align_kernel.cu:135-139).
inline int check_launch(const char* what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(err));
    return MI_ERR_LAUNCH;
  }
  return MI_OK;
}

// Every launching entry point calls this first: hipGetLastError() is sticky per host thread, so an
// error left behind by an unrelated earlier runtime call (PyTorch does not clear it) would
// otherwise be misreported as a failure of our launch.
inline void begin_call() { (void)hipGetLastError(); }

inline int ceil_div(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

int compute_units();  // of the current device; asked once (abi.hip)

// Grid size of a grid-stride element-wise kernel: at most 16 workgroups per compute unit of the device.
inline int grid_for(long long total, int block) {
  const long long max_blocks = (long long)compute_units() * 16;
  long long g = (total + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return static_cast<int>(g);
}

// Tuning knobs of the RoIAlign launchers.  Read ONCE from the environment at first use (thread-safe static
// initialisation) and immutable afterwards: launchers take them by value from tuning(), nothing on the launch path calls
// getenv or writes process-wide state, so concurrent calls from several host threads (the reference's thread-per-GPU
// convention, nn/parallel/parallel_apply.py:41-59) see one consistent configuration.
//   MI_ROI_ALIGN_IMPL=direct   generic one-lane-per-output kernels only (tests of the generic path, A/B baselines)
//   MI_ROI_ALIGN_NO_WS=1       ignore the caller's workspace (no records path)
//   MI_ROI_ALIGN_CAP=192|256|336|448|640   window pixels per channel of the NCHW forward LDS image
//   MI_ROI_ALIGN_FWD_SPLIT=1|2|4           workgroups an NCHW forward item's stages are dealt to (0: by the launch's size)
//   MI_ROI_ALIGN_FWD_FULL_WAIT=1           NCHW forward: vmcnt(0) in front of every stage (the check of the partial wait)
//   MI_ROI_ALIGN_SLAB=0|1|196|256|260|292  records-free NCHW forward (roi_align_fwd_slab) off / on / on with that LDS image capacity
//   MI_ROI_ALIGN_BWD_TH=8|16|32            rows per backward tile
//   MI_ROI_ALIGN_BWD_SLICE=n   RoIs per list slice of the planned backward (32; 0: no plan, no atomics)
//   MI_ROI_ALIGN_NHWC_V / _PB / _ORDER_MUL / _ZIGZAG   channels-last forward variants
//   MI_ROI_ALIGN_ABLATE=mask   only honoured by builds with -DMI_TUNING (tools/); release kernels compile it out
struct Tuning {
  bool force_direct, no_ws;
  int cap_px, bwd_tile_rows, bwd_slice;
  int nhwc_vec, nhwc_pb, nhwc_order_mul, nhwc_zigzag, fwd_split, fwd_full_wait;
  int slab;  // MI_ROI_ALIGN_SLAB: 0 = records-free forward off, 1 = on, >= 64: on with that LDS image capacity
  int ablate;
  int copy_variant;  // MI_COPY_VARIANT of mi_dbg_copy_float4 (tools/copy_sweep.py)
};
const Tuning& tuning();
void reload_tuning();  // mi_dbg_reload_tuning() only

#ifdef MI_TUNING
#define MI_ABLATE(mask) (mask)
#else
#define MI_ABLATE(mask) 0
#endif

#define MI_REQUIRE(cond, ...)          \
  do {                                 \
    if (!(cond)) {                     \
      ::mi::set_error(__VA_ARGS__);    \
      return MI_ERR_BAD_ARGUMENT;      \
    }                                  \
  } while (0)

}  // namespace mi
