// Micro-benchmark (tuning aid): throughput of LDS fp32 atomics on gfx950 -- what a tile kernel that accumulates in LDS can
// expect.  Every wave issues N ds_add_f32 (no return); 4 workgroups of 256 lanes per CU.
//   mode 0: 64 lanes active, 64 distinct banks (lane -> its own dword)
//   mode 1: 8 lanes active (exec-masked), distinct banks
//   mode 2: 64 lanes active, random addresses in a 32 KB region (bank conflicts as they fall)
//   mode 3: mode 2 with 32 lanes active
//   mode 4: plain ds_read + add + ds_write of the same addresses as mode 2 (not atomic)
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_bench lds_atomic_bench.hip ; run: ./lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(float* sink, int n) {
  __shared__ float acc[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) acc[i] = 0.f;
  __syncthreads();
  unsigned s = tid * 2654435761u + blockIdx.x;
  const bool on = MODE == 1 ? (lane & 7) == 0 : (MODE == 3 ? (lane & 1) == 0 : true);
  for (int i = 0; i < n; i++) {
    s = s * 1664525u + 1013904223u;
    const int idx = (MODE <= 1) ? tid + ((i & 15) << 8) : (int)(s >> 19);
    if (MODE == 4) {
      acc[idx] = acc[idx] + 1.0f;
    } else if (on) {
      __hip_atomic_fetch_add(&acc[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  float t = 0.f;
  for (int i = tid; i < 8192; i += 256) t += acc[i];
  if (t == -1.f) sink[0] = t;
}

template <int MODE>
void run(const char* what, int lanes) {
  float* sink;
  hipMalloc(&sink, 4);
  const int n = 4096, grid = 256 * 4;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<MODE><<<grid, 256>>>(sink, n);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 5; r++) k<MODE><<<grid, 256>>>(sink, n);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / 5;
  const double per_cu_instr = (double)n * 16;  // 16 waves per CU
  printf("%-58s %8.1f us  %6.1f ns per wave-instruction per CU  %5.2f lane-atomics per ns per CU\n", what, us, us * 1e3 / per_cu_instr,
         per_cu_instr * lanes / (us * 1e3));
}

int main() {
  run<0>("0: 64 lanes, own bank", 64);
  run<1>("1: 8 lanes (masked), own bank", 8);
  run<2>("2: 64 lanes, random addresses", 64);
  run<3>("3: 32 lanes, random addresses", 32);
  run<4>("4: read + add + write (not atomic), random addresses", 64);
  return 0;
}
