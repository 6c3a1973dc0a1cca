// Micro-benchmark (tuning aid): how fast the chip hands out workgroups.  An (almost) empty kernel -- one dword store from
// lane 0 when `touch` is set, nothing otherwise -- launched as G workgroups of T lanes with L bytes of LDS each.
// build: hipcc --offload-arch=gfx950 -O3 -o launch_rate launch_rate.hip ; run: ./launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(float* __restrict__ sink, int touch, int spin) {
  extern __shared__ float smem[];
  if (spin > 0) {
    const long long t0 = clock64();
    while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(1);
  }
  if (touch && threadIdx.x == 0) sink[blockIdx.x + blockIdx.y * gridDim.x] = smem[threadIdx.x & 1] + 1.f;
}

int main() {
  float* sink;
  hipMalloc(&sink, 1 << 22);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int cfg[][4] = {{16384, 64, 10752, 0}, {16384, 64, 1024, 0},  {16384, 64, 10752, 1}, {4096, 256, 43008, 0},
                        {4096, 256, 43008, 1}, {8192, 128, 21504, 0}, {32768, 64, 10752, 0}, {16384, 64, 10752, 2},
                        {16384, 64, 10752, 3}};
  for (auto& c : cfg) {
    const int spin = c[3] == 2 ? 4000 : c[3] == 3 ? 10000 : 0, touch = c[3] == 1;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int i = 0; i < 20; i++) k<<<dim3(c[0] / 4, 4), c[1], c[2]>>>(sink, touch, spin);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 200; i++) k<<<dim3(c[0] / 4, 4), c[1], c[2]>>>(sink, touch, spin);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("G %6d x %3d lanes, LDS %5d B, %s: %.2f us per launch (%.1f ns per workgroup)\n", c[0], c[1], c[2],
           touch ? "one store" : spin ? (spin == 4000 ? "spin 4000 clk (~40 us of 100 MHz ticks?)" : "spin 10000 clk") : "empty", ms * 5.f,
           ms * 5000.f / c[0]);
  }
  // LDS granularity: residency of 1-wave workgroups that spin ~4.8 us, by LDS request (time ~ 1 / resident waves per CU)
  for (int lds = 6144; lds <= 12288; lds += 256) {
    for (int i = 0; i < 5; i++) k<<<dim3(4096, 4), 64, lds>>>(sink, 0, 10000);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 50; i++) k<<<dim3(4096, 4), 64, lds>>>(sink, 0, 10000);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("LDS %5d B: %.2f us per launch -> ~%.1f resident waves per CU\n", lds, ms * 20.f, 16384.0 * 4.76 / 256.0 / (ms * 20.f - 2.5));
  }
  return 0;
}
