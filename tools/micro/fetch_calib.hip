// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access width the NCHW RoIAlign kernels use (tuning aid).
// Each workgroup streams its share of a buffer exactly once:
//   mode 0: buffer_load_dword ... lds   (4 B per lane, 256 B per wave instruction -- the LDS-DMA pattern)
//   mode 1: global_load_dwordx4 to VGPRs (16 B per lane -- the width MI355X_MICROARCH.md's "x2" correction is stated for)
// The buffer (512 MiB) exceeds the 256 MiB Infinity Cache, every byte is read once: FETCH_SIZE (KB) * 1024 / bytes is
// the factor to apply.  build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; run: ./fetch_calib <mode>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using srd_t = __attribute__((ext_vector_type(4))) unsigned;

__global__ void __launch_bounds__(256) stream_dma4(const float* __restrict__ src, size_t words_per_wg, float* sink) {
  __shared__ float buf[4 * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* base = src + (size_t)blockIdx.x * words_per_wg;
  const uintptr_t b = (uintptr_t)base;
  srd_t srd;
  srd.x = (unsigned)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
  srd.y = (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) & 0xffffu;
  srd.z = (unsigned)(words_per_wg * 4);
  srd.w = 0x00020000u;
  const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(buf + wave * 64);
  const unsigned lds_u = (unsigned)__builtin_amdgcn_readfirstlane((int)lds);
  for (size_t w = (size_t)wave * 64; w < words_per_wg; w += 256) {
    const unsigned voff = (unsigned)(w + lane) * 4u;
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, 0 offen lds" : : "s"(lds_u), "v"(voff), "s"(srd) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && buf[0] == 123.456f) sink[0] = buf[1];
}

__global__ void __launch_bounds__(256) stream_x4(const float4* __restrict__ src, size_t vec_per_wg, float* sink) {
  const float4* base = src + (size_t)blockIdx.x * vec_per_wg;
  float acc = 0.f;
  for (size_t i = threadIdx.x; i < vec_per_wg; i += 256) {
    const float4 v = base[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const size_t bytes = 512ull << 20, words = bytes / 4;
  float *src, *sink;
  hipMalloc(&src, bytes);
  hipMalloc(&sink, 64);
  hipMemset(src, 0, bytes);
  const int wgs = 4096;
  for (int rep = 0; rep < 3; rep++) {
    if (mode == 0)
      stream_dma4<<<wgs, 256>>>(src, words / wgs, sink);
    else
      stream_x4<<<wgs, 256>>>((const float4*)src, words / 4 / wgs, sink);
  }
  hipDeviceSynchronize();
  printf("mode %d: %zu bytes per launch, 3 launches\n", mode, bytes);
  return 0;
}
