/*
 * cuda_on_cpu.h -- TEST INFRASTRUCTURE.  A minimal host-side stand-in for the CUDA language
 * surface the reference's four .cu files use, so that oracle/build_ref.py can compile those
 * files FROM WHERE THEY LIE under /root/reference with g++ into oracle/_ref/*.so and run the
 * reference's own kernel bodies on the CPU.  It contains no reference code.
 *
 * Execution model: a launch `k<<<grid, block, shmem, stream>>>(args...)` (rewritten by
 * build_ref.py into cuda_on_cpu::launch(k, cuda_on_cpu::cfg(grid, block, ...), args...)) runs every
 * (blockIdx, threadIdx) sequentially.  `__shared__` becomes `static`, `__syncthreads()` a no-op;
 * a kernel that needs a barrier (only nms_kernel does: load tile, barrier, use tile) is compiled
 * with -DCUDA_ON_CPU_PASSES=2 so each block's threads run twice -- the second pass sees the
 * fully populated tile and overwrites the first pass's results (the kernel's writes are
 * idempotent).  atomicAdd is a plain add (single host thread).
 */
#ifndef ORACLE_CUDA_ON_CPU_H_
#define ORACLE_CUDA_ON_CPU_H_

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifndef CUDA_ON_CPU_PASSES
#define CUDA_ON_CPU_PASSES 1
#endif

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

static dim3 blockIdx, threadIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline

typedef int cudaError_t;
typedef void* cudaStream_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };

static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline const char* cudaGetErrorString(cudaError_t) { return "cuda_on_cpu"; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
template <typename T>
static inline cudaError_t cudaMalloc(T** p, size_t bytes) {
  *p = static_cast<T*>(std::malloc(bytes ? bytes : 1));
  return cudaSuccess;
}
static inline cudaError_t cudaFree(void* p) {
  std::free(p);
  return cudaSuccess;
}
static inline cudaError_t cudaMemcpy(void* dst, const void* src, size_t bytes, cudaMemcpyKind) {
  std::memcpy(dst, src, bytes);
  return cudaSuccess;
}
static inline void __syncthreads() {}
static inline float atomicAdd(float* address, float val) {
  float old = *address;
  *address = old + val;
  return old;
}

using std::max;
using std::min;

namespace cuda_on_cpu {
struct launch_cfg {
  dim3 grid, block;
};
template <typename... Rest>
static inline launch_cfg cfg(dim3 grid, dim3 block, Rest...) {
  return launch_cfg{grid, block};
}
template <typename Kernel, typename... Args>
static inline void launch(Kernel kernel, launch_cfg c, Args... args) {
  gridDim = c.grid;
  blockDim = c.block;
  for (unsigned bz = 0; bz < c.grid.z; bz++)
    for (unsigned by = 0; by < c.grid.y; by++)
      for (unsigned bx = 0; bx < c.grid.x; bx++) {
        blockIdx = dim3(bx, by, bz);
        for (int pass = 0; pass < CUDA_ON_CPU_PASSES; pass++)
          for (unsigned tz = 0; tz < c.block.z; tz++)
            for (unsigned ty = 0; ty < c.block.y; ty++)
              for (unsigned tx = 0; tx < c.block.x; tx++) {
                threadIdx = dim3(tx, ty, tz);
                kernel(args...);
              }
      }
}
}  // namespace cuda_on_cpu

#endif  // ORACLE_CUDA_ON_CPU_H_
