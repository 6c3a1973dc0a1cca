// Micro-benchmark (tuning aid): cost of moving RoI-sized windows of a 256x200x336 fp32 map into LDS on gfx950.
//   mode 0: LDS-DMA dword, lanes flattened over (row, col) of the window (64 px per instruction)
//   mode 1: LDS-DMA dwordx4, lanes flattened over (row, 4-px group)      (256 px per instruction)
//   mode 2: plain buffer_load_dword to VGPR + ds_write_b32 (same addressing as mode 0)
//   mode 3: LDS-DMA dword, one 32-lane row segment per half-wave (2 rows per instruction)
// build: hipcc --offload-arch=gfx950 -O3 -o dma_bench dma_bench.hip ; run: ./dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int H = 200, W = 336, C = 256;
using lds_ptr_t = __attribute__((address_space(3))) void*;

struct Win { int x0, y0, ww, nr; };

template <int MODE, int AUX>
__global__ void __launch_bounds__(256) k(const float* __restrict__ feat, const Win* __restrict__ wins, int nwin,
                                         float* __restrict__ sink, int items_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x % 8;
  float acc = 0.f;
  for (int it = 0; it < items_per_wg; it++) {
    const int wi = ((blockIdx.x / 8) + it * (gridDim.x / 8)) % nwin;
    const Win w = wins[wi];
    const int x0 = __builtin_amdgcn_readfirstlane(w.x0), y0 = __builtin_amdgcn_readfirstlane(w.y0);
    const int ww = __builtin_amdgcn_readfirstlane(w.ww), nr = __builtin_amdgcn_readfirstlane(w.nr);
    const float* src = feat + (size_t)(tile * 32 + wave * 8) * H * W;
    const uintptr_t b = (uintptr_t)src;
    const uintptr_t bu = ((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((float*)bu, 0, 8 * H * W * 4, 0x00020000);
    float* plane0 = smem + wave * 8 * 400;
    if (MODE == 0 || MODE == 2) {
      const int npx = nr * ww;
      const unsigned magic = (1u << 20) / (unsigned)ww + 1u;
      for (int kk = 0; kk * 64 < npx; kk++) {
        const unsigned p = kk * 64 + lane, q = (p * magic) >> 20, col = p - q * ww;
        const unsigned voff = ((y0 + q) * W + x0 + col) * 4u;
        if (p < (unsigned)npx) {
#pragma unroll
          for (int c = 0; c < 8; c++) {
            if (MODE == 0)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(plane0 + c * 400 + kk * 64), 4, voff, c * H * W * 4, 0, AUX);
            else
              plane0[c * 400 + p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, c * H * W * 4, 0));
          }
        }
      }
    } else if (MODE == 1) {
      const int xa = x0 & ~3, g4 = ((x0 + ww + 3) & ~3) - xa >> 2;  // 4-px groups per row
      const int ng = nr * g4;
      const unsigned magic = (1u << 20) / (unsigned)g4 + 1u;
      for (int kk = 0; kk * 64 < ng; kk++) {
        const unsigned p = kk * 64 + lane, q = (p * magic) >> 20, col = p - q * g4;
        const unsigned voff = ((y0 + q) * W + xa + col * 4) * 4u;
        if (p < (unsigned)ng) {
#pragma unroll
          for (int c = 0; c < 8; c++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(plane0 + c * 400 + kk * 256), 16, voff, c * H * W * 4, 0, AUX);
        }
      }
    } else {
      const int half = lane >> 5, lx = lane & 31;
      for (int y = 0; y < nr; y += 2) {
        const int yy = y + half;
        const unsigned voff = ((y0 + yy) * W + x0 + lx) * 4u;
        if (lx < ww && yy < nr) {
#pragma unroll
          for (int c = 0; c < 8; c++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(plane0 + c * 400 + y * 32), 4, voff, c * H * W * 4, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += smem[(tid * 37) % (32 * 400)];
    __syncthreads();
  }
  if (acc == 123.456f) sink[0] = acc;
}

#include <algorithm>
int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 0;  // 0 random, 1 sorted by (y band, x), 2 confined to 40 rows
  const int nwin = 512;
  std::vector<Win> w(nwin);
  srand(1);
  for (auto& e : w) {
    e.ww = 6 + rand() % 25; e.nr = 6 + rand() % 25;
    while (e.ww * e.nr > 384) { if (e.ww > e.nr) e.ww--; else e.nr--; }
    e.x0 = rand() % (W - e.ww - 4) ; e.y0 = rand() % (H - e.nr);
  }
  if (variant == 2) for (auto& e : w) e.y0 = e.y0 % 12;
  if (variant == 1) std::sort(w.begin(), w.end(), [](const Win& a, const Win& b) { return (a.y0 / 16) != (b.y0 / 16) ? a.y0 / 16 < b.y0 / 16 : a.x0 < b.x0; });
  long long px = 0; for (auto& e : w) px += (long long)e.ww * e.nr;
  float *feat, *sink; Win* dw;
  hipMalloc(&feat, (size_t)C * H * W * 4); hipMemset(feat, 0, (size_t)C * H * W * 4);
  hipMalloc(&sink, 4); hipMalloc(&dw, nwin * sizeof(Win));
  hipMemcpy(dw, w.data(), nwin * sizeof(Win), hipMemcpyHostToDevice);
  const int grid = 768, items = 6;  // 4608 (window, tile) items ~ config 2
  const double bytes = (double)px / nwin * grid * items * 32 * 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 12; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(a);
      for (int i = 0; i < 20; i++) {
        switch (mode) {
          case 0: k<0, 0><<<grid, 256, 32 * 400 * 4>>>(feat, dw, nwin, sink, items); break;
          case 1: k<0, 1><<<grid, 256, 32 * 400 * 4>>>(feat, dw, nwin, sink, items); break;
          case 2: k<0, 2><<<grid, 256, 32 * 400 * 4>>>(feat, dw, nwin, sink, items); break;
          case 3: k<0, 16><<<grid, 256, 32 * 400 * 4>>>(feat, dw, nwin, sink, items); break;
          case 4: k<0, 17><<<grid, 256, 32 * 400 * 4>>>(feat, dw, nwin, sink, items); break;
          case 5: k<0, 18><<<grid, 256, 32 * 400 * 4>>>(feat, dw, nwin, sink, items); break;
          case 6: k<1, 0><<<grid, 256, 32 * 400 * 4>>>(feat, dw, nwin, sink, items); break;
          case 7: k<1, 1><<<grid, 256, 32 * 400 * 4>>>(feat, dw, nwin, sink, items); break;
          case 8: k<1, 2><<<grid, 256, 32 * 400 * 4>>>(feat, dw, nwin, sink, items); break;
          case 9: k<1, 16><<<grid, 256, 32 * 400 * 4>>>(feat, dw, nwin, sink, items); break;
          case 10: k<1, 17><<<grid, 256, 32 * 400 * 4>>>(feat, dw, nwin, sink, items); break;
          case 11: k<2, 0><<<grid, 256, 32 * 400 * 4>>>(feat, dw, nwin, sink, items); break;
        }
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep == 2) printf("variant %d case %d (0-5: dword aux 0,1,2,16,17,18; 6-10: dwordx4 aux 0,1,2,16,17; 11: vgpr): %.2f us per launch\n", variant, mode, ms * 1000 / 20);
    }
  }
  printf("err %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
