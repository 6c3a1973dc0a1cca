// lds_dma.h -- gfx950 helpers shared by the RoIAlign kernels: wave-uniform values, buffer descriptors and the
// global -> LDS DMA (`buffer_load_dword ... lds`).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mi {
namespace {

using lds_cfloat_t = __attribute__((address_space(3))) const float*;
using const_int_ptr = const __attribute__((address_space(4))) int*;

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

using srd_t = __attribute__((ext_vector_type(4))) unsigned;
// raw buffer descriptor (stride 0) over [base, base + num_bytes): a lane whose offset lies outside reads 0
__device__ __forceinline__ srd_t make_srd(const void* base, unsigned num_bytes) {
  const uintptr_t b = reinterpret_cast<uintptr_t>(base);
  srd_t r;
  r.x = (unsigned)uniform((int)(b & 0xffffffffu));
  r.y = (unsigned)uniform((int)(b >> 32)) & 0xffffu;  // stride 0
  r.z = (unsigned)uniform((int)num_bytes);
  r.w = 0x00020000u;
  return r;
}

// One LDS-DMA piece: LDS[lds_base + lane * 4] = buffer[voff + soff] for the active lanes (buffer_load_dword ... lds).
// Issued through inline asm ON PURPOSE: hipcc orders every later LDS read of the kernel behind an LDS-DMA it can see
// (vmcnt(0) before the first ds_read, it cannot tell which part of the LDS image a piece lands in), which would
// serialise independent work with the landing of the image.  The kernels wait for their DMA explicitly
// (s_waitcnt vmcnt(0) in front of the barrier that publishes an image); nothing reads an image before.
__device__ __forceinline__ void dma_dword(srd_t srd, unsigned lds_base, unsigned voff, unsigned soff) {
  lds_base = (unsigned)uniform((int)lds_base);  // the "s" constraint alone does not move a VGPR-resident value
  soff = (unsigned)uniform((int)soff);
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, %3 offen lds"
               :
               : "s"(lds_base), "v"(voff), "s"(srd), "s"(soff)
               : "memory");
}

// tools/build_defines.sh MI_DMA_AUX_ID=1|2|3: cache-policy bits of the 16-byte window pieces (nt / sc1 / sc0 sc1); measured
// on the records-free forward, round 6: see profiles/r06_slab_forward.txt
#if MI_DMA_AUX_ID == 1
#define MI_DMA_AUX " nt"
#elif MI_DMA_AUX_ID == 2
#define MI_DMA_AUX " sc1"
#elif MI_DMA_AUX_ID == 3
#define MI_DMA_AUX " sc0 sc1"
#else
#define MI_DMA_AUX ""
#endif
// The 16-byte form: LDS[lds_base + lane * 16 .. + 15] = buffer[voff + soff .. + 15].  The LDS side needs no more than
// dword alignment (LDS-DMA writes are not subject to the alignment replay of ds_write_b128).
__device__ __forceinline__ void dma_dwordx4(srd_t srd, unsigned lds_base, unsigned voff, unsigned soff) {
  lds_base = (unsigned)uniform((int)lds_base);
  soff = (unsigned)uniform((int)soff);
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen" MI_DMA_AUX " lds"
               :
               : "s"(lds_base), "v"(voff), "s"(srd), "s"(soff)
               : "memory");
}

__device__ __forceinline__ unsigned lds_addr_uniform(const void* p) {
  return (unsigned)uniform((int)(unsigned)(uintptr_t)(lds_cfloat_t)p);
}

}  // namespace
}  // namespace mi
