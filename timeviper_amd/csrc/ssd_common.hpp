// Device helpers shared by the SSD scan kernels (ssd_slice.hip, ssd_head.hip, ssd_correct.hip): MFMA and
// transposing-LDS-read wrappers, LDS-DMA as inline asm, DPP wave scan, fast softplus.
#pragma once
#include "common.hpp"

namespace ssdk {

typedef __attribute__((address_space(3))) bf16x4 lds_v4;
typedef __attribute__((address_space(3))) void lptr_t;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 tr4(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)p);
}
__device__ __forceinline__ bf16x8 cat4(bf16x4 lo, bf16x4 hi) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
  return r;
}
__device__ __forceinline__ bf16x8 ld8(const unsigned char* p) { return *(const bf16x8*)p; }

// LDS-DMA as inline asm: hipcc then does not know an LDS write is in flight, so it inserts
// no vmcnt(0) in front of the fragment reads (it does for the builtin + ds_read_tr); every
// wait on these copies is the hand-counted s_waitcnt vmcnt(N) + barrier below.  M0 (the LDS
// destination base) is saved and restored inside the statement.
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t*)p);
}
// 16 bytes per active lane: LDS[lds_dst + 16*lane] = *(sbase + voff)
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// Four 1 KiB pieces with ONE M0 set-up: piece j copies *(sbase + vj + 1024 j) to LDS[lds_dst + 1024 j + 16*lane].
// The instruction offset moves the global and the LDS address alike, so the caller passes vj = (source offset of
// piece j) - 1024 j (never negative for rows of >= 256 bytes taken in order).
__device__ __forceinline__ void glds16x4(const void* sbase, unsigned v0, unsigned v1, unsigned v2, unsigned v3,
                                         unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %5\n\t"
               "global_load_lds_dwordx4 %2, %5 offset:1024\n\t"
               "global_load_lds_dwordx4 %3, %5 offset:2048\n\t"
               "global_load_lds_dwordx4 %4, %5 offset:3072\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_dst) : "memory");
}
// one dword per lane: LDS[lds_dst + 4*lane]
__device__ __forceinline__ void glds4(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// A 16-byte global load the compiler does not track: beside the hand-counted LDS-DMA copies hipcc waits
// vmcnt(0) at the first use of any ordinary load result — inside the tile loop that drained the copy
// queue at the top of every tile (the Q fragments are such results).  The caller's own counted wait covers
// these loads (they are older than the copies it leaves in flight); `settle` pins the uses behind it.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 gload16_async(const void* p) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ void settle(u32x4& r) { asm volatile("" : "+v"(r)); }

__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}

// inclusive prefix sum over the 64 lanes with DPP row shifts / broadcasts (no LDS)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_shift(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float wave_incl_scan_dpp(float v) {
  v += dpp_shift<0x111, 0xf>(v);   // row_shr:1
  v += dpp_shift<0x112, 0xf>(v);   // row_shr:2
  v += dpp_shift<0x114, 0xf>(v);   // row_shr:4
  v += dpp_shift<0x118, 0xf>(v);   // row_shr:8
  v += dpp_shift<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
  v += dpp_shift<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
  return v;
}
// softplus on the hardware exp/log units; the small-argument branch keeps full relative
// accuracy where 1+e^x would round (log1p(e) = e - e^2/2 + O(e^3))
__device__ __forceinline__ float softplus_fast(float x) {
  const float e = __expf(x);
  const float sp = e < 1e-3f ? e - 0.5f * e * e : __logf(1.f + e);
  return x > 20.f ? x : sp;
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

}  // namespace ssdk
