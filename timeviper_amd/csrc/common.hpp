// Shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/timeviper_hip.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define TV_WAVE 64

// 16-byte vector of T (bf16/f16: 8 lanes, f32: 4 lanes).
template <typename T> struct Vec16;
template <> struct Vec16<float> { enum { N = 4 }; typedef f32x4 type; };
template <> struct Vec16<bf16_t> { enum { N = 8 }; typedef bf16x8 type; };
template <> struct Vec16<f16_t> { enum { N = 8 }; typedef f16x8 type; };

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
// softplus with torch's default threshold (20): identical branch to
// F.softplus and to upstream's `dt < 20` guard.
__device__ __forceinline__ float softplus_f(float x) {
  return x > 20.f ? x : log1pf(__expf(x));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// The same sum without LDS: an inclusive scan of DPP row shifts / row broadcasts (6 vector instructions, no
// ds_bpermute round trips — __shfl_xor costs one per step on gfx9, ~100 cycles each behind an lgkmcnt wait) and one
// v_readlane of lane 63.  Wave-uniform result; the additions associate differently from wave_sum's butterfly.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float tv_dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = tv_dpp_add<0x111, 0xf>(v);   // row_shr:1
  v = tv_dpp_add<0x112, 0xf>(v);   // row_shr:2
  v = tv_dpp_add<0x114, 0xf>(v);   // row_shr:4
  v = tv_dpp_add<0x118, 0xf>(v);   // row_shr:8
  v = tv_dpp_add<0x142, 0xa>(v);   // row_bcast:15 -> rows 1, 3
  v = tv_dpp_add<0x143, 0xc>(v);   // row_bcast:31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ float wave_incl_scan(float v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// error plumbing (capi.cpp)
void tv_set_error(const char* fmt, ...);
#define TV_CHECK_ARG(cond, ...)                         \
  do {                                                  \
    if (!(cond)) {                                      \
      tv_set_error(__VA_ARGS__);                        \
      return TV_ERR_BAD_ARG;                            \
    }                                                   \
  } while (0)
#define TV_UNSUPPORTED(...)                             \
  do {                                                  \
    tv_set_error(__VA_ARGS__);                          \
    return TV_ERR_UNSUPPORTED;                          \
  } while (0)
#define TV_LAUNCH_CHECK()                                               \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) {                                            \
      tv_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,       \
                   hipGetErrorString(e__));                             \
      return TV_ERR_LAUNCH;                                             \
    }                                                                   \
    return TV_OK;                                                       \
  } while (0)
