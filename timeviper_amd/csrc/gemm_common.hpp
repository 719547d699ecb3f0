// Shared by the two bf16 GEMM kernels (gemm.hip: one work-group per tile, any shape; gemm_persist.hip: persistent
// work-groups with the tile epilogue hidden in the main loop, the ViT's big shapes).
#pragma once
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "ssd_common.hpp"

namespace tvgemm {
using namespace ssdk;

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_ACCUM = 2 };

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;          // one half-tile: 128 rows x 128 bytes
constexpr int TILE_BYTES = 4 * HALF_BYTES;        // A0 A1 W0 W1 of one K-tile
constexpr int RING_BYTES = 2 * TILE_BYTES;        // two K-tiles

struct GemmArgs {
  const bf16_t *A, *W;
  const void* bias;        // fp32 or bf16 (bias_f32), may be NULL
  bf16_t* C;
  int M, N, K;
  int64_t lda, ldw, ldc;
  int tiles_m, tiles_n, group_m;
  int bias_f32;
};

// same formula as norms.hip's gelu_erf (A&S 7.1.26): the fused and the two-pass path agree bit for bit
__device__ __forceinline__ float gelu_erf_f(float x) {
  // A&S 7.1.26 with 0.5 sqrt(2) folded into the polynomial: gelu(x) = x/2 + |x|/2 erf(|x| / sqrt 2) = x/2 + z (k erf(z)),
  // z = |x| / sqrt 2, k = sqrt(2) / 2 — no copysign, no |x/2| (the packed fp32 forms have no abs modifier), one multiply less
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.f));
  float p = __builtin_fmaf(0.750526965f, t, -1.02753365f);          // k x {1.061405429, -1.453152027, 1.421413741, -0.284496736, 0.254829592}
  p = __builtin_fmaf(p, t, 1.00509131f);
  p = __builtin_fmaf(p, t, -0.201169565f);
  p = __builtin_fmaf(p, t, 0.180191725f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
  const float r = __builtin_fmaf(-p, e, 0.70710678118654752f);       // k erf(z)
  return __builtin_fmaf(z, r, 0.5f * x);
}

#define GEMM_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// gemm_persist.hip: returns TV_ERR_UNSUPPORTED (without setting the error text) when the shape is not one the persistent
// kernel takes; the caller then runs the per-tile kernel.
bool persist_takes(const GemmArgs& a, int epilogue);
int launch_persist(const GemmArgs& a, int epilogue, hipStream_t st);
int persist_grid();        // work-groups of the persistent kernels (one per compute unit unless tv_gemm_set_persist says otherwise)

// gemm_drip.hip: persistent 256 x 192 tiles whose finished tile leaves during the next tile's K loop
bool drip_takes(const GemmArgs& a, int epilogue, int grid);
int launch_drip(const GemmArgs& a, int epilogue, int grid, hipStream_t st);

}  // namespace tvgemm
