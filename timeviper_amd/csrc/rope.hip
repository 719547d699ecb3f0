// Qwen2 ("next" row, BASELINE config 5): rotary position embedding applied in place to the q
// and k projections, and the SwiGLU gate  silu(gate) * up.  Elementwise, HBM-bound.
// Reference: apply_rotary_pos_emb / rotate_half (llm_repo/qwen2/modeling_qwen2.py:82-113,
// called :211-214), Qwen2MLP.forward (:78-80).
#include "common.hpp"

namespace {

// one lane = 8 elements (16 bytes) of the FIRST half of a head row and its partner in the
// second half:  lo' = lo*cos_lo - hi*sin_lo,  hi' = hi*cos_hi + lo*sin_hi
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ x, const T* __restrict__ cs,
                                                   const T* __restrict__ sn, int64_t rows, int H,
                                                   int D, int64_t xsl, int64_t xsh) {
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  const int hv = D / 2 / V;                              // vectors per half row
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * H * hv) return;
  const int v = (int)(i % hv);
  const int h = (int)((i / hv) % H);
  const int64_t l = i / ((int64_t)hv * H);
  T* p = x + l * xsl + (int64_t)h * xsh + v * V;
  const vec_t lo = *(const vec_t*)p, hi = *(const vec_t*)(p + D / 2);
  const vec_t cl = *(const vec_t*)(cs + l * D + v * V), ch = *(const vec_t*)(cs + l * D + D / 2 + v * V);
  const vec_t sl = *(const vec_t*)(sn + l * D + v * V), sh = *(const vec_t*)(sn + l * D + D / 2 + v * V);
  vec_t olo, ohi;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float a = to_f32(lo[e]), b = to_f32(hi[e]);
    olo[e] = from_f32<T>(a * to_f32(cl[e]) - b * to_f32(sl[e]));
    ohi[e] = from_f32<T>(b * to_f32(ch[e]) + a * to_f32(sh[e]));
  }
  *(vec_t*)p = olo;
  *(vec_t*)(p + D / 2) = ohi;
}

template <typename T>
__global__ __launch_bounds__(256) void silu_mul_kernel(const T* __restrict__ g, const T* __restrict__ u,
                                                       T* __restrict__ y, int64_t rows, int nvr,
                                                       int64_t gs, int64_t us, int64_t ys) {
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * nvr; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / nvr;
    const int c = (int)(i % nvr) * V;
    const vec_t a = *(const vec_t*)(g + r * gs + c), b = *(const vec_t*)(u + r * us + c);
    vec_t o;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float z = to_f32(a[e]);
      o[e] = from_f32<T>(z * __builtin_amdgcn_rcpf(1.f + __expf(-z)) * to_f32(b[e]));
    }
    *(vec_t*)(y + r * ys + c) = o;
  }
}

}  // namespace

extern "C" int tv_rope_fwd(void* x, const void* cos_, const void* sin_, int64_t rows, int heads,
                           int headdim, int64_t x_stride_l, int64_t x_stride_h, int dtype,
                           void* stream) {
  TV_CHECK_ARG(rows >= 0 && heads > 0 && headdim > 0, "rope: bad sizes");
  if (rows == 0) return TV_OK;
  TV_CHECK_ARG(x && cos_ && sin_, "rope: null pointer");
  const int vec = dtype == TV_F32 ? 4 : 8;
  if (headdim % (2 * vec) || x_stride_l % vec || x_stride_h % vec ||
      ((((uintptr_t)x) | ((uintptr_t)cos_) | ((uintptr_t)sin_)) & 15))
    TV_UNSUPPORTED("rope: head_dim must be a multiple of %d, pointers / strides 16-byte aligned", 2 * vec);
  const int64_t n = rows * heads * (headdim / 2 / vec);
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case TV_F32: rope_kernel<float><<<grid, 256, 0, s>>>((float*)x, (const float*)cos_, (const float*)sin_, rows, heads, headdim, x_stride_l, x_stride_h); break;
    case TV_BF16: rope_kernel<bf16_t><<<grid, 256, 0, s>>>((bf16_t*)x, (const bf16_t*)cos_, (const bf16_t*)sin_, rows, heads, headdim, x_stride_l, x_stride_h); break;
    case TV_F16: rope_kernel<f16_t><<<grid, 256, 0, s>>>((f16_t*)x, (const f16_t*)cos_, (const f16_t*)sin_, rows, heads, headdim, x_stride_l, x_stride_h); break;
    default: TV_UNSUPPORTED("rope: dtype %d", dtype);
  }
  TV_LAUNCH_CHECK();
}

extern "C" int tv_silu_mul_fwd(const void* gate, const void* up, void* y, int64_t rows, int dim,
                               int64_t gate_stride, int64_t up_stride, int64_t y_stride, int dtype,
                               void* stream) {
  TV_CHECK_ARG(rows >= 0 && dim > 0, "silu_mul: bad sizes");
  if (rows == 0) return TV_OK;
  TV_CHECK_ARG(gate && up && y, "silu_mul: null pointer");
  const int vec = dtype == TV_F32 ? 4 : 8;
  if (dim % vec || gate_stride % vec || up_stride % vec || y_stride % vec ||
      ((((uintptr_t)gate) | ((uintptr_t)up) | ((uintptr_t)y)) & 15))
    TV_UNSUPPORTED("silu_mul: dim / strides / pointers must be 16-byte multiples");
  const int nvr = dim / vec;
  const int64_t n = rows * nvr;
  const unsigned grid = (unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case TV_F32: silu_mul_kernel<float><<<grid, 256, 0, s>>>((const float*)gate, (const float*)up, (float*)y, rows, nvr, gate_stride, up_stride, y_stride); break;
    case TV_BF16: silu_mul_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)gate, (const bf16_t*)up, (bf16_t*)y, rows, nvr, gate_stride, up_stride, y_stride); break;
    case TV_F16: silu_mul_kernel<f16_t><<<grid, 256, 0, s>>>((const f16_t*)gate, (const f16_t*)up, (f16_t*)y, rows, nvr, gate_stride, up_stride, y_stride); break;
    default: TV_UNSUPPORTED("silu_mul: dtype %d", dtype);
  }
  TV_LAUNCH_CHECK();
}
