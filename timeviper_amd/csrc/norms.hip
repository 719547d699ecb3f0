// L1 / S4: RMSNorm (+fused residual add) and gated grouped RMSNorm. HBM-bound:
// every element is read once with 16-byte accesses, kept in registers across
// the reduction, and written once.
// Reference semantics: NemotronHRMSNorm.forward modeling_nano.py:897-903,
// NemotronHBlock residual add :966, MambaRMSNormGated :371-380 (mamba_ssm
// rmsnorm_fn with norm_before_gate=False).
#include <stdlib.h>
#include "common.hpp"

// -DTV_NORM_SHFL (dev, A/B): the row sums on the __shfl_xor butterfly (one ds_bpermute round trip per step) instead of DPP
#ifdef TV_NORM_SHFL
#define wave_sum_dpp wave_sum
#endif
namespace {

constexpr int NORM_THREADS = 256;
constexpr int NORM_MAXV = 4;  // 16-byte vectors cached per thread

template <typename T>
__device__ __forceinline__ float wload(const void* w, int wf32, int i) {
  return wf32 ? ((const float*)w)[i] : to_f32(((const T*)w)[i]);
}
// the V weights of one 16-byte activation vector, as floats (fp32 weights: V*4 bytes = two or
// one 16-byte loads; activation-dtype weights: one)
template <typename T>
__device__ __forceinline__ void wload_vec(const void* w, int wf32, int i0, float (&out)[Vec16<T>::N]) {
  constexpr int V = Vec16<T>::N;
  if (wf32) {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      const f32x4 v = *(const f32x4*)((const float*)w + i0 + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) out[4 * q + i] = v[i];
    }
  } else {
    const typename Vec16<T>::type v = *(const typename Vec16<T>::type*)((const T*)w + i0);
#pragma unroll
    for (int i = 0; i < V; ++i) out[i] = to_f32(v[i]);
  }
}

template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void rmsnorm_kernel(
    const T* __restrict__ x, const T* __restrict__ delta, const void* __restrict__ w,
    T* __restrict__ sum_out, T* __restrict__ y, int D, int64_t xs, int64_t ds, int64_t ss,
    int64_t ys, float eps, int wf32) {
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  __shared__ float red[NORM_THREADS / 64];
  const int64_t row = blockIdx.x;
  const int nv = D / V;
  const T* xr = x + row * xs;
  const T* dr = delta ? delta + row * ds : nullptr;
  float vals[NORM_MAXV][V];
  float ssq = 0.f;
#pragma unroll
  for (int k = 0; k < NORM_MAXV; ++k) {
    const int iv = threadIdx.x + k * NORM_THREADS;
    if (iv < nv) {
      vec_t v = *(const vec_t*)(xr + (int64_t)iv * V);
      if (dr) {
        vec_t d = *(const vec_t*)(dr + (int64_t)iv * V);
        vec_t sv;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          // the reference adds in the activation dtype: round, then normalise
          sv[i] = from_f32<T>(to_f32(v[i]) + to_f32(d[i]));
          vals[k][i] = to_f32(sv[i]);
        }
        if (sum_out) *(vec_t*)(sum_out + row * ss + (int64_t)iv * V) = sv;
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) vals[k][i] = to_f32(v[i]);
        if (sum_out) *(vec_t*)(sum_out + row * ss + (int64_t)iv * V) = v;
      }
#pragma unroll
      for (int i = 0; i < V; ++i) ssq = fmaf(vals[k][i], vals[k][i], ssq);
    }
  }
  ssq = wave_sum_dpp(ssq);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ssq;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_THREADS / 64; ++i) tot += red[i];
  const float rstd = rsqrtf(tot / (float)D + eps);
#pragma unroll
  for (int k = 0; k < NORM_MAXV; ++k) {
    const int iv = threadIdx.x + k * NORM_THREADS;
    if (iv < nv) {
      vec_t o;
      float wv[V];
      wload_vec<T>(w, wf32, iv * V, wv);
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = from_f32<T>(wv[i] * (vals[k][i] * rstd));
      *(vec_t*)(y + row * ys + (int64_t)iv * V) = o;
    }
  }
}

// LayerNorm with optional fused residual add (ViT blocks): s = x (+ delta); y = (s-mean)*rstd*w + b
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void layernorm_kernel(
    const T* __restrict__ x, const T* __restrict__ delta, const T* __restrict__ w,
    const T* __restrict__ bias, const float* __restrict__ row_bias, T* __restrict__ sum_out,
    T* __restrict__ y, int D, int64_t xs, int64_t ds, int64_t ss, int64_t ys, float eps) {
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  __shared__ float red[2][NORM_THREADS / 64];
  const int64_t row = blockIdx.x;
  const int nv = D / V;
  const T* xr = x + row * xs;
  const T* dr = delta ? delta + row * ds : nullptr;
  float vals[NORM_MAXV][V];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NORM_MAXV; ++k) {
    const int iv = threadIdx.x + k * NORM_THREADS;
    if (iv < nv) {
      vec_t v = *(const vec_t*)(xr + (int64_t)iv * V);
      if (dr) {
        vec_t d = *(const vec_t*)(dr + (int64_t)iv * V);
        vec_t sv;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          sv[i] = from_f32<T>(to_f32(v[i]) + to_f32(d[i]));
          vals[k][i] = to_f32(sv[i]);
        }
        if (sum_out) *(vec_t*)(sum_out + row * ss + (int64_t)iv * V) = sv;
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) vals[k][i] = to_f32(v[i]);
      }
      if (row_bias) {      // a constant row added before the statistics (sub-layer biases carried beside the stream)
#pragma unroll
        for (int i = 0; i < V; ++i) vals[k][i] += row_bias[(int64_t)iv * V + i];
      }
#pragma unroll
      for (int i = 0; i < V; ++i) s1 += vals[k][i];
    }
  }
  // two passes over the registers (mean, then sum of squared deviations), like torch's
  // LayerNorm: E[x^2] - mean^2 loses the variance of rows with |mean| >> std to cancellation
  s1 = wave_sum_dpp(s1);
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s1;
  __syncthreads();
  float t1 = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_THREADS / 64; ++i) t1 += red[0][i];
  const float mean = t1 / (float)D;
#pragma unroll
  for (int k = 0; k < NORM_MAXV; ++k) {
    const int iv = threadIdx.x + k * NORM_THREADS;
    if (iv < nv) {
#pragma unroll
      for (int i = 0; i < V; ++i) { const float d = vals[k][i] - mean; s2 = fmaf(d, d, s2); }
    }
  }
  s2 = wave_sum_dpp(s2);
  if ((threadIdx.x & 63) == 0) red[1][threadIdx.x >> 6] = s2;
  __syncthreads();
  float t2 = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_THREADS / 64; ++i) t2 += red[1][i];
  const float var = t2 / (float)D;
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int k = 0; k < NORM_MAXV; ++k) {
    const int iv = threadIdx.x + k * NORM_THREADS;
    if (iv < nv) {
      const vec_t wv = *(const vec_t*)(w + (int64_t)iv * V);
      vec_t bv = {};
      if (bias) bv = *(const vec_t*)(bias + (int64_t)iv * V);
      vec_t o;
#pragma unroll
      for (int i = 0; i < V; ++i)
        o[i] = from_f32<T>((vals[k][i] - mean) * rstd * to_f32(wv[i]) + to_f32(bv[i]));
      *(vec_t*)(y + row * ys + (int64_t)iv * V) = o;
    }
  }
}

// Rows of up to 64 * NORM_MAXV vectors (ViT widths): one WAVE per row — no LDS, no block
// barrier, four rows in flight per workgroup, every load issued before the first use.
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void layernorm_wave_kernel(
    const T* __restrict__ x, const T* __restrict__ delta, const T* __restrict__ w,
    const T* __restrict__ bias, const float* __restrict__ row_bias, T* __restrict__ sum_out,
    T* __restrict__ y, int64_t rows, int D, int64_t xs, int64_t ds, int64_t ss, int64_t ys, float eps) {
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  const int64_t row = (int64_t)blockIdx.x * (NORM_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int nv = D / V;
  const T* xr = x + row * xs;
  const T* dr = delta ? delta + row * ds : nullptr;
  vec_t xv[NORM_MAXV], dv[NORM_MAXV];
#pragma unroll
  for (int k = 0; k < NORM_MAXV; ++k) {
    const int iv = lane + k * 64;
    if (iv < nv) {
      xv[k] = *(const vec_t*)(xr + (int64_t)iv * V);
      if (dr) dv[k] = *(const vec_t*)(dr + (int64_t)iv * V);
    }
  }
  float vals[NORM_MAXV][V];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NORM_MAXV; ++k) {
    const int iv = lane + k * 64;
    if (iv < nv) {
      if (dr) {
        vec_t sv;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          sv[i] = from_f32<T>(to_f32(xv[k][i]) + to_f32(dv[k][i]));
          vals[k][i] = to_f32(sv[i]);
        }
        if (sum_out) *(vec_t*)(sum_out + row * ss + (int64_t)iv * V) = sv;
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) vals[k][i] = to_f32(xv[k][i]);
      }
      if (row_bias) {
#pragma unroll
        for (int i = 0; i < V; ++i) vals[k][i] += row_bias[(int64_t)iv * V + i];
      }
#pragma unroll
      for (int i = 0; i < V; ++i) s1 += vals[k][i];
    }
  }
  s1 = wave_sum_dpp(s1);
  const float mean = s1 / (float)D;
  // second register pass: sum of squared deviations (no E[x^2] - mean^2 cancellation)
#pragma unroll
  for (int k = 0; k < NORM_MAXV; ++k) {
    const int iv = lane + k * 64;
    if (iv < nv) {
#pragma unroll
      for (int i = 0; i < V; ++i) { const float d = vals[k][i] - mean; s2 = fmaf(d, d, s2); }
    }
  }
  s2 = wave_sum_dpp(s2);
  const float var = s2 / (float)D;
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int k = 0; k < NORM_MAXV; ++k) {
    const int iv = lane + k * 64;
    if (iv < nv) {
      const vec_t wv = *(const vec_t*)(w + (int64_t)iv * V);
      vec_t bv = {};
      if (bias) bv = *(const vec_t*)(bias + (int64_t)iv * V);
      vec_t o;
#pragma unroll
      for (int i = 0; i < V; ++i)
        o[i] = from_f32<T>((vals[k][i] - mean) * rstd * to_f32(wv[i]) + to_f32(bv[i]));
      *(vec_t*)(y + row * ys + (int64_t)iv * V) = o;
    }
  }
}

// The same rows, RPW consecutive ones per wave: weight, bias and the fp32 row bias are loaded ONCE per wave and stay in
// registers, and row r + 1 is requested before row r is reduced.  Per row the kernel above issues 5 loads for every
// 16 bytes of x it reads (x, 2 of the row bias, w, b: 11.5 KB through the CU's L1 for 2.3 KB from HBM); at the ViT's
// 1.5 M rows x 1 152 that load path, not HBM, set the pace of the row-bias form (4.9 against 5.3 TB/s for the plain one).
// NV = 16-byte vectors per lane (ceil(D / V / 64)), a template parameter so that the register count follows the row.
template <typename T, int NV, int RPW, bool DELTA>
__global__ __launch_bounds__(NORM_THREADS, DELTA ? 3 : 4) void layernorm_rows_kernel(
    const T* __restrict__ x, const T* __restrict__ delta, const T* __restrict__ w,
    const T* __restrict__ bias, const float* __restrict__ row_bias, T* __restrict__ sum_out,
    T* __restrict__ y, int64_t rows, int D, int64_t xs, int64_t ds, int64_t ss, int64_t ys, float eps) {
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  const int64_t row0 = ((int64_t)blockIdx.x * (NORM_THREADS / 64) + (threadIdx.x >> 6)) * RPW;
  if (row0 >= rows) return;
  const int lane = threadIdx.x & 63;
  const int nv = D / V;
  const int nrow = (int)(rows - row0 < RPW ? rows - row0 : RPW);
  vec_t wv[NV], bv[NV];
  float rb[NV][V];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int iv = lane + k * 64;
    wv[k] = vec_t{};
    bv[k] = vec_t{};
#pragma unroll
    for (int i = 0; i < V; ++i) rb[k][i] = 0.f;
    if (iv < nv) {
      wv[k] = *(const vec_t*)(w + (int64_t)iv * V);
      if (bias) bv[k] = *(const vec_t*)(bias + (int64_t)iv * V);
      if (row_bias) {
#pragma unroll
        for (int i = 0; i < V; i += 4) {
          const f32x4 r4 = *(const f32x4*)(row_bias + (int64_t)iv * V + i);
#pragma unroll
          for (int j = 0; j < 4; ++j) rb[k][i + j] = r4[j];
        }
      }
    }
  }
  constexpr int NVD = DELTA ? NV : 1;
  vec_t xv[NV], dv[NVD], xn[NV], dn[NVD];
  auto load_row = [&](int64_t row, vec_t (&xo)[NV], vec_t (&dd)[NVD]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int iv = lane + k * 64;
      if (iv < nv) {
        xo[k] = *(const vec_t*)(x + row * xs + (int64_t)iv * V);
        if (DELTA) dd[k % NVD] = *(const vec_t*)(delta + row * ds + (int64_t)iv * V);
      }
    }
  };
  load_row(row0, xv, dv);
  for (int r = 0; r < nrow; ++r) {
    const int64_t row = row0 + r;
    if (r + 1 < nrow) load_row(row + 1, xn, dn);
    float vals[NV][V];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int iv = lane + k * 64;
      if (iv < nv) {
        if (DELTA) {
          vec_t sv;
#pragma unroll
          for (int i = 0; i < V; ++i) {
            sv[i] = from_f32<T>(to_f32(xv[k][i]) + to_f32(dv[k % NVD][i]));
            vals[k][i] = to_f32(sv[i]);
          }
          if (sum_out) *(vec_t*)(sum_out + row * ss + (int64_t)iv * V) = sv;
        } else {
#pragma unroll
          for (int i = 0; i < V; ++i) vals[k][i] = to_f32(xv[k][i]);
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
          vals[k][i] += rb[k][i];
          s1 += vals[k][i];
        }
      }
    }
    s1 = wave_sum_dpp(s1);
    const float mean = s1 / (float)D;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int iv = lane + k * 64;
      if (iv < nv) {
#pragma unroll
        for (int i = 0; i < V; ++i) { const float d = vals[k][i] - mean; s2 = fmaf(d, d, s2); }
      }
    }
    s2 = wave_sum_dpp(s2);
    const float rstd = rsqrtf(s2 / (float)D + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int iv = lane + k * 64;
      if (iv < nv) {
        vec_t o;
#pragma unroll
        for (int i = 0; i < V; ++i)
          o[i] = from_f32<T>((vals[k][i] - mean) * rstd * to_f32(wv[k][i]) + to_f32(bv[k][i]));
        *(vec_t*)(y + row * ys + (int64_t)iv * V) = o;
      }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) xv[k] = xn[k];
#pragma unroll
    for (int k = 0; k < NVD; ++k) dv[k] = dn[k];
  }
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, the size of fp32 erff's own
// rounding): branch-free, 2 transcendental + ~12 plain VALU ops per element, so the
// kernel is bound by HBM and not by libm's piecewise erff (~45 ops with divergent paths).
// gelu(x) = 0.5 x (1 + erf(x / sqrt 2)), the reference's nn.GELU() formula.
__device__ __forceinline__ float gelu_erf(float x) {
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

// relu(x)^2 (NemotronHMLP's "relu2" activation, modeling_nano.py:993-994)
__device__ __forceinline__ float relu2(float x) {
  const float r = fmaxf(x, 0.f);
  return r * r;
}
enum { ACT_GELU = 0, ACT_RELU2 = 1 };
template <int ACT> __device__ __forceinline__ float act_apply(float x) {
  return ACT == ACT_GELU ? gelu_erf(x) : relu2(x);
}

// elementwise activation (exact-formula GELU / relu^2), 16 bytes per lane, grid-stride
template <typename T, int ACT>
__global__ __launch_bounds__(256) void gelu_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                   int64_t nvec, int64_t n) {
  constexpr int V = Vec16<T>::N;
  constexpr int U = 4;   // 16-byte loads in flight per lane before the first use
  typedef typename Vec16<T>::type vec_t;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < nvec; i += U * stride) {
    vec_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = *(const vec_t*)(x + (i + u * stride) * V);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      vec_t o;
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = from_f32<T>(act_apply<ACT>(to_f32(v[u][j])));
      *(vec_t*)(y + (i + u * stride) * V) = o;
    }
  }
  for (; i < nvec; i += stride) {
    const vec_t v = *(const vec_t*)(x + i * V);
    vec_t o;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = from_f32<T>(act_apply<ACT>(to_f32(v[j])));
    *(vec_t*)(y + i * V) = o;
  }
  if (blockIdx.x == 0) {   // ragged tail (< one vector)
    const int64_t k = nvec * V + threadIdx.x;
    if (k < n) y[k] = from_f32<T>(act_apply<ACT>(to_f32(x[k])));
  }
}

// one wave per (row, group)
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void rmsnorm_gated_kernel(
    const T* __restrict__ x, const T* __restrict__ z, const void* __restrict__ w,
    T* __restrict__ y, int64_t rows, int D, int gsz, int64_t xs, int64_t zs, int64_t ys,
    float eps, int wf32) {
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  const int ngroups = D / gsz;
  const int64_t wid = (int64_t)blockIdx.x * (NORM_THREADS / 64) + (threadIdx.x >> 6);
  if (wid >= rows * ngroups) return;
  const int64_t row = wid / ngroups;
  const int g = (int)(wid % ngroups);
  const int lane = threadIdx.x & 63;
  const int nv = gsz / V;
  const T* xr = x + row * xs + (int64_t)g * gsz;
  const T* zr = z ? z + row * zs + (int64_t)g * gsz : nullptr;
  // every load of the (row, group) is issued before the first use
  vec_t xv[NORM_MAXV], zv[NORM_MAXV];
#pragma unroll
  for (int k = 0; k < NORM_MAXV; ++k) {
    const int iv = lane + k * 64;
    if (iv < nv) {
      xv[k] = *(const vec_t*)(xr + (int64_t)iv * V);
      if (zr) zv[k] = *(const vec_t*)(zr + (int64_t)iv * V);
    }
  }
  float vals[NORM_MAXV][V];
  float ssq = 0.f;
#pragma unroll
  for (int k = 0; k < NORM_MAXV; ++k) {
    const int iv = lane + k * 64;
    if (iv < nv) {
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float v = to_f32(xv[k][i]);
        if (zr) {   // x * silu(z), sigmoid through the hardware reciprocal
          const float g = to_f32(zv[k][i]);
          v *= g * __builtin_amdgcn_rcpf(1.f + __expf(-g));
        }
        vals[k][i] = v;
        ssq = fmaf(v, v, ssq);
      }
    }
  }
  ssq = wave_sum_dpp(ssq);
  const float rstd = rsqrtf(ssq / (float)gsz + eps);
  T* yr = y + row * ys + (int64_t)g * gsz;
#pragma unroll
  for (int k = 0; k < NORM_MAXV; ++k) {
    const int iv = lane + k * 64;
    if (iv < nv) {
      vec_t o;
      float wv[V];
      wload_vec<T>(w, wf32, g * gsz + iv * V, wv);
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = from_f32<T>(wv[i] * (vals[k][i] * rstd));
      *(vec_t*)(yr + (int64_t)iv * V) = o;
    }
  }
}

template <typename T>
int launch_rms(const void* x, const void* delta, const void* w, void* sum_out, void* y,
               int64_t rows, int D, int64_t xs, int64_t ds, int64_t ss, int64_t ys, float eps,
               int wf32, hipStream_t s) {
  constexpr int V = Vec16<T>::N;
  if (D % V || D / V > NORM_THREADS * NORM_MAXV)
    TV_UNSUPPORTED("rmsnorm: dim %d not a multiple of %d or larger than %d", D, V,
                   V * NORM_THREADS * NORM_MAXV);
  rmsnorm_kernel<T><<<dim3((unsigned)rows), NORM_THREADS, 0, s>>>(
      (const T*)x, (const T*)delta, w, (T*)sum_out, (T*)y, D, xs, ds, ss, ys, eps, wf32);
  TV_LAUNCH_CHECK();
}

template <typename T>
int launch_gated(const void* x, const void* z, const void* w, void* y, int64_t rows, int D,
                 int gsz, int64_t xs, int64_t zs, int64_t ys, float eps, int wf32,
                 hipStream_t s) {
  constexpr int V = Vec16<T>::N;
  if (gsz % V || gsz / V > 64 * NORM_MAXV)
    TV_UNSUPPORTED("rmsnorm_gated: group_size %d not a multiple of %d or larger than %d", gsz,
                   V, V * 64 * NORM_MAXV);
  const int64_t nwaves = rows * (D / gsz);
  const int64_t nblk = (nwaves + NORM_THREADS / 64 - 1) / (NORM_THREADS / 64);
  rmsnorm_gated_kernel<T><<<dim3((unsigned)nblk), NORM_THREADS, 0, s>>>(
      (const T*)x, (const T*)z, w, (T*)y, rows, D, gsz, xs, zs, ys, eps, wf32);
  TV_LAUNCH_CHECK();
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <typename T>
int launch_ln(const void* x, const void* delta, const void* w, const void* b, const float* rb, void* sum_out,
              void* y, int64_t rows, int D, int64_t xs, int64_t ds, int64_t ss, int64_t ys, float eps,
              hipStream_t s) {
  constexpr int V = Vec16<T>::N;
  if (D % V || D / V > NORM_THREADS * NORM_MAXV)
    TV_UNSUPPORTED("layernorm: dim %d not a multiple of %d or larger than %d", D, V,
                   V * NORM_THREADS * NORM_MAXV);
  constexpr int RPW = 8;
  static const int rows_off = [] { const char* e = getenv("TV_LN_ROWS"); return e && atoi(e) == 0; }();     // dev: A/B
  if (D / V <= 64 * NORM_MAXV && rows >= 4096 * RPW && !rows_off && !delta) {
    // many rows (the ViT's stream form: row bias, no residual operand): a wave keeps the parameters in registers for RPW
    // rows — 1 272 against 1 416 us per 1.49 M rows x 1 152 (5.41 against 4.86 TB/s); with a residual operand the
    // one-row kernel is the faster one (2 502 against 2 654 us: the second operand's registers cost a wave per SIMD)
    const int64_t nblk = (rows + (NORM_THREADS / 64) * RPW - 1) / ((NORM_THREADS / 64) * RPW);
    const int nvl = (D / V + 63) / 64;
#define TV_LN_ROWS(NV)                                                                                              \
    layernorm_rows_kernel<T, NV, RPW, false><<<dim3((unsigned)nblk), NORM_THREADS, 0, s>>>(                         \
        (const T*)x, nullptr, (const T*)w, (const T*)b, rb, (T*)sum_out, (T*)y, rows, D, xs, ds, ss, ys, eps)
    if (nvl == 1) TV_LN_ROWS(1);
    else if (nvl == 2) TV_LN_ROWS(2);
    else if (nvl == 3) TV_LN_ROWS(3);
    else TV_LN_ROWS(4);
#undef TV_LN_ROWS
  } else if (D / V <= 64 * NORM_MAXV) {
    const int64_t nblk = (rows + NORM_THREADS / 64 - 1) / (NORM_THREADS / 64);
    layernorm_wave_kernel<T><<<dim3((unsigned)nblk), NORM_THREADS, 0, s>>>(
        (const T*)x, (const T*)delta, (const T*)w, (const T*)b, rb, (T*)sum_out, (T*)y, rows, D, xs, ds, ss,
        ys, eps);
  } else {
    layernorm_kernel<T><<<dim3((unsigned)rows), NORM_THREADS, 0, s>>>(
        (const T*)x, (const T*)delta, (const T*)w, (const T*)b, rb, (T*)sum_out, (T*)y, D, xs, ds, ss, ys, eps);
  }
  TV_LAUNCH_CHECK();
}

}  // namespace

extern "C" int tv_rmsnorm_fwd(const void* x, const void* delta, const void* weight,
                              void* sum_out, void* y, int64_t rows, int dim, int64_t x_stride,
                              int64_t delta_stride, int64_t sum_stride, int64_t y_stride,
                              float eps, int dtype, int wdtype, void* stream) {
  TV_CHECK_ARG(weight && (rows == 0 || (x && y)), "rmsnorm: null pointer");   // empty tensors have no storage
  TV_CHECK_ARG(rows >= 0 && dim > 0, "rmsnorm: bad sizes");
  if (rows == 0) return TV_OK;
  if (wdtype != TV_F32 && wdtype != dtype) TV_UNSUPPORTED("rmsnorm: wdtype must be f32 or dtype");
  const int vec = dtype == TV_F32 ? 4 : 8;
  if (!aligned16(x) || !aligned16(y) || !aligned16(weight) || (delta && !aligned16(delta)) ||
      (sum_out && !aligned16(sum_out)) || x_stride % vec || y_stride % vec ||
      (delta && delta_stride % vec) || (sum_out && sum_stride % vec))
    TV_UNSUPPORTED("rmsnorm: pointers/strides must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int wf32 = wdtype == TV_F32;
  switch (dtype) {
    case TV_F32:
      return launch_rms<float>(x, delta, weight, sum_out, y, rows, dim, x_stride, delta_stride,
                               sum_stride, y_stride, eps, 1, s);
    case TV_BF16:
      return launch_rms<bf16_t>(x, delta, weight, sum_out, y, rows, dim, x_stride, delta_stride,
                                sum_stride, y_stride, eps, wf32, s);
    case TV_F16:
      return launch_rms<f16_t>(x, delta, weight, sum_out, y, rows, dim, x_stride, delta_stride,
                               sum_stride, y_stride, eps, wf32, s);
  }
  TV_UNSUPPORTED("rmsnorm: dtype %d", dtype);
}

extern "C" int tv_rmsnorm_gated_fwd(const void* x, const void* z, const void* weight, void* y,
                                    int64_t rows, int dim, int group_size, int64_t x_stride,
                                    int64_t z_stride, int64_t y_stride, float eps, int dtype,
                                    int wdtype, void* stream) {
  TV_CHECK_ARG(weight && (rows == 0 || (x && y)), "rmsnorm_gated: null pointer");
  TV_CHECK_ARG(rows >= 0 && dim > 0 && group_size > 0 && dim % group_size == 0,
               "rmsnorm_gated: bad sizes (dim %d, group %d)", dim, group_size);
  if (rows == 0) return TV_OK;
  if (wdtype != TV_F32 && wdtype != dtype)
    TV_UNSUPPORTED("rmsnorm_gated: wdtype must be f32 or dtype");
  const int vec = dtype == TV_F32 ? 4 : 8;
  if (!aligned16(x) || !aligned16(y) || !aligned16(weight) || (z && !aligned16(z)) || x_stride % vec ||
      y_stride % vec || (z && z_stride % vec))
    TV_UNSUPPORTED("rmsnorm_gated: pointers/strides must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int wf32 = wdtype == TV_F32;
  switch (dtype) {
    case TV_F32:
      return launch_gated<float>(x, z, weight, y, rows, dim, group_size, x_stride, z_stride,
                                 y_stride, eps, 1, s);
    case TV_BF16:
      return launch_gated<bf16_t>(x, z, weight, y, rows, dim, group_size, x_stride, z_stride,
                                  y_stride, eps, wf32, s);
    case TV_F16:
      return launch_gated<f16_t>(x, z, weight, y, rows, dim, group_size, x_stride, z_stride,
                                 y_stride, eps, wf32, s);
  }
  TV_UNSUPPORTED("rmsnorm_gated: dtype %d", dtype);
}

extern "C" int tv_layernorm_fwd(const void* x, const void* delta, const void* weight,
                                const void* bias, const void* row_bias, void* sum_out, void* y, int64_t rows, int dim,
                                int64_t x_stride, int64_t delta_stride, int64_t sum_stride,
                                int64_t y_stride, float eps, int dtype, void* stream) {
  TV_CHECK_ARG(weight && (rows == 0 || (x && y)), "layernorm: null pointer");
  TV_CHECK_ARG(rows >= 0 && dim > 0, "layernorm: bad sizes");
  if (rows == 0) return TV_OK;
  const int vec = dtype == TV_F32 ? 4 : 8;
  if (!aligned16(x) || !aligned16(y) || !aligned16(weight) || (bias && !aligned16(bias)) || (row_bias && !aligned16(row_bias)) ||
      (delta && !aligned16(delta)) || (sum_out && !aligned16(sum_out)) || x_stride % vec ||
      y_stride % vec || (delta && delta_stride % vec) || (sum_out && sum_stride % vec))
    TV_UNSUPPORTED("layernorm: pointers/strides must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case TV_F32:
      return launch_ln<float>(x, delta, weight, bias, (const float*)row_bias, sum_out, y, rows, dim, x_stride, delta_stride,
                              sum_stride, y_stride, eps, s);
    case TV_BF16:
      return launch_ln<bf16_t>(x, delta, weight, bias, (const float*)row_bias, sum_out, y, rows, dim, x_stride,
                               delta_stride, sum_stride, y_stride, eps, s);
    case TV_F16:
      return launch_ln<f16_t>(x, delta, weight, bias, (const float*)row_bias, sum_out, y, rows, dim, x_stride, delta_stride,
                              sum_stride, y_stride, eps, s);
  }
  TV_UNSUPPORTED("layernorm: dtype %d", dtype);
}

namespace {
template <int ACT>
int launch_act(const char* name, const void* x, void* y, int64_t n, int dtype, void* stream) {
  if (n < 0) { tv_set_error("%s: bad size", name); return TV_ERR_BAD_ARG; }
  if (n == 0) return TV_OK;
  if (!x || !y) { tv_set_error("%s: null pointer", name); return TV_ERR_BAD_ARG; }
  const int vec = dtype == TV_F32 ? 4 : 8;
  if (!aligned16(x) || !aligned16(y)) TV_UNSUPPORTED("%s: pointers must be 16-byte aligned", name);
  const int64_t nvec = n / vec;
  const unsigned grid = (unsigned)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 + (nvec == 0) : 8192);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case TV_F32: gelu_kernel<float, ACT><<<grid, 256, 0, s>>>((const float*)x, (float*)y, nvec, n); break;
    case TV_BF16: gelu_kernel<bf16_t, ACT><<<grid, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, nvec, n); break;
    case TV_F16: gelu_kernel<f16_t, ACT><<<grid, 256, 0, s>>>((const f16_t*)x, (f16_t*)y, nvec, n); break;
    default: TV_UNSUPPORTED("%s: dtype %d", name, dtype);
  }
  TV_LAUNCH_CHECK();
}
}  // namespace

extern "C" int tv_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
  return launch_act<ACT_GELU>("gelu", x, y, n, dtype, stream);
}

extern "C" int tv_relu2_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
  return launch_act<ACT_RELU2>("relu2", x, y, n, dtype, stream);
}
