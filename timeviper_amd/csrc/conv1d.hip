// S2: causal depthwise conv1d (+bias +SiLU), channels-last, HBM-bound.
// Each lane owns 16 bytes of channels and slides a K-row register window down
// TL consecutive tokens, so every x row is read once per tile (+K-1 halo rows
// that hit L2) with fully coalesced 16-byte accesses.
// Reference semantics: modeling_nano.py:619-624 / :705 (zero left pad).
#include "common.hpp"

namespace {

constexpr int CONV_TL = 64;       // tokens per tile
constexpr int CONV_THREADS = 256;

template <typename T, int K>
__global__ __launch_bounds__(CONV_THREADS) void conv1d_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ bias,
    const T* __restrict__ halo, T* __restrict__ y, int L, int C, int64_t xsb,
    int64_t xsl, int64_t ysb, int64_t ysl, int silu) {
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  const int cv = blockIdx.x * CONV_THREADS + threadIdx.x;
  const int c0 = cv * V;
  if (c0 >= C) return;
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * CONV_TL;
  const int t1 = min(t0 + CONV_TL, L);

  float wk[K][V], bs[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
#pragma unroll
    for (int j = 0; j < K; ++j) wk[j][i] = to_f32(w[(int64_t)(c0 + i) * K + j]);
    bs[i] = bias ? to_f32(bias[c0 + i]) : 0.f;
  }

  const T* xb = x + (int64_t)b * xsb + c0;
  T* yb = y + (int64_t)b * ysb + c0;
  const T* hb = halo ? halo + (int64_t)b * (K - 1) * C + c0 : nullptr;

  // window[j] holds row t-(K-1)+j
  float win[K][V];
#pragma unroll
  for (int j = 0; j < K - 1; ++j) {
    const int t = t0 - (K - 1) + j;
    vec_t v;
    bool have = true;
    if (t >= 0) v = *(const vec_t*)(xb + (int64_t)t * xsl);
    else if (hb) v = *(const vec_t*)(hb + (int64_t)(t + K - 1) * C);
    else have = false;
#pragma unroll
    for (int i = 0; i < V; ++i) win[j][i] = have ? to_f32(v[i]) : 0.f;
  }

  for (int t = t0; t < t1; ++t) {
    vec_t v = *(const vec_t*)(xb + (int64_t)t * xsl);
    vec_t o;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      win[K - 1][i] = to_f32(v[i]);
      float acc = bs[i];
#pragma unroll
      for (int j = 0; j < K; ++j) acc = fmaf(wk[j][i], win[j][i], acc);
      if (silu) acc = silu_f(acc);
      o[i] = from_f32<T>(acc);
#pragma unroll
      for (int j = 0; j < K - 1; ++j) win[j][i] = win[j + 1][i];
    }
    *(vec_t*)(yb + (int64_t)t * ysl) = o;
  }
}

// Mamba-2 xBC variant: the three channel segments go to three destinations; B and C are
// written group-major (B,G,L,N) so that one group's consecutive tokens are contiguous.
struct XbcOut {
  void *yx, *yb, *yc;
  int d_inner, G, N;
};
template <typename T, int K>
__global__ __launch_bounds__(CONV_THREADS) void conv1d_xbc_kernel(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ bias,
    const T* __restrict__ halo, XbcOut o, int L, int C, int64_t xsb, int64_t xsl, int silu) {
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  const int cv = blockIdx.x * CONV_THREADS + threadIdx.x;
  const int c0 = cv * V;
  if (c0 >= C) return;
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * CONV_TL;
  const int t1 = min(t0 + CONV_TL, L);
  // destination of this lane's 16 bytes: row pointer at t=0 and row stride
  T* yb;
  int64_t ysl;
  const int gn = o.G * o.N;
  if (c0 < o.d_inner) {
    yb = (T*)o.yx + (int64_t)b * L * o.d_inner + c0;
    ysl = o.d_inner;
  } else {
    const bool isB = c0 < o.d_inner + gn;
    const int cc = c0 - o.d_inner - (isB ? 0 : gn);
    const int g = cc / o.N, n = cc - g * o.N;
    yb = (T*)(isB ? o.yb : o.yc) + ((int64_t)b * o.G + g) * L * o.N + n;
    ysl = o.N;
  }
  float wk[K][V], bs[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
#pragma unroll
    for (int j = 0; j < K; ++j) wk[j][i] = to_f32(w[(int64_t)(c0 + i) * K + j]);
    bs[i] = bias ? to_f32(bias[c0 + i]) : 0.f;
  }
  const T* xb = x + (int64_t)b * xsb + c0;
  const T* hb = halo ? halo + (int64_t)b * (K - 1) * C + c0 : nullptr;
  float win[K][V];
#pragma unroll
  for (int j = 0; j < K - 1; ++j) {
    const int t = t0 - (K - 1) + j;
    vec_t v;
    bool have = true;
    if (t >= 0) v = *(const vec_t*)(xb + (int64_t)t * xsl);
    else if (hb) v = *(const vec_t*)(hb + (int64_t)(t + K - 1) * C);
    else have = false;
#pragma unroll
    for (int i = 0; i < V; ++i) win[j][i] = have ? to_f32(v[i]) : 0.f;
  }
  auto step = [&](int t, const vec_t& v) {
    vec_t ov;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      win[K - 1][i] = to_f32(v[i]);
      float acc = bs[i];
#pragma unroll
      for (int j = 0; j < K; ++j) acc = fmaf(wk[j][i], win[j][i], acc);
      if (silu) acc *= __builtin_amdgcn_rcpf(1.f + __expf(-acc));   // x * sigmoid(x)
      ov[i] = from_f32<T>(acc);
#pragma unroll
      for (int j = 0; j < K - 1; ++j) win[j][i] = win[j + 1][i];
    }
    *(vec_t*)(yb + (int64_t)t * ysl) = ov;
  };
  // four rows of loads in flight ahead of the arithmetic
  constexpr int U = 4;
  int t = t0;
  for (; t + U <= t1; t += U) {
    vec_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = *(const vec_t*)(xb + (int64_t)(t + u) * xsl);
#pragma unroll
    for (int u = 0; u < U; ++u) step(t + u, v[u]);
  }
  for (; t < t1; ++t) step(t, *(const vec_t*)(xb + (int64_t)t * xsl));
}

template <typename T, int K>
__global__ void conv1d_update_kernel(const T* __restrict__ x, T* __restrict__ state,
                                     const T* __restrict__ w, const T* __restrict__ bias,
                                     T* __restrict__ y, int C, int silu) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (c >= C) return;
  T* st = state + ((int64_t)b * C + c) * K;
  float acc = bias ? to_f32(bias[c]) : 0.f;
  float win[K];
#pragma unroll
  for (int j = 0; j < K - 1; ++j) win[j] = to_f32(st[j + 1]);
  win[K - 1] = to_f32(x[(int64_t)b * C + c]);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    acc = fmaf(to_f32(w[(int64_t)c * K + j]), win[j], acc);
    st[j] = from_f32<T>(win[j]);
  }
  if (silu) acc = silu_f(acc);
  y[(int64_t)b * C + c] = from_f32<T>(acc);
}

template <typename T>
int launch_conv(const void* x, const void* w, const void* bias, const void* halo, void* y,
                int B, int L, int C, int K, int64_t xsb, int64_t xsl, int64_t ysb, int64_t ysl,
                int silu, hipStream_t s) {
  constexpr int V = Vec16<T>::N;
  dim3 grid((C / V + CONV_THREADS - 1) / CONV_THREADS, (L + CONV_TL - 1) / CONV_TL, B);
#define TV_CONV_CASE(KK)                                                                   \
  case KK:                                                                                 \
    conv1d_fwd_kernel<T, KK><<<grid, CONV_THREADS, 0, s>>>((const T*)x, (const T*)w,       \
        (const T*)bias, (const T*)halo, (T*)y, L, C, xsb, xsl, ysb, ysl, silu);            \
    break;
  switch (K) {
    TV_CONV_CASE(2)
    TV_CONV_CASE(3)
    TV_CONV_CASE(4)
  }
#undef TV_CONV_CASE
  TV_LAUNCH_CHECK();
}

template <typename T>
int launch_conv_xbc(const void* x, const void* w, const void* bias, const void* halo, XbcOut o,
                    int B, int L, int C, int K, int64_t xsb, int64_t xsl, int silu, hipStream_t s) {
  constexpr int V = Vec16<T>::N;
  dim3 grid((C / V + CONV_THREADS - 1) / CONV_THREADS, (L + CONV_TL - 1) / CONV_TL, B);
#define TV_CONVX_CASE(KK)                                                                  \
  case KK:                                                                                 \
    conv1d_xbc_kernel<T, KK><<<grid, CONV_THREADS, 0, s>>>((const T*)x, (const T*)w,       \
        (const T*)bias, (const T*)halo, o, L, C, xsb, xsl, silu);                          \
    break;
  switch (K) {
    TV_CONVX_CASE(2)
    TV_CONVX_CASE(3)
    TV_CONVX_CASE(4)
  }
#undef TV_CONVX_CASE
  TV_LAUNCH_CHECK();
}

template <typename T>
int launch_conv_update(const void* x, void* st, const void* w, const void* bias, void* y, int B,
                       int C, int K, int silu, hipStream_t s) {
  dim3 grid((C + 255) / 256, B);
#define TV_CONVU_CASE(KK)                                                                  \
  case KK:                                                                                 \
    conv1d_update_kernel<T, KK><<<grid, 256, 0, s>>>((const T*)x, (T*)st, (const T*)w,     \
                                                     (const T*)bias, (T*)y, C, silu);      \
    break;
  switch (K) {
    TV_CONVU_CASE(2)
    TV_CONVU_CASE(3)
    TV_CONVU_CASE(4)
  }
#undef TV_CONVU_CASE
  TV_LAUNCH_CHECK();
}

}  // namespace

extern "C" int tv_causal_conv1d_fwd(const void* x, const void* weight, const void* bias,
                                    const void* halo, void* y, int batch, int seqlen,
                                    int channels, int kernel, int64_t x_stride_b,
                                    int64_t x_stride_l, int64_t y_stride_b, int64_t y_stride_l,
                                    int dtype, int silu, void* stream) {
  TV_CHECK_ARG(weight && (seqlen == 0 || (x && y)), "conv1d: null pointer");   // empty tensors have no storage
  TV_CHECK_ARG(batch > 0 && seqlen >= 0 && channels > 0, "conv1d: bad sizes");
  if (kernel < 2 || kernel > 4) TV_UNSUPPORTED("conv1d: kernel width %d not in [2,4]", kernel);
  if (seqlen == 0) return TV_OK;
  hipStream_t s = (hipStream_t)stream;
  const int vec = dtype == TV_F32 ? 4 : 8;
  if (channels % vec || x_stride_l % vec || y_stride_l % vec || x_stride_b % vec ||
      y_stride_b % vec || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
    TV_UNSUPPORTED("conv1d: channels/strides/pointers must be 16-byte aligned");
  switch (dtype) {
    case TV_F32:
      return launch_conv<float>(x, weight, bias, halo, y, batch, seqlen, channels, kernel,
                                x_stride_b, x_stride_l, y_stride_b, y_stride_l, silu, s);
    case TV_BF16:
      return launch_conv<bf16_t>(x, weight, bias, halo, y, batch, seqlen, channels, kernel,
                                 x_stride_b, x_stride_l, y_stride_b, y_stride_l, silu, s);
    case TV_F16:
      return launch_conv<f16_t>(x, weight, bias, halo, y, batch, seqlen, channels, kernel,
                                x_stride_b, x_stride_l, y_stride_b, y_stride_l, silu, s);
  }
  TV_UNSUPPORTED("conv1d: dtype %d", dtype);
}

extern "C" int tv_causal_conv1d_xbc_fwd(const void* x, const void* weight, const void* bias,
                                        const void* halo, void* y_x, void* y_b, void* y_c,
                                        int batch, int seqlen, int d_inner, int ngroups,
                                        int dstate, int kernel, int64_t x_stride_b,
                                        int64_t x_stride_l, int dtype, int silu, void* stream) {
  TV_CHECK_ARG(weight && (seqlen == 0 || (x && y_x && y_b && y_c)), "conv1d_xbc: null pointer");
  TV_CHECK_ARG(batch > 0 && seqlen >= 0 && d_inner > 0 && ngroups > 0 && dstate > 0,
               "conv1d_xbc: bad sizes");
  if (kernel < 2 || kernel > 4) TV_UNSUPPORTED("conv1d_xbc: kernel width %d not in [2,4]", kernel);
  if (seqlen == 0) return TV_OK;
  const int vec = dtype == TV_F32 ? 4 : 8;
  const int C = d_inner + 2 * ngroups * dstate;
  if (d_inner % vec || dstate % vec || x_stride_l % vec || x_stride_b % vec ||
      ((uintptr_t)x & 15) || ((uintptr_t)y_x & 15) || ((uintptr_t)y_b & 15) || ((uintptr_t)y_c & 15))
    TV_UNSUPPORTED("conv1d_xbc: segments/strides/pointers must be 16-byte aligned");
  XbcOut o;
  o.yx = y_x; o.yb = y_b; o.yc = y_c; o.d_inner = d_inner; o.G = ngroups; o.N = dstate;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case TV_F32:
      return launch_conv_xbc<float>(x, weight, bias, halo, o, batch, seqlen, C, kernel, x_stride_b,
                                    x_stride_l, silu, s);
    case TV_BF16:
      return launch_conv_xbc<bf16_t>(x, weight, bias, halo, o, batch, seqlen, C, kernel, x_stride_b,
                                     x_stride_l, silu, s);
    case TV_F16:
      return launch_conv_xbc<f16_t>(x, weight, bias, halo, o, batch, seqlen, C, kernel, x_stride_b,
                                    x_stride_l, silu, s);
  }
  TV_UNSUPPORTED("conv1d_xbc: dtype %d", dtype);
}

extern "C" int tv_causal_conv1d_update(const void* x, void* conv_state, const void* weight,
                                       const void* bias, void* y, int batch, int channels,
                                       int kernel, int dtype, int silu, void* stream) {
  TV_CHECK_ARG(x && conv_state && weight && y, "conv1d_update: null pointer");
  TV_CHECK_ARG(batch > 0 && channels > 0, "conv1d_update: bad sizes");
  if (kernel < 2 || kernel > 4) TV_UNSUPPORTED("conv1d_update: kernel width %d", kernel);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case TV_F32:
      return launch_conv_update<float>(x, conv_state, weight, bias, y, batch, channels, kernel,
                                       silu, s);
    case TV_BF16:
      return launch_conv_update<bf16_t>(x, conv_state, weight, bias, y, batch, channels, kernel,
                                        silu, s);
    case TV_F16:
      return launch_conv_update<f16_t>(x, conv_state, weight, bias, y, batch, channels, kernel,
                                       silu, s);
  }
  TV_UNSUPPORTED("conv1d_update: dtype %d", dtype);
}
