// S2: causal depthwise conv1d (+bias +SiLU), channels-last, HBM-bound.
// Each lane owns 16 bytes of channels and slides a K-row register window down
// TL consecutive tokens, so every x row is read once per tile (+K-1 halo rows
// that hit L2) with fully coalesced 16-byte accesses.
// Reference semantics: modeling_nano.py:619-624 / :705 (zero left pad).
#include "ssd_common.hpp"

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
    const T* __restrict__ halo, XbcOut o, int L, int C, int c_end, int64_t xsb, int64_t xsl, int silu) {
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  const int cv = blockIdx.x * CONV_THREADS + threadIdx.x;
  const int c0 = cv * V;
  if (c0 >= c_end) return;       // c_end = C, or d_inner when the B / C channels run in conv1d_bc_cb_kernel
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


// ------------------------------------------------------------------ B / C channels + causal C.B^T of a chunk
// The B and C channels of ONE (64-token chunk, group): convolution + SiLU exactly as conv1d_xbc_kernel
// (same fp32 operations in the same order), group-major stores — and, while the chunk's 64 x 128 B and C tiles are
// still on the chip (LDS), the causal part of C.B^T in the fragment order the scan kernel's mask waves read
// (ssd_slice.hip: 6 fragments of 512 bf16 per (chunk, group); fragment (ti, sp), lane (lc, kq) holds
// CB[16 ti + lc][32 sp + 8 kq + 0..7]).  Round 2 recomputed it in a pre-pass of the scan that read B and C back from
// memory (0.8 GB and ~100 us per layer at 164 k tokens).
// grid (nchunks, G, batch), 256 threads: thread = (strip of 8 tokens, 16 bytes of the group's 256 B / C channels).
constexpr int CBQ = 64, CBN = 128, CB_FRAGS = 6;
template <int K>
__global__ __launch_bounds__(256) void conv1d_bc_cb_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias,
    const bf16_t* __restrict__ halo, bf16_t* __restrict__ yb, bf16_t* __restrict__ yc, bf16_t* __restrict__ cb,
    int L, int C, int d_inner, int G, int nchunks, int64_t xsb, int64_t xsl, int silu) {
  using namespace ssdk;
  __shared__ __attribute__((aligned(16))) bf16_t tile[2][CBQ * CBN];     // [B | C][token][n], 16-byte chunk ^ (token & 15)
  const int c = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  const int strip = tid >> 5, l32 = tid & 31;
  const int isC = l32 >> 4, nch = l32 & 15;                // 16-byte chunk of the group's 128 states
  const int c0 = d_inner + (isC ? G * CBN : 0) + g * CBN + nch * 8;
  const int t0 = c * CBQ + strip * 8;
  float wk[K][8], bs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < K; ++j) wk[j][i] = to_f32(w[(int64_t)(c0 + i) * K + j]);
    bs[i] = bias ? to_f32(bias[c0 + i]) : 0.f;
  }
  const bf16_t* xb = x + (int64_t)b * xsb + c0;
  const bf16_t* hb = halo ? halo + (int64_t)b * (K - 1) * C + c0 : nullptr;
  bf16_t* yg = (isC ? yc : yb) + ((int64_t)b * G + g) * L * CBN + nch * 8;
  bf16_t* tl = tile[isC];
  // all rows of the strip (and its K-1 predecessors) in flight, then the arithmetic
  bf16x8 v[K - 1 + 8];
#pragma unroll
  for (int j = 0; j < K - 1 + 8; ++j) {
    const int t = t0 - (K - 1) + j;
    bf16x8 z = {};
    if (t >= 0 && t < L) v[j] = *(const bf16x8*)(xb + (int64_t)t * xsl);
    else if (t < 0 && hb) v[j] = *(const bf16x8*)(hb + (int64_t)(t + K - 1) * C);
    else v[j] = z;
  }
  float win[K][8];
#pragma unroll
  for (int j = 0; j < K - 1; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) win[j][i] = to_f32(v[j][i]);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int t = t0 + u;
    bf16x8 ov;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      win[K - 1][i] = to_f32(v[K - 1 + u][i]);
      float acc = bs[i];
#pragma unroll
      for (int j = 0; j < K; ++j) acc = fmaf(wk[j][i], win[j][i], acc);
      if (silu) acc *= __builtin_amdgcn_rcpf(1.f + __expf(-acc));   // x * sigmoid(x)
      ov[i] = (bf16_t)acc;
#pragma unroll
      for (int j = 0; j < K - 1; ++j) win[j][i] = win[j + 1][i];
    }
    if (t >= L) ov = bf16x8{};                    // rows past the sequence end: finite, never stored
    else *(bf16x8*)(yg + (int64_t)t * CBN) = ov;
    const int row = strip * 8 + u;
    *(bf16x8*)(tl + row * CBN + ((nch ^ (row & 15)) << 3)) = ov;
  }
  __syncthreads();
  if (!cb) return;
  const int lane = tid & 63, wave = tid >> 6;
  if (wave == 3) return;
  const int lc = lane & 15, kq = lane >> 4;
  // as ssd_cb_kernel: wave 0: fragments (2,0) (2,1); wave 1: (3,0) (3,1); wave 2: (0,0) (1,0)
  const int nct = wave == 2 ? 2 : 1, nsp = wave == 2 ? 1 : 2;
  const int ti0 = wave == 0 ? 2 : wave == 1 ? 3 : 0;
  auto rd = [&](const bf16_t* tlp, int row, int ks) {
    return *(const bf16x8*)(tlp + row * CBN + (((4 * ks + kq) ^ (row & 15)) << 3));
  };
  bf16x8 cf[2][4], b0[2][4], b1[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (u < nct) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) cf[u][ks] = rd(tile[1], 16 * (ti0 + u) + lc, ks);
    }
    if (u < nsp) {
      const int s_lo = 32 * u + 8 * (lc >> 2) + (lc & 3);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        b0[u][ks] = rd(tile[0], s_lo, ks);
        b1[u][ks] = rd(tile[0], s_lo + 4, ks);
      }
    }
  }
#pragma unroll
  for (int ff = 0; ff < 2; ++ff) {
    const int f = wave == 0 ? 2 + ff : wave == 1 ? 4 + ff : ff;
    const int ci = wave == 2 ? ff : 0, si = wave == 2 ? 0 : ff;
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      d0 = mfma16(b0[si][ks], cf[ci][ks], d0);
      d1 = mfma16(b1[si][ks], cf[ci][ks], d1);
    }
    // elements above the diagonal (s > t) are stored as zeros: the head-per-wave march uses the fragments as they are
    const int f_ti = f == 0 ? 0 : f == 1 ? 1 : f < 4 ? 2 : 3, f_sp = (f == 3 || f == 5) ? 1 : 0;
    const int t_in = 16 * f_ti + lc, s_in = 32 * f_sp + 8 * kq;
    bf16x8 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[r] = s_in + r <= t_in ? (bf16_t)d0[r] : (bf16_t)0.f;
      o[4 + r] = s_in + 4 + r <= t_in ? (bf16_t)d1[r] : (bf16_t)0.f;
    }
    *(bf16x8*)(cb + ((((int64_t)b * G + g) * nchunks + c) * CB_FRAGS + f) * 512 + lane * 8) = o;
  }
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
                    int B, int L, int C, int K, int64_t xsb, int64_t xsl, int silu, hipStream_t s,
                    int c_end = -1) {
  constexpr int V = Vec16<T>::N;
  if (c_end < 0) c_end = C;
  dim3 grid((c_end / V + CONV_THREADS - 1) / CONV_THREADS, (L + CONV_TL - 1) / CONV_TL, B);
#define TV_CONVX_CASE(KK)                                                                  \
  case KK:                                                                                 \
    conv1d_xbc_kernel<T, KK><<<grid, CONV_THREADS, 0, s>>>((const T*)x, (const T*)w,       \
        (const T*)bias, (const T*)halo, o, L, C, c_end, xsb, xsl, silu);                   \
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

static int conv1d_xbc_impl(const void* x, const void* weight, const void* bias,
                           const void* halo, void* y_x, void* y_b, void* y_c, void* cb,
                           int batch, int seqlen, int d_inner, int ngroups,
                           int dstate, int kernel, int64_t x_stride_b,
                           int64_t x_stride_l, int dtype, int silu, void* stream, const char* who) {
  TV_CHECK_ARG(weight && (seqlen == 0 || (x && y_x && y_b && y_c)), "%s: null pointer", who);
  TV_CHECK_ARG(batch > 0 && seqlen >= 0 && d_inner > 0 && ngroups > 0 && dstate > 0,
               "%s: bad sizes", who);
  if (kernel < 2 || kernel > 4) TV_UNSUPPORTED("%s: kernel width %d not in [2,4]", who, kernel);
  if (seqlen == 0) return TV_OK;
  const int vec = dtype == TV_F32 ? 4 : 8;
  const int C = d_inner + 2 * ngroups * dstate;
  if (d_inner % vec || dstate % vec || x_stride_l % vec || x_stride_b % vec ||
      ((uintptr_t)x & 15) || ((uintptr_t)y_x & 15) || ((uintptr_t)y_b & 15) || ((uintptr_t)y_c & 15))
    TV_UNSUPPORTED("%s: segments/strides/pointers must be 16-byte aligned", who);
  XbcOut o;
  o.yx = y_x; o.yb = y_b; o.yc = y_c; o.d_inner = d_inner; o.G = ngroups; o.N = dstate;
  hipStream_t s = (hipStream_t)stream;
  if (cb) {
    // the scan's C.B^T fragments ride along: x channels in the streaming kernel, the B / C channels of every
    // (chunk, group) in conv1d_bc_cb_kernel, which multiplies the two tiles while they are in LDS
    if (dtype != TV_BF16 || dstate != CBN || ((uintptr_t)cb & 15))
      TV_UNSUPPORTED("%s: the C.B^T output needs bf16 and d_state %d (got dtype %d, N %d)", who, CBN, dtype, dstate);
    const int rc = launch_conv_xbc<bf16_t>(x, weight, bias, halo, o, batch, seqlen, C, kernel, x_stride_b,
                                           x_stride_l, silu, s, d_inner);
    if (rc != TV_OK) return rc;
    const int nchunks = (seqlen + CBQ - 1) / CBQ;
    const dim3 grid(nchunks, ngroups, batch);
#define TV_CONVCB_CASE(KK)                                                                                  \
  case KK:                                                                                                  \
    conv1d_bc_cb_kernel<KK><<<grid, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)weight,                   \
        (const bf16_t*)bias, (const bf16_t*)halo, (bf16_t*)y_b, (bf16_t*)y_c, (bf16_t*)cb, seqlen, C,       \
        d_inner, ngroups, nchunks, x_stride_b, x_stride_l, silu);                                           \
    break;
    switch (kernel) {
      TV_CONVCB_CASE(2)
      TV_CONVCB_CASE(3)
      TV_CONVCB_CASE(4)
    }
#undef TV_CONVCB_CASE
    TV_LAUNCH_CHECK();
  }
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
  TV_UNSUPPORTED("%s: dtype %d", who, dtype);
}

extern "C" int tv_causal_conv1d_xbc_fwd(const void* x, const void* weight, const void* bias,
                                        const void* halo, void* y_x, void* y_b, void* y_c,
                                        int batch, int seqlen, int d_inner, int ngroups,
                                        int dstate, int kernel, int64_t x_stride_b,
                                        int64_t x_stride_l, int dtype, int silu, void* stream) {
  return conv1d_xbc_impl(x, weight, bias, halo, y_x, y_b, y_c, nullptr, batch, seqlen, d_inner, ngroups, dstate,
                         kernel, x_stride_b, x_stride_l, dtype, silu, stream, "conv1d_xbc");
}

extern "C" size_t tv_ssd_cb_bytes(int batch, int seqlen, int ngroups) {
  if (batch <= 0 || seqlen <= 0 || ngroups <= 0) return 0;
  return (size_t)batch * ngroups * ((seqlen + CBQ - 1) / CBQ) * CB_FRAGS * 512 * sizeof(bf16_t);
}

extern "C" int tv_causal_conv1d_xbc_cb_fwd(const void* x, const void* weight, const void* bias,
                                           const void* halo, void* y_x, void* y_b, void* y_c, void* cb,
                                           int batch, int seqlen, int d_inner, int ngroups,
                                           int dstate, int kernel, int64_t x_stride_b,
                                           int64_t x_stride_l, int dtype, int silu, void* stream) {
  TV_CHECK_ARG(cb || seqlen == 0, "conv1d_xbc_cb: null C.B^T buffer");
  return conv1d_xbc_impl(x, weight, bias, halo, y_x, y_b, y_c, cb, batch, seqlen, d_inner, ngroups, dstate,
                         kernel, x_stride_b, x_stride_l, dtype, silu, stream, "conv1d_xbc_cb");
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
