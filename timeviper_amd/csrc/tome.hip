// V3: one round of ToMe bipartite soft matching + size-weighted merge (reference
// timeviper/model/projector/tome.py:14-83: bipartite_soft_matching + merge_wavg), per frame:
//   metric  = head-mean of x (C/heads dims), L2-normalised
//   even i  -> best odd j = argmax_j <metric[2i], metric[2j+1]>
//   the r even tokens with the largest best score are merged into their odd match, the
//   other evens are kept (in descending-score order, as the reference's gather does)
//   x'  = [kept evens | odds (+ merged evens)], size-weighted average;  size' = summed sizes
// Four kernels per round, all frames at once (the reference path is ~12 torch launches per
// round):
//   tome_metric_kernel  one wave per token: head-mean + norm, fp32
//   tome_match_kernel   64 evens x all odds per workgroup, 4x4 register tiles from LDS
//   tome_sort_kernel    one workgroup per frame: bitonic sort of (score desc, index asc)
//   tome_merge_kernel   one wave per OUTPUT row: gathers its sources in index order
// Arithmetic is fp32 throughout (the reference rounds to the activation dtype after every
// step); sources are accumulated in a fixed order, so results are deterministic.
#include "common.hpp"

namespace {

constexpr int TM_THREADS = 256;
constexpr int TM_MAXD = 96;       // head-mean dims (C / heads) supported (SigLIP 72, DINOv2 64, InternVideo2 88)
constexpr int TM_SORT = 512;      // max evens per frame (sort network width)

template <typename T>
__global__ __launch_bounds__(TM_THREADS) void tome_metric_kernel(
    const T* __restrict__ x, float* __restrict__ metric, int64_t tokens, int C, int heads) {
  const int64_t tok = (int64_t)blockIdx.x * (TM_THREADS / 64) + (threadIdx.x >> 6);
  if (tok >= tokens) return;
  const int lane = threadIdx.x & 63;
  const int dh = C / heads;
  const T* xr = x + tok * C;
  float m[2] = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int d = lane + 64 * k;
    if (d < dh) {
      float acc = 0.f;
      for (int h = 0; h < heads; ++h) acc += to_f32(xr[h * dh + d]);
      m[k] = acc / (float)heads;
    }
  }
  const float ss = wave_sum(m[0] * m[0] + m[1] * m[1]);
  const float inv = 1.f / sqrtf(ss);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int d = lane + 64 * k;
    if (d < dh) metric[tok * dh + d] = m[k] * inv;
  }
}

// grid (ceil(ne/64), frames).  best_val / best_idx (frames, ne)
__global__ __launch_bounds__(TM_THREADS) void tome_match_kernel(
    const float* __restrict__ metric, float* __restrict__ best_val, int* __restrict__ best_idx,
    int T, int dh) {
  __shared__ float sE[64][TM_MAXD + 1], sO[64][TM_MAXD + 1];
  const int ne = (T + 1) / 2, no = T / 2;
  const int f = blockIdx.y, e0 = blockIdx.x * 64;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const float* mf = metric + (int64_t)f * T * dh;
  for (int i = tid; i < 64 * dh; i += TM_THREADS) {
    const int r = i / dh, d = i - r * dh;
    sE[r][d] = (e0 + r < ne) ? mf[(int64_t)(2 * (e0 + r)) * dh + d] : 0.f;
  }
  float bv[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {0, 0, 0, 0};
  for (int o0 = 0; o0 < no; o0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * dh; i += TM_THREADS) {
      const int r = i / dh, d = i - r * dh;
      sO[r][d] = (o0 + r < no) ? mf[(int64_t)(2 * (o0 + r) + 1) * dh + d] : 0.f;
    }
    __syncthreads();
    float acc[4][4] = {};
    for (int d = 0; d < dh; ++d) {
      float av[4], bw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = sE[4 * ty + i][d]; bw[i] = sO[tx + 16 * i][d]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bw[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = o0 + tx + 16 * j;           // columns visited in increasing order per lane
        if (o < no && acc[i][j] > bv[i]) { bv[i] = acc[i][j]; bi[i] = o; }
      }
  }
  // reduce over the 16 lanes (tx) that share a row: larger value wins, ties -> lower index
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv[i], off, 64);
      const int oi = __shfl_xor(bi[i], off, 64);
      if (ov > bv[i] || (ov == bv[i] && oi < bi[i])) { bv[i] = ov; bi[i] = oi; }
    }
    const int e = e0 + 4 * ty + i;
    if (tx == 0 && e < ne) {
      best_val[(int64_t)f * ne + e] = bv[i];
      best_idx[(int64_t)f * ne + e] = bi[i];
    }
  }
}

// one workgroup per frame: order[k] = even index with the k-th largest best score
__global__ __launch_bounds__(TM_SORT) void tome_sort_kernel(const float* __restrict__ best_val,
                                                            int* __restrict__ order, int ne) {
  __shared__ float sv[TM_SORT];
  __shared__ int si[TM_SORT];
  const int f = blockIdx.x, tid = threadIdx.x;
  sv[tid] = tid < ne ? best_val[(int64_t)f * ne + tid] : -INFINITY;
  si[tid] = tid < ne ? tid : (1 << 30);
  __syncthreads();
  for (int k = 2; k <= TM_SORT; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int p = tid ^ j;
      if (p > tid) {
        const float a = sv[tid], b = sv[p];
        const int ia = si[tid], ib = si[p];
        // "a before b": larger score first, ties by lower index
        const bool a_first = a > b || (a == b && ia < ib);
        const bool up = (tid & k) == 0;
        if (up ? !a_first : a_first) { sv[tid] = b; sv[p] = a; si[tid] = ib; si[p] = ia; }
      }
      __syncthreads();
    }
  if (tid < ne) order[(int64_t)f * ne + tid] = si[tid];
}

// one wave per output row.  rows [0, ne-r): kept evens order[r + row]; rows [ne-r, ne-r+no):
// odd j plus every merged even whose match is j (sources taken in sort order)
template <typename T>
__global__ __launch_bounds__(TM_THREADS) void tome_merge_kernel(
    const T* __restrict__ x, const T* __restrict__ size_in, const int* __restrict__ order,
    const int* __restrict__ best_idx, T* __restrict__ x_out, T* __restrict__ size_out, int frames,
    int T_, int C, int r) {
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  const int ne = (T_ + 1) / 2, no = T_ / 2, To = T_ - r;
  const int64_t wid = (int64_t)blockIdx.x * (TM_THREADS / 64) + (threadIdx.x >> 6);
  if (wid >= (int64_t)frames * To) return;
  const int f = (int)(wid / To), row = (int)(wid % To);
  const int lane = threadIdx.x & 63;
  const T* xf = x + (int64_t)f * T_ * C;
  const T* sf = size_in ? size_in + (int64_t)f * T_ : nullptr;
  const int* of = order + (int64_t)f * ne;
  const int* bf = best_idx + (int64_t)f * ne;
  const int nv = C / V;                      // 16-byte vectors per row; lane handles lane, lane+64, ...
  constexpr int MAXV = 4;                    // C <= 64 * 4 * V
  float acc[MAXV][V];
#pragma unroll
  for (int k = 0; k < MAXV; ++k)
#pragma unroll
    for (int i = 0; i < V; ++i) acc[k][i] = 0.f;
  float stot = 0.f;
  auto add_token = [&](int tok) {
    const float s = sf ? to_f32(sf[tok]) : 1.f;
    stot += s;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int iv = lane + 64 * k;
      if (iv < nv) {
        const vec_t v = *(const vec_t*)(xf + (int64_t)tok * C + (int64_t)iv * V);
#pragma unroll
        for (int i = 0; i < V; ++i) acc[k][i] = fmaf(to_f32(v[i]), s, acc[k][i]);
      }
    }
  };
  if (row < ne - r) {
    add_token(2 * of[r + row]);
  } else {
    const int j = row - (ne - r);
    add_token(2 * j + 1);
    for (int k0 = 0; k0 < r; k0 += 64) {      // scan the merged evens, 64 at a time
      const int k = k0 + lane;
      const int src = k < r ? of[k] : -1;
      const bool hit = k < r && bf[src] == j;
      unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
      while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        add_token(2 * __builtin_amdgcn_readlane(src, l));
      }
    }
    (void)no;
  }
  const float inv = 1.f / stot;
  T* xo = x_out + ((int64_t)f * To + row) * C;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int iv = lane + 64 * k;
    if (iv < nv) {
      vec_t o;
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = from_f32<T>(acc[k][i] * inv);
      *(vec_t*)(xo + (int64_t)iv * V) = o;
    }
  }
  if (lane == 0) size_out[(int64_t)f * To + row] = from_f32<T>(stot);
}

size_t ws_layout(int frames, int T, int dh, size_t* o_best, size_t* o_idx, size_t* o_order) {
  const size_t ne = (size_t)(T + 1) / 2;
  size_t off = (size_t)frames * T * dh * sizeof(float);
  off = (off + 255) & ~(size_t)255;
  *o_best = off;
  off += (size_t)frames * ne * sizeof(float);
  off = (off + 255) & ~(size_t)255;
  *o_idx = off;
  off += (size_t)frames * ne * sizeof(int);
  off = (off + 255) & ~(size_t)255;
  *o_order = off;
  off += (size_t)frames * ne * sizeof(int);
  return (off + 255) & ~(size_t)255;
}

template <typename T>
int run_round(const void* x, const void* size_in, void* x_out, void* size_out, int frames, int T_,
              int C, int heads, int r, void* ws, hipStream_t s) {
  const int dh = C / heads, ne = (T_ + 1) / 2;
  size_t o_best, o_idx, o_order;
  ws_layout(frames, T_, dh, &o_best, &o_idx, &o_order);
  float* metric = (float*)ws;
  float* best_val = (float*)((char*)ws + o_best);
  int* best_idx = (int*)((char*)ws + o_idx);
  int* order = (int*)((char*)ws + o_order);
  const int64_t tokens = (int64_t)frames * T_;
  tome_metric_kernel<T><<<dim3((unsigned)((tokens + 3) / 4)), TM_THREADS, 0, s>>>(
      (const T*)x, metric, tokens, C, heads);
  tome_match_kernel<<<dim3((ne + 63) / 64, frames), TM_THREADS, 0, s>>>(metric, best_val, best_idx,
                                                                         T_, dh);
  tome_sort_kernel<<<dim3(frames), TM_SORT, 0, s>>>(best_val, order, ne);
  const int64_t rows = (int64_t)frames * (T_ - r);
  tome_merge_kernel<T><<<dim3((unsigned)((rows + 3) / 4)), TM_THREADS, 0, s>>>(
      (const T*)x, (const T*)size_in, order, best_idx, (T*)x_out, (T*)size_out, frames, T_, C, r);
  TV_LAUNCH_CHECK();
}

}  // namespace

extern "C" size_t tv_tome_workspace_bytes(int frames, int tokens, int dim, int heads) {
  if (frames <= 0 || tokens <= 0 || dim <= 0 || heads <= 0 || dim % heads) return 0;
  size_t a, b, c;
  return ws_layout(frames, tokens, dim / heads, &a, &b, &c);
}

extern "C" int tv_tome_merge_round(const void* x, const void* size_in, void* x_out, void* size_out,
                                   int frames, int tokens, int dim, int heads, int r, int dtype,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  TV_CHECK_ARG(frames >= 0 && tokens >= 2 && dim > 0 && heads > 0 && dim % heads == 0,
               "tome: bad sizes (frames %d tokens %d dim %d heads %d)", frames, tokens, dim, heads);
  TV_CHECK_ARG(r > 0 && r <= tokens / 2, "tome: r = %d must be in [1, tokens/2]", r);
  if (frames == 0) return TV_OK;
  TV_CHECK_ARG(x && x_out && size_out && workspace, "tome: null pointer");
  const int vec = dtype == TV_F32 ? 4 : 8;
  if (dim % vec || dim / vec > 64 * 4) TV_UNSUPPORTED("tome: dim %d not a multiple of %d or too wide", dim, vec);
  if (dim / heads > TM_MAXD) TV_UNSUPPORTED("tome: %d dims per head > %d", dim / heads, TM_MAXD);
  if ((tokens + 1) / 2 > TM_SORT) TV_UNSUPPORTED("tome: %d tokens per frame > %d", tokens, 2 * TM_SORT);
  if ((((uintptr_t)x) | ((uintptr_t)x_out) | ((uintptr_t)workspace)) & 15)
    TV_UNSUPPORTED("tome: pointers must be 16-byte aligned");
  TV_CHECK_ARG(workspace_bytes >= tv_tome_workspace_bytes(frames, tokens, dim, heads),
               "tome: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case TV_F32: return run_round<float>(x, size_in, x_out, size_out, frames, tokens, dim, heads, r, workspace, s);
    case TV_BF16: return run_round<bf16_t>(x, size_in, x_out, size_out, frames, tokens, dim, heads, r, workspace, s);
    case TV_F16: return run_round<f16_t>(x, size_in, x_out, size_out, frames, tokens, dim, heads, r, workspace, s);
    default: TV_UNSUPPORTED("tome: dtype %d", dtype);
  }
}
