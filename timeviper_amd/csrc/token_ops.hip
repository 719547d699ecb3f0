// T1/T2: TransV / pdrop token operations — bit-exact integer index work plus
// the HBM-bound row gather and the single-query ranking that replaces the
// reference's (L,L)-mask path.
// Reference: pdrop_no_pack modeling_nano.py:1779-2095.
#include "common.hpp"

namespace {

// ---------------------------------------------------------------- gather
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src,
                                                          const int64_t* __restrict__ index,
                                                          T* __restrict__ dst, int nv,
                                                          int64_t ss, int64_t ds) {
  typedef typename Vec16<T>::type vec_t;
  constexpr int V = Vec16<T>::N;
  const int64_t r = blockIdx.x;
  const int64_t sr = index[r];
  const T* s = src + sr * ss;
  T* d = dst + r * ds;
  for (int iv = threadIdx.x; iv < nv; iv += blockDim.x)
    *(vec_t*)(d + (int64_t)iv * V) = *(const vec_t*)(s + (int64_t)iv * V);
}

// ------------------------------------------------- uniform keep indices
// torch.linspace(0, n-1, keep, dtype=long) as ATen's CPU kernel evaluates it
// (RangeFactoriesKernel.cpp linspace_kernel): step is a double, the first
// half counts up from start, the second half down from end, each value is
// truncated toward zero.  __dmul_rn/__dsub_rn forbid FMA contraction so the
// double arithmetic is bit-identical to the host's mul-then-add.
__global__ void uniform_keep_kernel(int64_t* __restrict__ out, int64_t n, int64_t keep,
                                    int64_t offset) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= keep) return;
  if (keep == 1) {
    out[0] = offset;
    return;
  }
  const double start = 0.0, end = (double)(n - 1);
  const double step = __ddiv_rn(__dsub_rn(end, start), (double)(keep - 1));
  const int64_t half = keep / 2;
  double v;
  if (i < half) v = __dadd_rn(start, __dmul_rn(step, (double)i));
  else v = __dsub_rn(end, __dmul_rn(step, (double)(keep - i - 1)));
  out[i] = (int64_t)v + offset;
}

// ------------------------------------------------------ dropped indices
__global__ void dropped_kernel(const int64_t* __restrict__ keep, int64_t n_keep, int64_t start,
                               int64_t n, int64_t* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int64_t key = start + p;
  int64_t lo = 0, hi = n_keep;  // lower_bound
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keep[mid] < key) lo = mid + 1; else hi = mid;
  }
  if (lo < n_keep && keep[lo] == key) return;  // kept
  out[p - lo] = key;
}

// ----------------------------------------------------------- attn rank
// logits[k][h] = scale * q_h . K[k, g(h)]   (key-major: per-shard pieces concatenate along dim 0)
template <typename T>
__global__ __launch_bounds__(256) void rank_logits_kernel(const T* __restrict__ q,
                                                          const T* __restrict__ k,
                                                          float* __restrict__ logits,
                                                          int n_keys, int Hq, int Hkv, int D,
                                                          int64_t ksl, int64_t ksh,
                                                          float divisor) {
  extern __shared__ float qs[];  // Hq*D
  for (int i = threadIdx.x; i < Hq * D; i += blockDim.x) qs[i] = to_f32(q[i]);
  __syncthreads();
  const int rep = Hq / Hkv;
  // one wave per key; lanes split D
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int key = blockIdx.x * 4 + wave;
  if (key >= n_keys) return;
  for (int g = 0; g < Hkv; ++g) {
    const T* kr = k + (int64_t)key * ksl + (int64_t)g * ksh;
    float kv[4];
    int nd = 0;
    for (int d = lane; d < D && nd < 4; d += 64) kv[nd++] = to_f32(kr[d]);
    for (int r = 0; r < rep; ++r) {
      const int h = g * rep + r;
      float acc = 0.f;
      int j = 0;
      for (int d = lane; d < D && j < 4; d += 64, ++j) acc = fmaf(qs[h * D + d], kv[j], acc);
      acc = wave_sum(acc);
      // the reference forms q.K^T and the "/ math.sqrt(head_dim)" in the activation dtype (:1923-1927) before the
      // fp32 softmax: round twice like it does, and DIVIDE by the fp32 divisor like it does (a product with the
      // reciprocal lands one fp32 ulp off for some values, which the second rounding turns into a bf16 step)
      acc = to_f32(from_f32<T>(acc));
      if (lane == 0) logits[(int64_t)key * Hq + h] = to_f32(from_f32<T>(__fdiv_rn(acc, divisor)));
    }
  }
}

// per head: m = max, s = sum exp(l - m)
__global__ __launch_bounds__(1024) void rank_stats_kernel(const float* __restrict__ logits,
                                                          float* __restrict__ stats,
                                                          int n_keys, int Hq) {
  __shared__ float red[16];
  const int h = blockIdx.x;
  const float* l = logits + h;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n_keys; i += blockDim.x) m = fmaxf(m, l[(int64_t)i * Hq]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < n_keys; i += blockDim.x) s += expf(l[(int64_t)i * Hq] - m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    stats[2 * h] = m;
    stats[2 * h + 1] = t;
  }
}

// scores[j] = mean_h round_T(softmax prob), rounded to T again (the reference
// casts the fp32 softmax to the activation dtype before torch.mean, :1929-1939)
template <typename T>
__global__ void rank_scores_kernel(const float* __restrict__ logits,
                                   const float* __restrict__ stats, float* __restrict__ scores,
                                   int n_keys, int Hq, int vis_start, int n_vis) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_vis) return;
  const int key = vis_start + j;
  float acc = 0.f;
  for (int h = 0; h < Hq; ++h) {
    float p = 0.f;
    if (key < n_keys) p = expf(logits[(int64_t)key * Hq + h] - stats[2 * h]) / stats[2 * h + 1];
    acc += to_f32(from_f32<T>(p));
  }
  scores[j] = to_f32(from_f32<T>(acc / (float)Hq));
}

template <typename T>
int launch_gather(const void* src, const int64_t* index, void* dst, int64_t n, int D,
                  int64_t ss, int64_t ds, hipStream_t s) {
  constexpr int V = Vec16<T>::N;
  const int nv = D / V;
  int threads = ((nv + 63) / 64) * 64;
  if (threads > 256) threads = 256;
  gather_rows_kernel<T><<<dim3((unsigned)n), threads, 0, s>>>((const T*)src, index, (T*)dst, nv,
                                                              ss, ds);
  TV_LAUNCH_CHECK();
}

template <typename T>
int launch_rank_logits(const void* q, const void* k, float* logits, int n_keys, int Hq, int Hkv,
                       int D, int64_t ksl, int64_t ksh, float scale, hipStream_t s) {
  // logits = (q . k) / divisor: the default scale 1 / sqrt(D) becomes the reference's own divisor, float(sqrt(D))
  // (torch divides the bf16 tensor by the Python double rounded to fp32), any other scale its fp32 reciprocal
  const double sq = sqrt((double)D);
  const float divisor = fabs((double)scale * sq - 1.0) < 1e-6 ? (float)sq : (float)(1.0 / (double)scale);
  rank_logits_kernel<T><<<dim3((n_keys + 3) / 4), 256, (size_t)Hq * D * sizeof(float), s>>>(
      (const T*)q, (const T*)k, logits, n_keys, Hq, Hkv, D, ksl, ksh, divisor);
  TV_LAUNCH_CHECK();
}

template <typename T>
int launch_rank_scores(const float* logits, void* scores, int n_keys, int Hq, int vis_start,
                       int n_vis, float* stats, hipStream_t s) {
  rank_stats_kernel<<<dim3(Hq), 1024, 0, s>>>(logits, stats, n_keys, Hq);
  rank_scores_kernel<T><<<dim3((n_vis + 255) / 256), 256, 0, s>>>(logits, stats, (float*)scores,
                                                                   n_keys, Hq, vis_start, n_vis);
  TV_LAUNCH_CHECK();
}

int rank_logits_dispatch(const void* q, const void* k, float* logits, int n_keys, int Hq, int Hkv,
                         int D, int64_t ksl, int64_t ksh, float scale, int dtype, hipStream_t s) {
  switch (dtype) {
    case TV_F32: return launch_rank_logits<float>(q, k, logits, n_keys, Hq, Hkv, D, ksl, ksh, scale, s);
    case TV_BF16: return launch_rank_logits<bf16_t>(q, k, logits, n_keys, Hq, Hkv, D, ksl, ksh, scale, s);
    case TV_F16: return launch_rank_logits<f16_t>(q, k, logits, n_keys, Hq, Hkv, D, ksl, ksh, scale, s);
  }
  TV_UNSUPPORTED("attn_rank: dtype %d", dtype);
}

int rank_scores_dispatch(const float* logits, void* scores, int n_keys, int Hq, int vis_start,
                         int n_vis, float* stats, int dtype, hipStream_t s) {
  switch (dtype) {
    case TV_F32: return launch_rank_scores<float>(logits, scores, n_keys, Hq, vis_start, n_vis, stats, s);
    case TV_BF16: return launch_rank_scores<bf16_t>(logits, scores, n_keys, Hq, vis_start, n_vis, stats, s);
    case TV_F16: return launch_rank_scores<f16_t>(logits, scores, n_keys, Hq, vis_start, n_vis, stats, s);
  }
  TV_UNSUPPORTED("attn_rank: dtype %d", dtype);
}

}  // namespace

extern "C" int tv_gather_rows(const void* src, const int64_t* index, void* dst, int64_t n_rows,
                              int dim, int64_t src_stride, int64_t dst_stride, int dtype,
                              void* stream) {
  TV_CHECK_ARG(n_rows >= 0 && dim > 0, "gather_rows: bad sizes");
  if (n_rows == 0) return TV_OK;
  TV_CHECK_ARG(src && index && dst, "gather_rows: null pointer");
  const int vec = dtype == TV_F32 ? 4 : 8;
  if (dim % vec || src_stride % vec || dst_stride % vec || ((uintptr_t)src & 15) ||
      ((uintptr_t)dst & 15))
    TV_UNSUPPORTED("gather_rows: rows must be 16-byte aligned multiples");
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case TV_F32: return launch_gather<float>(src, index, dst, n_rows, dim, src_stride, dst_stride, s);
    case TV_BF16: return launch_gather<bf16_t>(src, index, dst, n_rows, dim, src_stride, dst_stride, s);
    case TV_F16: return launch_gather<f16_t>(src, index, dst, n_rows, dim, src_stride, dst_stride, s);
  }
  TV_UNSUPPORTED("gather_rows: dtype %d", dtype);
}

extern "C" int tv_uniform_keep_indices(int64_t* out, int64_t n_tokens, int64_t keep,
                                       int64_t offset, void* stream) {
  TV_CHECK_ARG(out, "uniform_keep_indices: null pointer");
  TV_CHECK_ARG(n_tokens >= 1 && keep >= 0, "uniform_keep_indices: bad sizes");
  if (keep == 0) return TV_OK;
  uniform_keep_kernel<<<dim3((unsigned)((keep + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
      out, n_tokens, keep, offset);
  TV_LAUNCH_CHECK();
}

extern "C" int tv_dropped_indices(const int64_t* keep_sorted, int64_t n_keep, int64_t start,
                                  int64_t n, int64_t* out_dropped, void* stream) {
  TV_CHECK_ARG(n >= 0 && n_keep >= 0 && n_keep <= n, "dropped_indices: bad sizes");
  if (n == 0 || n_keep == n) return TV_OK;
  TV_CHECK_ARG(out_dropped && (keep_sorted || n_keep == 0), "dropped_indices: null pointer");
  dropped_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
      keep_sorted, n_keep, start, n, out_dropped);
  TV_LAUNCH_CHECK();
}

extern "C" size_t tv_attn_rank_workspace_bytes(int n_keys, int nheads_q) {
  return ((size_t)nheads_q * (size_t)n_keys + 2 * (size_t)nheads_q) * sizeof(float);
}

extern "C" int tv_attn_rank_logits(const void* q, const void* k, void* logits, int n_keys,
                                   int nheads_q, int nheads_kv, int headdim, int64_t k_stride_l,
                                   int64_t k_stride_h, float scale, int dtype, void* stream) {
  TV_CHECK_ARG(n_keys >= 0 && nheads_q > 0 && nheads_kv > 0 && nheads_q % nheads_kv == 0 && headdim > 0,
               "attn_rank_logits: bad sizes");
  if (n_keys == 0) return TV_OK;
  TV_CHECK_ARG(q && k && logits, "attn_rank_logits: null pointer");
  if (headdim > 256) TV_UNSUPPORTED("attn_rank: headdim %d > 256", headdim);
  return rank_logits_dispatch(q, k, (float*)logits, n_keys, nheads_q, nheads_kv, headdim, k_stride_l,
                              k_stride_h, scale, dtype, (hipStream_t)stream);
}

extern "C" int tv_attn_rank_scores_from_logits(const void* logits, void* scores, int n_keys,
                                               int nheads_q, int vis_start, int n_vis, int dtype,
                                               void* workspace, size_t workspace_bytes,
                                               void* stream) {
  TV_CHECK_ARG(n_keys > 0 && nheads_q > 0 && n_vis >= 0 && vis_start >= 0,
               "attn_rank_scores_from_logits: bad sizes");
  TV_CHECK_ARG(logits && scores && workspace, "attn_rank_scores_from_logits: null pointer");
  if (workspace_bytes < 2 * (size_t)nheads_q * sizeof(float)) {
    tv_set_error("attn_rank_scores_from_logits: workspace too small");
    return TV_ERR_WORKSPACE;
  }
  if (n_vis == 0) return TV_OK;
  return rank_scores_dispatch((const float*)logits, scores, n_keys, nheads_q, vis_start, n_vis,
                              (float*)workspace, dtype, (hipStream_t)stream);
}

extern "C" int tv_attn_rank_scores(const void* q, const void* k, void* scores, int n_keys,
                                   int nheads_q, int nheads_kv, int headdim, int64_t k_stride_l,
                                   int64_t k_stride_h, int vis_start, int n_vis, float scale,
                                   int dtype, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  TV_CHECK_ARG(q && k && scores && workspace, "attn_rank: null pointer");
  TV_CHECK_ARG(n_keys > 0 && nheads_q > 0 && nheads_kv > 0 && nheads_q % nheads_kv == 0 &&
                   headdim > 0 && n_vis >= 0 && vis_start >= 0,
               "attn_rank: bad sizes");
  if (headdim > 256) TV_UNSUPPORTED("attn_rank: headdim %d > 256", headdim);
  if (workspace_bytes < tv_attn_rank_workspace_bytes(n_keys, nheads_q)) {
    tv_set_error("attn_rank: workspace too small");
    return TV_ERR_WORKSPACE;
  }
  if (n_vis == 0) return TV_OK;
  // the two halves a sequence-sharded caller runs separately (logits of its own keys; statistics
  // and scores over the gathered logits): the same kernels, so 1 GPU and N GPUs keep the same tokens
  hipStream_t s = (hipStream_t)stream;
  float* logits = (float*)workspace;
  float* stats = logits + (int64_t)nheads_q * n_keys;
  const int st = rank_logits_dispatch(q, k, logits, n_keys, nheads_q, nheads_kv, headdim, k_stride_l,
                                      k_stride_h, scale, dtype, s);
  if (st != TV_OK) return st;
  return rank_scores_dispatch(logits, scores, n_keys, nheads_q, vis_start, n_vis, stats, dtype, s);
}
