// S3 (generic path, chunk-parallel): the selective scan for shapes the MFMA marches do not take — any dtype, small d_state
// (BASELINE config 1: fp32, 32 heads x 64, d_state 16, L = 1 024) — in the CHUNKED form the reference's CPU path states
// (NemotronHMamba2Mixer.torch_forward, modeling_nano.py:775-851: chunk-local decays :792-806, diagonal blocks Y_diag
// :808-818, per-chunk states :820-824, the recurrence over chunk states :826-832, off-diagonal Y_off :833-836) instead of
// ssd_generic.hip's token recurrence: that kernel walks L dependent steps per head (1 024 x ~0.3 us = 331 us for 0.3 MB of
// data, 60 launch latencies), this one has no dependency longer than the number of chunks.
//
//   launch 1, work-group = (chunk of 64 tokens, head, batch):  cs_t = inclusive sum of dt_t A_h inside the chunk;
//       M[t][s] = [s <= t] exp(cs_t - cs_s) dt_s (C_t . B_s);   y_t = sum_s M[t][s] x_s + D x_t   (Y_diag);
//       S_c = sum_s exp(cs_end - cs_s) dt_s x_s (x) B_s;        workspace <- S_c, cs_end
//   launch 2, same grid:  S_in(c) = the recurrence over the chunks before c (S <- exp(cs_end_j) S + S_j, j < c: each
//       work-group folds its own predecessors — at most 63 states of P x N floats from L2 — so no chain of launches or
//       work-groups exists);  y_t += exp(cs_t) C_t . S_in(c)   (Y_off);  the last chunk writes the final state and the
//       total decay.
// fp32 arithmetic throughout (every exponent is <= 0: decays only), y in the caller's dtype (a bf16 / f16 y is rounded once
// per launch: the generic path's tolerance).  Takes d_state <= 64, head_dim <= 128, 2 .. 64 chunks; everything else stays on
// the token recurrence.
#include "common.hpp"

namespace {

constexpr int CQ = 64;          // tokens per chunk
constexpr int CH_THREADS = 256;
constexpr int CH_MAXCHUNKS = 64;

struct ChArgs {
  const void *x, *dt, *Bm, *Cm;
  const float *A, *D, *dt_bias, *init;
  void* y;
  float *final_state, *total_decay;
  float* ws;                     // [B][H][nch][P * N] chunk states, then [B][H][nch] chunk decays
  int L, H, P, G, N, nch;
  int64_t xsb, xsl, dsb, dsl, bsb, bsl, bsg, csb, csl, csg, ysb, ysl;
  int softplus, group_map;
  float dt_min, dt_max;
};

__device__ __forceinline__ float* ws_state(const ChArgs& a, int b, int h, int c) {
  return a.ws + (((int64_t)b * a.H + h) * a.nch + c) * ((int64_t)a.P * a.N);
}
__device__ __forceinline__ float* ws_decay(const ChArgs& a, int b, int h) {
  return a.ws + (int64_t)gridDim.z * a.H * a.nch * ((int64_t)a.P * a.N) + ((int64_t)b * a.H + h) * a.nch;
}

// dt of the chunk -> d_t (softplus, clamp), cs_t (inclusive sum of d_t A) in LDS; tokens past the end: d = 0
template <typename T>
__device__ __forceinline__ void chunk_decays(const ChArgs& a, int b, int h, int t0, int tt, float* sD, float* sCs) {
  if (threadIdx.x < 64) {
    const int t = threadIdx.x;
    float d = 0.f;
    if (t < tt) {
      d = to_f32(((const T*)a.dt)[(int64_t)b * a.dsb + (int64_t)(t0 + t) * a.dsl + h]) + (a.dt_bias ? a.dt_bias[h] : 0.f);
      if (a.softplus) d = softplus_f(d);
      d = fminf(fmaxf(d, a.dt_min), a.dt_max);
    }
    sD[t] = d;
    sCs[t] = wave_incl_scan(d * a.A[h]);
  }
}

template <typename T>
__global__ __launch_bounds__(CH_THREADS) void ssd_chunk_local_kernel(ChArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = a.N, P = a.P, PP = (P + 3) & ~3, NS = N | 1;      // (odd row stride of B / C: a wave reads 64 rows at one n)
  float* sX = smem;                    // [64][PP]
  float* sB = sX + CQ * PP;            // [64][NS]
  float* sC = sB + CQ * NS;            // [64][NS]
  float* sM = sC + CQ * NS;            // [64][65]
  float* sD = sM + CQ * 65;            // [64]
  float* sCs = sD + CQ;                // [64]
  float* sW = sCs + CQ;                // [64]: exp(cs_end - cs_s) dt_s
  // two work-groups per chunk: the even one computes M and Y_diag, the odd one the chunk's state (both need x and B; the
  // problem is too small to fill the chip otherwise: 512 work-groups of ~10 us each)
  const int c = blockIdx.x >> 1, part = blockIdx.x & 1, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int t0 = c * CQ, tt = min(CQ, a.L - t0);
  const int g = a.group_map ? (h % a.G) : (h / (a.H / a.G));
  const T* xb = (const T*)a.x + (int64_t)b * a.xsb + (int64_t)h * P;
  const T* Bb = (const T*)a.Bm + (int64_t)b * a.bsb + (int64_t)g * a.bsg;
  const T* Cb = (const T*)a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg;
  chunk_decays<T>(a, b, h, t0, tt, sD, sCs);
  // Staging.  Wave w takes the rows w, w + 4, ..., lane = column; eight rows' loads are issued before the first LDS store
  // (clamped addresses + a select instead of a branch around the load, no integer division by run-time sizes: written as
  // one element per loop pass, hipcc waited for every load before issuing the next — 20 round trips a work-group)
  {
    const int w = tid >> 6, lane = tid & 63;
    const int nc = min(lane, N - 1);
    for (int kb = 0; kb < CQ / 4; kb += 8) {
      float bv[8], cv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int tc = min(w + 4 * (kb + k), tt - 1);
        bv[k] = to_f32(Bb[(int64_t)(t0 + tc) * a.bsl + nc]);
        cv[k] = to_f32(Cb[(int64_t)(t0 + tc) * a.csl + nc]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int t = w + 4 * (kb + k);
        if (lane < N) {
          sB[t * NS + lane] = t < tt ? bv[k] : 0.f;
          sC[t * NS + lane] = t < tt ? cv[k] : 0.f;
        }
      }
    }
    for (int p0 = 0; p0 < PP; p0 += 64) {
      const int p = p0 + lane, pc = min(p, P - 1);
      for (int kb = 0; kb < CQ / 4; kb += 8) {
        float xv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = to_f32(xb[(int64_t)(t0 + min(w + 4 * (kb + k), tt - 1)) * a.xsl + pc]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int t = w + 4 * (kb + k);
          if (p < PP) sX[t * PP + p] = (t < tt && p < P) ? xv[k] : 0.f;
        }
      }
    }
  }
  __syncthreads();
  if (part == 1) {
    // the chunk's state from a zero start, and its decay
    if (tid < CQ) sW[tid] = __expf(sCs[CQ - 1] - sCs[tid]) * sD[tid];
    __syncthreads();
    const float cend = sCs[CQ - 1];                        // (tokens past the end add 0: cs stays at the last real value)
    float* So = ws_state(a, b, h, c);
    for (int idx = tid; idx < P * N; idx += CH_THREADS) {
      const int p = idx / N, n = idx - p * N;
      float acc = 0.f;
#pragma unroll 8
      for (int s = 0; s < tt; ++s) acc = fmaf(sW[s] * sX[s * PP + p], sB[s * NS + n], acc);
      So[idx] = acc;
    }
    if (tid == 0) ws_decay(a, b, h)[c] = cend;
    return;
  }
  // M[t][s]: causal, decayed, dt-weighted C.B^T of the chunk
  for (int i = tid; i < CQ * CQ; i += CH_THREADS) {
    const int t = i >> 6, s = i & 63;
    float v = 0.f;
    if (s <= t) {
      float dot = 0.f;
      for (int n = 0; n < N; ++n) dot = fmaf(sC[t * NS + n], sB[s * NS + n], dot);
      v = __expf(sCs[t] - sCs[s]) * sD[s] * dot;
    }
    sM[t * 65 + s] = v;
  }
  __syncthreads();
  // Y_diag: 4 x 4 (token, column) tiles
  {
    const float Dh = a.D ? a.D[h] : 0.f;
    T* yb = (T*)a.y + (int64_t)b * a.ysb + (int64_t)h * P;
    const int ptiles = PP >> 2;
    for (int tile = tid; tile < 16 * ptiles; tile += CH_THREADS) {
      const int tq = tile / ptiles, pq = tile - tq * ptiles;
      const int tb = 4 * tq, pb = 4 * pq;
      float acc[4][4] = {};
      const int send = min(tb + 3, tt - 1);
      for (int s = 0; s <= send; ++s) {
        const f32x4 xv = *(const f32x4*)(sX + s * PP + pb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float m = sM[(tb + i) * 65 + s];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(m, xv[j], acc[i][j]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = tb + i;
        if (t >= tt) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int p = pb + j;
          if (p < P) yb[(int64_t)(t0 + t) * a.ysl + p] = from_f32<T>(fmaf(Dh, sX[t * PP + p], acc[i][j]));
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(CH_THREADS) void ssd_chunk_carry_kernel(ChArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = a.N, P = a.P;
  float* sS = smem;                    // [P][N + 1]
  float* sC = sS + P * (N + 1);        // [64][N]
  float* sD = sC + CQ * N;             // [64]
  float* sCs = sD + CQ;                // [64]
  float* sE = sCs + CQ;                // [CH_MAXCHUNKS]: exp(decay of chunk j)
  // two work-groups per chunk, each with half of the head's columns [p_lo, p_hi)
  const int c = blockIdx.x >> 1, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int p_lo = (blockIdx.x & 1) ? (P + 1) / 2 : 0, p_hi = (blockIdx.x & 1) ? P : (P + 1) / 2, PH = p_hi - p_lo;
  const int t0 = c * CQ, tt = min(CQ, a.L - t0);
  const int g = a.group_map ? (h % a.G) : (h / (a.H / a.G));
  const T* Cb = (const T*)a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg;
  const float* dec = ws_decay(a, b, h);
  const bool last = c == a.nch - 1;
  if (c == 0 && !a.init && !last) return;                 // nothing enters the first chunk
  chunk_decays<T>(a, b, h, t0, tt, sD, sCs);
  {
    const int w = tid >> 6, lane = tid & 63;
    const int nc = min(lane, N - 1);
    for (int kb = 0; kb < CQ / 4; kb += 8) {
      float cv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) cv[k] = to_f32(Cb[(int64_t)(t0 + min(w + 4 * (kb + k), tt - 1)) * a.csl + nc]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int t = w + 4 * (kb + k);
        if (lane < N) sC[t * N + lane] = t < tt ? cv[k] : 0.f;
      }
    }
  }
  // the state that enters chunk c: the recurrence over the chunks before it
  const float* S0 = ws_state(a, b, h, 0);
  const int64_t pn = (int64_t)P * N;
  const float dlast = dec[c];
  float dsum = 0.f;
  if (last && tid == 0 && p_lo == 0) for (int j = 0; j < a.nch; ++j) dsum += dec[j];
  if (tid < a.nch) sE[tid] = __expf(dec[tid]);
  __syncthreads();
  // (four state elements per thread and pass, the loop over the predecessors unrolled: sixteen independent loads in flight
  // instead of one L2 round trip per predecessor and element)
  for (int base = p_lo * N; base < p_hi * N; base += 4 * CH_THREADS) {
    int idx[4];
    float acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      idx[k] = min(base + tid + k * CH_THREADS, p_hi * N - 1);
      acc[k] = a.init ? a.init[(((int64_t)b * a.H + h) * P) * N + idx[k]] : 0.f;
    }
    int j = 0;
    for (; j + 4 <= c; j += 4) {                   // sixteen loads in flight
      float v[4][4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[jj][k] = S0[(int64_t)(j + jj) * pn + idx[k]];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fmaf(sE[j + jj], acc[k], v[jj][k]);
    }
    for (; j < c; ++j) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = S0[(int64_t)j * pn + idx[k]];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = fmaf(sE[j], acc[k], v[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (base + tid + k * CH_THREADS >= p_hi * N) continue;
      const int p = idx[k] / N, n = idx[k] - p * N;
      sS[p * (N + 1) + n] = acc[k];
      if (last && a.final_state)
        a.final_state[(((int64_t)b * a.H + h) * P) * N + idx[k]] = fmaf(__expf(dlast), acc[k], S0[(int64_t)c * pn + idx[k]]);
    }
  }
  if (last && a.total_decay && tid == 0 && p_lo == 0) a.total_decay[(int64_t)b * a.H + h] = dsum;
  if (c == 0 && !a.init) return;                           // (the last chunk of a one-chunk... unreachable: nch >= 2)
  __syncthreads();
  // Y_off: y_t += exp(cs_t) C_t . S_in
  T* yb = (T*)a.y + (int64_t)b * a.ysb + (int64_t)h * P;
  // wave w takes the rows w, w + 4, ..., lane = column of this work-group's half; eight rows' old values are loaded together
  {
    const int w = tid >> 6, lane = tid & 63;
    const int p = p_lo + min(lane, PH - 1);
    for (int kb = 0; kb < CQ / 4; kb += 8) {
      float yo[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) yo[k] = to_f32(yb[(int64_t)(t0 + min(w + 4 * (kb + k), tt - 1)) * a.ysl + p]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int t = w + 4 * (kb + k);
        float s_ = 0.f;
        for (int n = 0; n < N; ++n) s_ = fmaf(sC[t * N + n], sS[p * (N + 1) + n], s_);
        if (t < tt && lane < PH) yb[(int64_t)(t0 + t) * a.ysl + p] = from_f32<T>(fmaf(__expf(sCs[t]), s_, yo[k]));
      }
    }
  }
}

template <typename T>
int launch_chunked(const ChArgs& a, int B, hipStream_t st) {
  const dim3 grid((unsigned)(2 * a.nch), (unsigned)a.H, (unsigned)B);
  const int PP = (a.P + 3) & ~3;
  const size_t lds1 = (size_t)(CQ * PP + 2 * CQ * (a.N | 1) + CQ * 65 + 3 * CQ) * sizeof(float);
  const size_t lds2 = (size_t)(a.P * (a.N + 1) + CQ * a.N + 2 * CQ + CH_MAXCHUNKS) * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)ssd_chunk_local_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)ssd_chunk_carry_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
  if (e != hipSuccess) {
    tv_set_error("ssd_scan (chunked): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return TV_ERR_LAUNCH;
  }
  ssd_chunk_local_kernel<T><<<grid, CH_THREADS, lds1, st>>>(a);
  ssd_chunk_carry_kernel<T><<<grid, CH_THREADS, lds2, st>>>(a);
  TV_LAUNCH_CHECK();
}

}  // namespace

// called from ssd_scan.hip (dispatcher)
bool tv_ssd_chunked_supported(int seqlen, int headdim, int dstate) {
  const int nch = (seqlen + CQ - 1) / CQ;
  return dstate <= 64 && headdim <= 128 && nch >= 2 && nch <= CH_MAXCHUNKS;
}
size_t tv_ssd_chunked_workspace_bytes(int batch, int seqlen, int nheads, int headdim, int dstate) {
  if (!tv_ssd_chunked_supported(seqlen, headdim, dstate)) return 0;
  const size_t nch = (size_t)(seqlen + CQ - 1) / CQ;
  return (size_t)batch * nheads * nch * ((size_t)headdim * dstate + 1) * sizeof(float);
}
int tv_ssd_chunked_launch(const void* x, const void* dt, const void* A, const void* Bm, const void* Cm, const void* D,
                          const void* dt_bias, const void* init_state, void* y, void* final_state, void* total_decay,
                          int batch, int seqlen, int nheads, int headdim, int ngroups, int dstate, int64_t xsb, int64_t xsl,
                          int64_t dsb, int64_t dsl, int64_t bsb, int64_t bsl, int64_t bsg, int64_t csb, int64_t csl,
                          int64_t csg, int64_t ysb, int64_t ysl, int dtype, int dt_softplus, float dt_min, float dt_max,
                          int group_map, void* workspace, hipStream_t st) {
  ChArgs a;
  a.x = x; a.dt = dt; a.Bm = Bm; a.Cm = Cm;
  a.A = (const float*)A; a.D = (const float*)D; a.dt_bias = (const float*)dt_bias;
  a.init = (const float*)init_state; a.y = y; a.final_state = (float*)final_state;
  a.total_decay = (float*)total_decay;
  a.ws = (float*)workspace;
  a.L = seqlen; a.H = nheads; a.P = headdim; a.G = ngroups; a.N = dstate; a.nch = (seqlen + CQ - 1) / CQ;
  a.xsb = xsb; a.xsl = xsl; a.dsb = dsb; a.dsl = dsl; a.bsb = bsb; a.bsl = bsl; a.bsg = bsg;
  a.csb = csb; a.csl = csl; a.csg = csg; a.ysb = ysb; a.ysl = ysl;
  a.softplus = dt_softplus; a.group_map = group_map; a.dt_min = dt_min; a.dt_max = dt_max;
  switch (dtype) {
    case TV_F32: return launch_chunked<float>(a, batch, st);
    case TV_BF16: return launch_chunked<bf16_t>(a, batch, st);
    case TV_F16: return launch_chunked<f16_t>(a, batch, st);
  }
  TV_UNSUPPORTED("ssd_scan: dtype %d", dtype);
}
