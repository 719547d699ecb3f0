// S3 (generic path): Mamba-2 selective scan as the exact fp32 token recurrence
//   S_t = exp(dt_t A_h) S_{t-1} + dt_t x_t (x) B_t ;  y_t = S_t . C_t + D_h x_t
// for any dtype / head_dim / d_state (BASELINE config 1: fp32, N=16).  One
// workgroup owns (batch, head, 16 columns of P); a thread owns one column and
// every 16th state index n, keeps its state slice in registers for the whole
// sequence, and the 16 n-lanes of a column reduce y with DPP shuffles.  Token
// tiles are staged through LDS with coalesced loads.  The bf16 Nano-shape fast
// paths are the MFMA marches of ssd_head.hip / ssd_slice.hip; small d_state with a
// workspace goes to the chunk-parallel form in ssd_chunked.hip.
// Reference semantics: modeling_nano.py:639-653, CPU twin :775-851.
#include <type_traits>
#include "common.hpp"

namespace {

constexpr int GS_TT = 32;    // tokens staged per tile
constexpr int GS_PB = 16;    // P columns per workgroup
constexpr int GS_NL = 16;    // n-lanes per column
constexpr int GS_THREADS = GS_PB * GS_NL;

struct SsdArgs {
  const void *x, *dt, *Bm, *Cm;
  const float *A, *D, *dt_bias, *init;
  void* y;
  float *final_state, *total_decay;
  int L, H, P, G, N;
  int64_t xsb, xsl, dsb, dsl, bsb, bsl, bsg, csb, csl, csg, ysb, ysl;
  int softplus, group_map;
  float dt_min, dt_max;
};

template <typename T, int NPT>
__global__ __launch_bounds__(GS_THREADS) void ssd_generic_kernel(SsdArgs a) {
  extern __shared__ float smem[];
  const int N = a.N;
  float* sB = smem;                    // [TT][N]
  float* sC = sB + GS_TT * N;          // [TT][N]
  float* sX = sC + GS_TT * N;          // [TT][PB]
  float* sY = sX + GS_TT * GS_PB;      // [TT][PB]
  float* sDt = sY + GS_TT * GS_PB;     // [TT]
  float* sDa = sDt + GS_TT;            // [TT]

  const int pb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  const int pl = tid / GS_NL, nl = tid % GS_NL;
  const int p = pb * GS_PB + pl;
  const int g = a.group_map ? (h % a.G) : (h / (a.H / a.G));
  const float Ah = a.A[h];
  const float Dh = a.D ? a.D[h] : 0.f;
  const float bias = a.dt_bias ? a.dt_bias[h] : 0.f;

  const T* xb = (const T*)a.x + (int64_t)b * a.xsb + (int64_t)h * a.P;
  const T* dtb = (const T*)a.dt + (int64_t)b * a.dsb + h;
  const T* Bb = (const T*)a.Bm + (int64_t)b * a.bsb + (int64_t)g * a.bsg;
  const T* Cb = (const T*)a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg;
  T* yb = (T*)a.y + (int64_t)b * a.ysb + (int64_t)h * a.P;

  float s[NPT];
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int n = nl + GS_NL * j;
    s[j] = (a.init && p < a.P && n < N)
               ? a.init[(((int64_t)b * a.H + h) * a.P + p) * N + n] : 0.f;
  }
  float decay_sum = 0.f;

  for (int t0 = 0; t0 < a.L; t0 += GS_TT) {
    const int tt = min(GS_TT, a.L - t0);
    __syncthreads();
    for (int i = tid; i < tt * N; i += GS_THREADS) {
      const int t = i / N, n = i % N;
      sB[i] = to_f32(Bb[(int64_t)(t0 + t) * a.bsl + n]);
      sC[i] = to_f32(Cb[(int64_t)(t0 + t) * a.csl + n]);
    }
    for (int i = tid; i < tt * GS_PB; i += GS_THREADS) {
      const int t = i / GS_PB, c = pb * GS_PB + i % GS_PB;
      sX[i] = c < a.P ? to_f32(xb[(int64_t)(t0 + t) * a.xsl + c]) : 0.f;
    }
    if (tid < tt) {
      float d = to_f32(dtb[(int64_t)(t0 + tid) * a.dsl]) + bias;
      if (a.softplus) d = softplus_f(d);
      d = fminf(fmaxf(d, a.dt_min), a.dt_max);
      sDt[tid] = d;
      sDa[tid] = d * Ah;
    }
    __syncthreads();
    for (int t = 0; t < tt; ++t) {
      const float dtv = sDt[t];
      const float da = sDa[t];
      const float dec = expf(da);
      const float xr = sX[t * GS_PB + pl];
      const float xv = dtv * xr;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < NPT; ++j) {
        const int n = nl + GS_NL * j;
        if (n < N) {
          s[j] = fmaf(dec, s[j], xv * sB[t * N + n]);
          acc = fmaf(s[j], sC[t * N + n], acc);
        }
      }
      acc += __shfl_xor(acc, 8, 64);
      acc += __shfl_xor(acc, 4, 64);
      acc += __shfl_xor(acc, 2, 64);
      acc += __shfl_xor(acc, 1, 64);
      if (nl == 0) sY[t * GS_PB + pl] = fmaf(Dh, xr, acc);
      decay_sum += da;
    }
    __syncthreads();
    for (int i = tid; i < tt * GS_PB; i += GS_THREADS) {
      const int t = i / GS_PB, c = pb * GS_PB + i % GS_PB;
      if (c < a.P) yb[(int64_t)(t0 + t) * a.ysl + c] = from_f32<T>(sY[i]);
    }
  }
  if (a.final_state && p < a.P) {
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int n = nl + GS_NL * j;
      if (n < N) a.final_state[(((int64_t)b * a.H + h) * a.P + p) * N + n] = s[j];
    }
  }
  if (a.total_decay && pb == 0 && tid == 0) a.total_decay[(int64_t)b * a.H + h] = decay_sum;
}

template <typename T>
int launch_generic(const SsdArgs& a, int B, hipStream_t st) {
  const int npt = (a.N + GS_NL - 1) / GS_NL;
  dim3 grid((a.P + GS_PB - 1) / GS_PB, a.H, B);
  const size_t lds = (size_t)(2 * GS_TT * a.N + 2 * GS_TT * GS_PB + 2 * GS_TT) * sizeof(float);
#define TV_GS_CASE(NP)                                                                \
  if (npt <= NP) {                                                                    \
    ssd_generic_kernel<T, NP><<<grid, GS_THREADS, lds, st>>>(a);                      \
    TV_LAUNCH_CHECK();                                                                \
  }
  TV_GS_CASE(1)
  TV_GS_CASE(2)
  TV_GS_CASE(4)
  TV_GS_CASE(8)
  TV_GS_CASE(16)
#undef TV_GS_CASE
  TV_UNSUPPORTED("ssd_scan: d_state %d > 256", a.N);
}

// ---------------------------------------------------------------- decode
template <typename T>
__global__ void state_update_kernel(float* __restrict__ state, const T* __restrict__ x,
                                    const T* __restrict__ dt, const float* __restrict__ A,
                                    const T* __restrict__ Bm, const T* __restrict__ Cm,
                                    const float* __restrict__ D, const float* __restrict__ dt_bias,
                                    T* __restrict__ y, int H, int P, int G, int N, int softplus) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  if (p >= P) return;
  const int g = h / (H / G);
  float d = to_f32(dt[(int64_t)b * H + h]) + (dt_bias ? dt_bias[h] : 0.f);
  if (softplus) d = softplus_f(d);
  const float dec = expf(d * A[h]);
  const float xr = to_f32(x[((int64_t)b * H + h) * P + p]);
  const float xv = d * xr;
  float* s = state + (((int64_t)b * H + h) * P + p) * N;
  const T* Br = Bm + ((int64_t)b * G + g) * N;
  const T* Cr = Cm + ((int64_t)b * G + g) * N;
  float acc = 0.f;
  for (int n = 0; n < N; ++n) {
    const float v = fmaf(dec, s[n], xv * to_f32(Br[n]));
    s[n] = v;
    acc = fmaf(v, to_f32(Cr[n]), acc);
  }
  y[((int64_t)b * H + h) * P + p] = from_f32<T>(fmaf(D ? D[h] : 0.f, xr, acc));
}

// Decode step for d_state = 4 LPR (16 .. 256): LPR lanes share a state row (16 bytes = 4 fp32 each), a wave covers
// 64 / LPR rows per load — coalesced 512-byte rows instead of one thread walking its row with 4-byte loads at a 512-byte
// lane stride (36.7 us for the 2 x 5.2 MB of a Nano layer, round 4).  The launch is a latency chain, not a stream (a Nano
// layer is 10 240 rows = 5.2 MB, 20 KB a CU; an empty launch of this grid inside the decode graph takes 1.85 us):
// round 6 keeps ONE state row per lane group in flight per wave (5 120 waves; two and four rows a wave measured 4.7 and
// 5.6 us against 4.3), requests every scalar of the row (dt, dt_bias, A, D, x) and its B / C pieces in front of the state,
// reads and writes the state non-temporally (nothing re-reads it before the next token, 16.6 GB of weights later) and sums
// over d_state on DPP (row_shr inside a 16-lane row, row_bcast across rows: the group's last lane holds the sum) instead
// of five ds_bpermute round trips: 5.5 -> 4.3 us a launch back to back in a graph (devtools/bench_ssu.py), 7.9 -> 5.2 us
// inside the decode step.
template <int LPR>
__device__ __forceinline__ float group_sum_dpp(float v) {      // valid in the last lane of every LPR-lane group
  if (LPR >= 2) v = tv_dpp_add<0x111, 0xf>(v);                  // row_shr:1
  if (LPR >= 4) v = tv_dpp_add<0x112, 0xf>(v);                  // row_shr:2
  if (LPR >= 8) v = tv_dpp_add<0x114, 0xf>(v);                  // row_shr:4
  if (LPR >= 16) v = tv_dpp_add<0x118, 0xf>(v);                 // row_shr:8
  if (LPR >= 32) v = tv_dpp_add<0x142, 0xa>(v);                 // row_bcast:15 -> rows 1, 3
  if (LPR >= 64) v = tv_dpp_add<0x143, 0xc>(v);                 // row_bcast:31 -> rows 2, 3
  return v;
}

template <typename T, int LPR>
__global__ __launch_bounds__(256) void state_update_chain_kernel(float* __restrict__ state, const T* __restrict__ x,
                                                                        const T* __restrict__ dt, const float* __restrict__ A,
                                                                        const T* __restrict__ Bm, const T* __restrict__ Cm,
                                                                        const float* __restrict__ D,
                                                                        const float* __restrict__ dt_bias, T* __restrict__ y,
                                                                        int H, int P, int G, int softplus, int rows_total) {
  constexpr int N = 4 * LPR, RPW = 64 / LPR, IT = 1, WAVES = 4;      // IT: state rows per lane group in flight
  typedef typename std::conditional<std::is_same<T, float>::value, f32x4, typename std::conditional<std::is_same<T, bf16_t>::value, bf16x4, f16x4>::type>::type vec4_t;
  const int lane = threadIdx.x & 63;
  const int w = (int)blockIdx.x * WAVES + (threadIdx.x >> 6);
  const int sub = lane / LPR, ln = lane % LPR;
  const int hpg = H / G;
  int row[IT];
  f32x4 sv[IT];
  float dtv[IT], xr[IT], av[IT], bv[IT], dv[IT];
  vec4_t Bv[IT], Cv[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    row[it] = (w * IT + it) * RPW + sub;                // (b, h, p) flattened
    const int rc = row[it] < rows_total ? row[it] : rows_total - 1;
    const int bh = (int)((unsigned)rc / (unsigned)P);
    const int b = (int)((unsigned)bh / (unsigned)H);
    const int h = bh - b * H;
    const int g = (int)((unsigned)h / (unsigned)hpg);
    sv[it] = __builtin_nontemporal_load((const f32x4*)(state + (int64_t)rc * N) + ln);
    dtv[it] = to_f32(dt[bh]);
    xr[it] = to_f32(x[rc]);
    av[it] = A[h];
    bv[it] = dt_bias ? dt_bias[h] : 0.f;
    dv[it] = D ? D[h] : 0.f;
    Bv[it] = *(const vec4_t*)(Bm + (int64_t)(b * G + g) * N + 4 * ln);
    Cv[it] = *(const vec4_t*)(Cm + (int64_t)(b * G + g) * N + 4 * ln);
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const bool ok = row[it] < rows_total;
    float d = dtv[it] + bv[it];
    if (softplus) d = softplus_f(d);
    const float dec = expf(d * av[it]);
    const float xv = d * xr[it];
    float part = 0.f;
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = fmaf(dec, sv[it][i], xv * to_f32(Bv[it][i]));
      part = fmaf(v[i], to_f32(Cv[it][i]), part);
    }
    if (ok) __builtin_nontemporal_store(v, (f32x4*)(state + (int64_t)row[it] * N) + ln);
    part = group_sum_dpp<LPR>(part);
    if (ok && ln == LPR - 1) y[row[it]] = from_f32<T>(fmaf(dv[it], xr[it], part));
  }
}

}  // namespace

// called from ssd_scan.hip (dispatcher)
int tv_ssd_generic_launch(const void* x, const void* dt, const void* A, const void* Bm,
                          const void* Cm, const void* D, const void* dt_bias,
                          const void* init_state, void* y, void* final_state, void* total_decay,
                          int batch, int seqlen, int nheads, int headdim, int ngroups, int dstate,
                          int64_t xsb, int64_t xsl, int64_t dsb, int64_t dsl, int64_t bsb,
                          int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg, int64_t ysb, int64_t ysl,
                          int dtype, int dt_softplus, float dt_min, float dt_max, int group_map,
                          hipStream_t st) {
  SsdArgs a;
  a.x = x; a.dt = dt; a.Bm = Bm; a.Cm = Cm;
  a.A = (const float*)A; a.D = (const float*)D; a.dt_bias = (const float*)dt_bias;
  a.init = (const float*)init_state; a.y = y; a.final_state = (float*)final_state;
  a.total_decay = (float*)total_decay;
  a.L = seqlen; a.H = nheads; a.P = headdim; a.G = ngroups; a.N = dstate;
  a.xsb = xsb; a.xsl = xsl; a.dsb = dsb; a.dsl = dsl; a.bsb = bsb; a.bsl = bsl; a.bsg = bsg;
  a.csb = csb; a.csl = csl; a.csg = csg; a.ysb = ysb; a.ysl = ysl;
  a.softplus = dt_softplus; a.group_map = group_map; a.dt_min = dt_min; a.dt_max = dt_max;
  switch (dtype) {
    case TV_F32: return launch_generic<float>(a, batch, st);
    case TV_BF16: return launch_generic<bf16_t>(a, batch, st);
    case TV_F16: return launch_generic<f16_t>(a, batch, st);
  }
  TV_UNSUPPORTED("ssd_scan: dtype %d", dtype);
}

extern "C" int tv_selective_state_update(void* state, const void* x, const void* dt,
                                         const void* A, const void* Bm, const void* Cm,
                                         const void* D, const void* dt_bias, void* y, int batch,
                                         int nheads, int headdim, int ngroups, int dstate,
                                         int dtype, int dt_softplus, void* stream) {
  TV_CHECK_ARG(state && x && dt && A && Bm && Cm && y, "selective_state_update: null pointer");
  TV_CHECK_ARG(batch > 0 && nheads > 0 && headdim > 0 && ngroups > 0 && dstate > 0 &&
                   nheads % ngroups == 0,
               "selective_state_update: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  // d_state 16 / 32 / 64 / 128 / 256 with 16-byte aligned rows: lanes share a row (coalesced); anything else: a thread per row
  const int lpr = dstate / 4;
  if (dstate % 4 == 0 && (lpr == 4 || lpr == 8 || lpr == 16 || lpr == 32 || lpr == 64) && (((uintptr_t)state) & 15) == 0 &&
      (((uintptr_t)Bm | (uintptr_t)Cm) & 15) == 0 && (int64_t)batch * nheads * headdim < (1ll << 31)) {
    const int64_t rows = (int64_t)batch * nheads * headdim;
    const int64_t rows_per_block = 4 * (64 / lpr);              // 4 waves x rows per wave-instruction
    const dim3 rgrid((unsigned)((rows + rows_per_block - 1) / rows_per_block));
#define TV_SUR(T, LPR)                                                                                        \
    state_update_chain_kernel<T, LPR><<<rgrid, 256, 0, s>>>((float*)state, (const T*)x, (const T*)dt,         \
        (const float*)A, (const T*)Bm, (const T*)Cm, (const float*)D, (const float*)dt_bias, (T*)y, nheads,   \
        headdim, ngroups, dt_softplus, (int)rows)
#define TV_SUR_T(T)                                                                                           \
    switch (lpr) { case 4: TV_SUR(T, 4); break; case 8: TV_SUR(T, 8); break; case 16: TV_SUR(T, 16); break;     \
                   case 32: TV_SUR(T, 32); break; default: TV_SUR(T, 64); break; }
    switch (dtype) {
      case TV_F32: TV_SUR_T(float); break;
      case TV_BF16: TV_SUR_T(bf16_t); break;
      case TV_F16: TV_SUR_T(f16_t); break;
      default: TV_UNSUPPORTED("selective_state_update: dtype %d", dtype);
    }
#undef TV_SUR_T
#undef TV_SUR
    TV_LAUNCH_CHECK();
  }
  dim3 grid((headdim + 63) / 64, nheads, batch);
#define TV_SU(T)                                                                              \
  state_update_kernel<T><<<grid, 64, 0, s>>>((float*)state, (const T*)x, (const T*)dt,        \
      (const float*)A, (const T*)Bm, (const T*)Cm, (const float*)D, (const float*)dt_bias,    \
      (T*)y, nheads, headdim, ngroups, dstate, dt_softplus)
  switch (dtype) {
    case TV_F32: TV_SU(float); break;
    case TV_BF16: TV_SU(bf16_t); break;
    case TV_F16: TV_SU(f16_t); break;
    default: TV_UNSUPPORTED("selective_state_update: dtype %d", dtype);
  }
#undef TV_SU
  TV_LAUNCH_CHECK();
}
