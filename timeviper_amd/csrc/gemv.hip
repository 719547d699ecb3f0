// Decode-step linear layers: y[m][n] = sum_k f(x)[m][k] W[n][k] (+ bias[n]) for M <= 4 rows (one token per sequence).
//
// A generated token moves every weight of the stack once (16.6 GB for Nemotron-Nano-9B-v2) through matrix-VECTOR
// products: HBM-bound, 2 flops a byte.  The library's skinny-GEMM kernels reach 2.3 - 3.4 TB/s on these shapes
// (bench.py --config decode, rocprofv3), and every product is surrounded by single-row kernels — RMSNorm + residual add
// in front of in_proj / up_proj, relu^2 in front of down_proj, the gated group norm in front of out_proj — that take
// 7 - 10 us each for a few KiB.  This kernel streams W with 16-byte loads (1 KiB per load instruction, 12 - 16 of them in
// flight per wave) against f(x) held in LDS as bf16.  A work-group owns whole rows of W (b, b + grid, ...) and its four
// waves split K — with N = 4 480 rows (out_proj, down_proj) a wave per row leaves the 2 048 waves of the chip 2 or 3
// rows each (73 % balance), a work-group per row 8 or 9 (97 %); the four partial sums meet in LDS.  f — the single-row
// operator in front of the product — is computed in the prologue, redundantly in every work-group (x is a few KiB in L2):
//   PRO_NONE      f(x) = x
//   PRO_RMSNORM   s = bf16(x + delta) (written to `sum_out` by work-group 0: the new residual stream),
//                 f = bf16(w * (s * rsqrt(mean(s^2) + eps)))             NemotronHRMSNorm + block add, modeling_nano.py:897-903, :966
//   PRO_RELU2     f = bf16(relu(x)^2)                                    NemotronHMLP, :993-994
//   PRO_GATED     v = x * silu(z), f = bf16(w * (v * rsqrt(mean_group(v^2) + eps)))     MambaRMSNormGated, :371-380
// with the rounding points of the stand-alone kernels (norms.hip), so the fused step and the unfused one agree to the
// accumulation order of the dot products.  fp32 accumulation (v_dot2c_f32_bf16), one rounding of y.
#include "ssd_common.hpp"

namespace {

enum { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_RELU2 = 2, PRO_GATED = 3 };
constexpr int GV_THREADS = 256;
constexpr int GV_WAVES = GV_THREADS / 64;
constexpr int GV_MAXM = 4;

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct GemvArgs {
  const bf16_t *x, *W, *bias, *delta, *gate;
  const void* norm_w;
  bf16_t *y, *sum_out;
  int M, N, K, group, norm_w_f32;
  int64_t xs, ldw, ys, ds, ss, gs;
  float eps;
  // epilogue (row-per-wave kernel): rows [conv_lo, conv_hi) of y pass through a width-4 causal-conv update + SiLU
  bf16_t* conv_state;              // (M, conv_hi - conv_lo, 4): shifted left, the new value appended
  const bf16_t *conv_w, *conv_b;   // (channels, 4), (channels) or NULL
  int conv_lo, conv_hi;
};

// the 8 norm weights of one 16-byte piece of x: requested here (fp32 weights: two 16-byte loads, bf16: one), turned into
// floats by normw_get — AFTER every other request of the prologue is out: a conversion next to the load is a wait for it
struct NormW { f32x4 lo, hi; };
__device__ __forceinline__ NormW normw_load(const void* w, int f32, int i0) {
  // no branch: a value that arrives through one of two paths is copied where they join, and a copy is a wait for the load
  const char* base = (const char*)w + (size_t)i0 * (f32 ? 4 : 2);
  NormW r;
  r.lo = *(const f32x4*)base;
  r.hi = *(const f32x4*)(base + (f32 ? 16 : 0));       // (bf16 weights: the same 16 bytes again)
  return r;
}
__device__ __forceinline__ void normw_get(const NormW& r, int f32, float (&out)[8]) {
  if (f32) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { out[j] = r.lo[j]; out[4 + j] = r.hi[j]; }
  } else {
    const bf16x8 v = __builtin_bit_cast(bf16x8, r.lo);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = (float)v[j];
  }
}

// sum over the 64 lanes, as a wave-uniform value (DPP row shifts / broadcasts + one readlane: no LDS crossbar)
__device__ __forceinline__ float wave_total(float v) {
  v = ssdk::wave_incl_scan_dpp(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float dot8(bf16x8 a, bf16x8 b, float acc) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    acc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{a[2 * j], a[2 * j + 1]}, bf16x2{b[2 * j], b[2 * j + 1]}, acc, false);
  return acc;
}

// f(x) -> LDS (bf16, [M][K]); ends with a barrier
template <int PRO, typename F>
__device__ __forceinline__ void gemv_prologue(const GemvArgs& a, bf16_t* fx, float* red, F&& after_loads) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, nv = K / 8;
  for (int m = 0; m < a.M; ++m) {
    const bf16_t* xr = a.x + (int64_t)m * a.xs;
    bf16_t* fr = fx + (int64_t)m * K;
    if (PRO == PRO_NONE || PRO == PRO_RELU2) {
      if (m == 0) after_loads();
      for (int iv = tid; iv < nv; iv += GV_THREADS) {
        bf16x8 v = *(const bf16x8*)(xr + 8 * iv);
        if (PRO == PRO_RELU2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float r = fmaxf((float)v[j], 0.f);
            v[j] = (bf16_t)(r * r);
          }
        }
        *(bf16x8*)(fr + 8 * iv) = v;
      }
    } else if (PRO == PRO_RMSNORM) {
      constexpr int MAXV = 4;                        // K <= 8 * 256 * 4 (launcher)
      float vals[MAXV][8];
      NormW wq[MAXV];                                // (the weights are requested with x, not behind the reduction)
      float ssq = 0.f;
      bf16x8 xq[MAXV], dq[MAXV];
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {               // (lanes past the end request the last piece again: no branch around a load)
        const int iv = min(tid + k * GV_THREADS, nv - 1);
        wq[k] = normw_load(a.norm_w, a.norm_w_f32, 8 * iv);
        xq[k] = *(const bf16x8*)(xr + 8 * iv);
        dq[k] = *(const bf16x8*)((a.delta ? a.delta + (int64_t)m * a.ds : xr) + 8 * iv);      // (no delta: x again, unused)
      }
      __builtin_amdgcn_sched_barrier(0);
      if (m == 0) after_loads();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        const int iv = tid + k * GV_THREADS;
        if (iv < nv) {
          bf16x8 v = xq[k];
          if (a.delta) {
            const bf16x8 d = dq[k];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16_t)((float)v[j] + (float)d[j]);     // the add rounds to bf16 (:966)
          }
          if (a.sum_out && blockIdx.x == 0) *(bf16x8*)(a.sum_out + (int64_t)m * a.ss + 8 * iv) = v;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            vals[k][j] = (float)v[j];
            ssq = fmaf(vals[k][j], vals[k][j], ssq);
          }
        }
      }
      ssq = wave_sum(ssq);
      __syncthreads();                               // (red is reused by the next row)
      if (lane == 0) red[wave] = ssq;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int i = 0; i < GV_WAVES; ++i) tot += red[i];
      const float rstd = rsqrtf(tot / (float)K + a.eps);
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        const int iv = tid + k * GV_THREADS;
        if (iv < nv) {
          bf16x8 o;
          float wv[8];
          normw_get(wq[k], a.norm_w_f32, wv);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(wv[j] * (vals[k][j] * rstd));
          *(bf16x8*)(fr + 8 * iv) = o;
        }
      }
    } else {                                         // PRO_GATED: a wave per group of `group` channels
      constexpr int MAXV = 4, GPW = 1;               // group <= 8 * 64 * 4 (launcher)
      const int ngroups = K / a.group, gv = a.group / 8;
      if (m == 0 && wave >= ngroups) after_loads();
      for (int g0 = wave; g0 < ngroups; g0 += GPW * GV_WAVES) {
        bf16x8 xv[GPW][MAXV], zv[GPW][MAXV];
        NormW wq[GPW][MAXV];                         // (x, gate and weights of a group are requested together, the first group's in front of W)
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
          const int g = min(g0 + i * GV_WAVES, ngroups - 1);
#pragma unroll
          for (int k = 0; k < MAXV; ++k) {
            const int iv = min(lane + k * 64, gv - 1);
            xv[i][k] = *(const bf16x8*)(xr + (int64_t)g * a.group + 8 * iv);
            zv[i][k] = *(const bf16x8*)((a.gate ? a.gate + (int64_t)m * a.gs : xr) + (int64_t)g * a.group + 8 * iv);
            wq[i][k] = normw_load(a.norm_w, a.norm_w_f32, g * a.group + 8 * iv);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (m == 0 && g0 == wave) after_loads();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
          const int g = g0 + i * GV_WAVES;
          if (g < ngroups) {
            float vals[MAXV][8];
            float ssq = 0.f;
#pragma unroll
            for (int k = 0; k < MAXV; ++k) {
              const int iv = lane + k * 64;
              if (iv < gv) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  float f = (float)xv[i][k][j];
                  if (a.gate) {
                    const float gt = (float)zv[i][k][j];
                    f *= gt * __builtin_amdgcn_rcpf(1.f + __expf(-gt));
                  }
                  vals[k][j] = f;
                  ssq = fmaf(f, f, ssq);
                }
              }
            }
            ssq = wave_sum(ssq);
            const float rstd = rsqrtf(ssq / (float)a.group + a.eps);
#pragma unroll
            for (int k = 0; k < MAXV; ++k) {
              const int iv = lane + k * 64;
              if (iv < gv) {
                bf16x8 o;
                float wv[8];
                normw_get(wq[i][k], a.norm_w_f32, wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(wv[j] * (vals[k][j] * rstd));
                *(bf16x8*)(fr + (int64_t)g * a.group + 8 * iv) = o;
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();

}

// 16 bytes of W; NT: non-temporal (a weight byte is used once a token: it should not displace x / the norm weights in L2)
template <bool NT>
__device__ __forceinline__ bf16x8 ldw8(const bf16_t* p) {
  return NT ? __builtin_nontemporal_load((const bf16x8*)p) : *(const bf16x8*)p;
}

// MT: 1 = one row of x (the batch-1 decode step), GV_MAXM = up to four (a.M);  R rows of W at a time, NB load
// instructions per row and batch (R * NB loads in flight per wave).  Used for K >= 8 192.
template <int PRO, int MT, int R, int NB, bool NT, bool PF>
__global__ __launch_bounds__(GV_THREADS) void gemv_bf16_kernel(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gv_smem[];
  bf16_t* fx = (bf16_t*)gv_smem;                     // [M][K]
  __shared__ float red[GV_WAVES];
  constexpr int GB = 4;                              // row groups between two meetings of the waves
  __shared__ float part[2][GB][GV_WAVES][R][MT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, nv = K / 8;

  // This wave's share of K: 16-byte pieces [p0, p0 + np) of every row, 64 per load instruction.  The first batch of a
  // row group is issued one group AHEAD — for the first group before the prologue (the loads do not depend on f(x)), for
  // the others before the math of the group in front of them.
  const int G = gridDim.x;
  const int pw = (nv + GV_WAVES - 1) / GV_WAVES, p0 = wave * pw;
  const int np = min(nv - p0, pw);                   // (<= 0: nothing for this wave)
  const int nins = np > 0 ? (np + 63) / 64 : 0;
  auto load_batch = [&](bf16x8 (&dst)[NB][R], int n0, int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (n0 + r * G < a.N) {                        // (work-group uniform)
        const bf16_t* wp = a.W + (int64_t)(n0 + r * G) * a.ldw + 8 * (p0 + lane);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const bf16x8 z = {};
          dst[u][r] = 64 * (i0 + u) + lane < np ? ldw8<NT>(wp + 512 * (i0 + u)) : z;
        }
      }
    }
  };
  bf16x8 cur[NB][R];
  // the requests for the first rows of W go out BEHIND those for the prologue's operands: a wave's loads return in order, and
  // the prologue must not wait for 16 KB of W to learn x
  gemv_prologue<PRO>(a, fx, red, [&]() __attribute__((always_inline)) { if ((int)blockIdx.x < a.N) load_batch(cur, blockIdx.x, 0); });

  // ---------------------------------------------------------------- the products
  const bf16_t* fl = fx + 8 * (p0 + lane);
  int par = 0, gi = 0, nflush = blockIdx.x;          // gi: groups since the last meeting; nflush: first row of the first of them
  for (int n0 = blockIdx.x; n0 < a.N; n0 += R * G) {
    float acc[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[r][m] = 0.f;
    auto fma_batch = [&](const bf16x8 (&w)[NB][R], int i0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < NB; ++u)
        if (64 * (i0 + u) + lane < np) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
            if (MT == 1 || m < a.M) {
              const bf16x8 xv = *(const bf16x8*)(fl + (int64_t)m * K + 512 * (i0 + u));
#pragma unroll
              for (int r = 0; r < R; ++r)
                if (n0 + r * G < a.N) acc[r][m] = dot8(w[u][r], xv, acc[r][m]);
            }
        }
    };
    const bool more = n0 + R * G < a.N;
    bf16x8 nxt[PF ? NB : 1][PF ? R : 1];
    if constexpr (PF) {
      if (more) load_batch(nxt, n0 + R * G, 0);
    } else {
      if (n0 != (int)blockIdx.x) load_batch(cur, n0, 0);
    }
    fma_batch(cur, 0);
    for (int i0 = NB; i0 < nins; i0 += NB) {
      bf16x8 wv[NB][R];
      load_batch(wv, n0, i0);
      fma_batch(wv, i0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float sm = wave_total(acc[r][m]);
        if (lane == 0) part[par][gi][wave][r][m] = sm;
      }
    // the four partial sums of a row meet in LDS — every GB groups, not every group: between two meetings the waves run
    // free (a barrier per 2 rows of 20 KB paced the K = 10 240 products); the other parity takes the next GB groups
    if (++gi == GB || !more) {
      __syncthreads();
      for (int i = tid; i < gi * R * MT; i += GV_THREADS) {
        const int j = i / (R * MT), r = (i / MT) % R, m = i % MT, n = nflush + (j * R + r) * G;
        if (n < a.N && (MT == 1 || m < a.M)) {
          float sm = 0.f;
#pragma unroll
          for (int w = 0; w < GV_WAVES; ++w) sm += part[par][j][w][r][m];
          a.y[(int64_t)m * a.ys + n] = (bf16_t)(sm + (a.bias ? (float)a.bias[n] : 0.f));
        }
      }
      par ^= 1; gi = 0; nflush = n0 + R * G;
    }
    if constexpr (PF) {
      if (more) {
#pragma unroll
        for (int u = 0; u < NB; ++u)
#pragma unroll
          for (int r = 0; r < R; ++r) cur[u][r] = nxt[u][r];
      }
    }
  }
}

// The same product with a ROW of W per wave (two at a time, 16 loads in flight), for K < 8 192: a row is 3 - 16 load
// instructions, too few to split four ways (the waves of a work-group would meet at a barrier every 12 loads), and N is
// large where K is small in this model (in_proj 22 656 x 4 480, up_proj 15 680 x 4 480: 8 - 11 rows a wave).
template <int PRO, int MT, bool NT, int R, int NB>
__global__ __launch_bounds__(GV_THREADS) void gemv_rows_kernel(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gv_smem[];
  bf16_t* fx = (bf16_t*)gv_smem;                     // [M][K]
  __shared__ float red[GV_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, nv = K / 8;
  const int TW = gridDim.x * GV_WAVES, gw = blockIdx.x * GV_WAVES + wave;
  const int nins = (nv + 63) / 64;
  auto load_batch = [&](bf16x8 (&dst)[NB][R], int n0, int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (n0 + r * TW < a.N) {                       // (wave uniform)
        const bf16_t* wp = a.W + (int64_t)(n0 + r * TW) * a.ldw + 8 * lane;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const bf16x8 z = {};
          dst[u][r] = 64 * (i0 + u) + lane < nv ? ldw8<NT>(wp + 512 * (i0 + u)) : z;
        }
      }
    }
  };
  bf16x8 cur[NB][R];
  // in flight while the prologue runs, requested behind the prologue's own operands (a wave's loads return in order)
  gemv_prologue<PRO>(a, fx, red, [&]() __attribute__((always_inline)) { if (gw < a.N) load_batch(cur, gw, 0); });

  const bf16_t* fl = fx + 8 * lane;
  for (int n0 = gw; n0 < a.N; n0 += R * TW) {
    float acc[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[r][m] = 0.f;
    auto fma_batch = [&](const bf16x8 (&w)[NB][R], int i0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < NB; ++u)
        if (64 * (i0 + u) + lane < nv) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
            if (MT == 1 || m < a.M) {
              const bf16x8 xv = *(const bf16x8*)(fl + (int64_t)m * K + 512 * (i0 + u));
#pragma unroll
              for (int r = 0; r < R; ++r)
                if (n0 + r * TW < a.N) acc[r][m] = dot8(w[u][r], xv, acc[r][m]);
            }
        }
    };
    if (n0 != gw) load_batch(cur, n0, 0);
    // lane r finishes row r of the pair; its conv-update operands (epilogue) are requested now, behind the W loads
    const int nmine = n0 + min(lane, R - 1) * TW;
    const bool conv_mine = MT == 1 && a.conv_state && lane < R && nmine < a.N && nmine >= a.conv_lo && nmine < a.conv_hi;
    bf16x4 c_old = {}, c_w = {};
    float c_b = 0.f;
    if (conv_mine) {
      const int c = nmine - a.conv_lo;
      c_old = *(const bf16x4*)(a.conv_state + (int64_t)c * 4);
      c_w = *(const bf16x4*)(a.conv_w + (int64_t)c * 4);
      if (a.conv_b) c_b = (float)a.conv_b[c];
    }
    fma_batch(cur, 0);
    for (int i0 = NB; i0 < nins; i0 += NB) {
      load_batch(cur, n0, i0);
      fma_batch(cur, i0);
    }
    float tot[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) tot[r][m] = (MT == 1 || m < a.M) ? wave_total(acc[r][m]) : 0.f;
    if (lane < R && nmine < a.N) {
      const float bs = a.bias ? (float)a.bias[nmine] : 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
        if (MT == 1 || m < a.M) {
          float v = tot[0][m];
#pragma unroll
          for (int r = 1; r < R; ++r) v = lane == r ? tot[r][m] : v;
          bf16_t out = (bf16_t)(v + bs);
          if (a.conv_state && nmine >= a.conv_lo && nmine < a.conv_hi) {
            // tv_causal_conv1d_update on this channel, with its arithmetic (conv1d.hip): window = state[1..3], out
            const int c = nmine - a.conv_lo;
            bf16_t* st = a.conv_state + ((int64_t)m * (a.conv_hi - a.conv_lo) + c) * 4;
            bf16x4 old = c_old, wv = c_w;
            float acc_c = c_b;
            if (MT != 1) {                           // (several rows of x: loaded here)
              old = *(const bf16x4*)st;
              wv = *(const bf16x4*)(a.conv_w + (int64_t)c * 4);
              acc_c = a.conv_b ? (float)a.conv_b[c] : 0.f;
            }
            const bf16x4 win = {old[1], old[2], old[3], out};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc_c = fmaf((float)wv[j], (float)win[j], acc_c);
            *(bf16x4*)st = win;
            out = (bf16_t)silu_f(acc_c);
          }
          a.y[(int64_t)m * a.ys + nmine] = out;
        }
    }
  }
}

// ---------------------------------------------------------------- round 6: one row of x, split-K, f(x) in registers
// The split-K kernel above stages f(x) in LDS behind a barrier: 1.5 - 3 us in front of the first product of a work-group
// that streams 160 - 280 KB in 16 - 24 us.  With one row of x a wave needs only ITS quarter of f(x) (5 - 8 pieces of 16
// bytes a lane: 20 - 32 registers), so it can hold it itself and start its products as soon as its own operands are
// there: no LDS, no barrier in front of the stream (out_proj 20.7 -> 18.0 us, down_proj 26.0 -> 23.9 us).
// Same rounding points and the same accumulation order of the products as the LDS kernel.  (The row-per-wave kernel
// stays on LDS: a wave would have to hold — and every one of the 2 048 waves to fetch — the whole of x, delta and the
// norm weights: measured slower inside the decode step, 3.71 against 3.54 ms a token.)
// split-K: wave w of a work-group holds pieces [w pw, w pw + np) of f(x) (np <= 64 NB), R rows of W at a time
// FULL: every wave's slice is exactly NB load instructions (no lane is ever past its end: no predicate anywhere)
template <int PRO, int R, int NB, bool NT, bool FULL>
__global__ __launch_bounds__(GV_THREADS) void gemv_splitk_reg_kernel(GemvArgs a) {
  static_assert(PRO == PRO_NONE || PRO == PRO_RELU2 || PRO == PRO_GATED, "prologues of the split-K register kernel");
  constexpr int GB = 4, MAXG = 4;                    // row groups between two meetings of the waves; norm groups a slice
  __shared__ float part[2][GB][GV_WAVES][R];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, nv = K / 8;
  const int G = gridDim.x;
  const int pw = (nv + GV_WAVES - 1) / GV_WAVES, p0 = wave * pw;
  const int np = min(nv - p0, pw);                   // (= pw > 0: launcher)
  bf16x8 fx[NB];
  bf16x8 zv[PRO == PRO_GATED ? NB : 1];
  NormW wq[PRO == PRO_GATED ? NB : 1];
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int q = 64 * u + lane;
    const bf16x8 z = {};
    const int qc = p0 + (FULL ? q : min(q, np - 1));   // (np = pw > 0, launcher; lanes past the end: the last piece again, unused)
    fx[u] = *(const bf16x8*)(a.x + 8 * qc);
    if constexpr (PRO == PRO_GATED) {
      zv[u] = *(const bf16x8*)((a.gate ? a.gate : a.x) + 8 * qc);
      wq[u] = normw_load(a.norm_w, a.norm_w_f32, 8 * qc);
    }
    (void)z;
  }
  auto load_rows = [&](bf16x8 (&dst)[NB][R], int n0) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (n0 + r * G < a.N) {                        // (work-group uniform)
        const bf16_t* wp = a.W + (int64_t)(n0 + r * G) * a.ldw + 8 * (p0 + lane);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const bf16x8 z = {};
          dst[u][r] = (FULL || 64 * u + lane < np) ? ldw8<NT>(wp + 512 * u) : z;
        }
      }
    }
  };
  bf16x8 cur[NB][R];
  load_rows(cur, blockIdx.x);
  __builtin_amdgcn_sched_barrier(0);                 // (the prologue's arithmetic stays behind the requests for W)
  if constexpr (PRO == PRO_RELU2) {
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float r = fmaxf((float)fx[u][j], 0.f);
        fx[u][j] = (bf16_t)(r * r);
      }
  } else if constexpr (PRO == PRO_GATED) {
    // the slice holds whole norm groups (launcher): piece q belongs to group q / gv of the slice
    const int gv = a.group / 8;
    float vals[NB][8], psq[NB];
    int gid[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int q = 64 * u + lane;
      gid[u] = (FULL || q < np) ? (int)((unsigned)q / (unsigned)gv) : -1;
      psq[u] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = (float)fx[u][j];
        if (a.gate) {
          const float gt = (float)zv[u][j];
          f *= gt * __builtin_amdgcn_rcpf(1.f + __expf(-gt));
        }
        vals[u][j] = f;
        psq[u] = fmaf(f, f, psq[u]);
      }
    }
    float rstd[MAXG];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
      float sg = 0.f;
#pragma unroll
      for (int u = 0; u < NB; ++u) sg += gid[u] == g ? psq[u] : 0.f;
      rstd[g] = rsqrtf(wave_sum_dpp(sg) / (float)a.group + a.eps);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      float rs = rstd[0];
#pragma unroll
      for (int g = 1; g < MAXG; ++g) rs = gid[u] == g ? rstd[g] : rs;
      float wv[8];
      normw_get(wq[u], a.norm_w_f32, wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) fx[u][j] = (bf16_t)(wv[j] * (vals[u][j] * rs));
    }
  }

  int par = 0, gi = 0, nflush = blockIdx.x;          // gi: groups since the last meeting; nflush: first row of the first of them
  for (int n0 = blockIdx.x; n0 < a.N; n0 += R * G) {
    if (n0 != (int)blockIdx.x) load_rows(cur, n0);
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < NB; ++u)
      if (FULL || 64 * u + lane < np) {
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (n0 + r * G < a.N) acc[r] = dot8(cur[u][r], fx[u], acc[r]);
      }
    const bool more = n0 + R * G < a.N;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float sm = wave_total(acc[r]);
      if (lane == 0) part[par][gi][wave][r] = sm;
    }
    if (++gi == GB || !more) {
      __syncthreads();
      for (int i = tid; i < gi * R; i += GV_THREADS) {
        const int j = i / R, r = i % R, n = nflush + (j * R + r) * G;
        if (n < a.N) {
          float sm = 0.f;
#pragma unroll
          for (int w = 0; w < GV_WAVES; ++w) sm += part[par][j][w][r];
          a.y[n] = (bf16_t)(sm + (a.bias ? (float)a.bias[n] : 0.f));
        }
      }
      par ^= 1; gi = 0; nflush = n0 + R * G;
    }
  }
}

int cu_count() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
      cus = 256;                               // MI355X
    return cus;
  }();
  return n;
}

// dev switches (A/B runs): TV_GEMV_NT = 0 plain loads of W; TV_GEMV_WGS = work-groups per CU (default 2)
int gemv_nt() { static const int v = [] { const char* e = getenv("TV_GEMV_NT"); return e ? atoi(e) : 1; }(); return v; }
int gemv_wgs() { static const int v = [] { const char* e = getenv("TV_GEMV_WGS"); return e ? atoi(e) : 2; }(); return v < 1 ? 1 : v; }

template <int PRO, int MT, int R, int NB, bool NT, bool PF>
int launch_gemv_r(const GemvArgs& a, hipStream_t st) {
  const size_t lds = (size_t)a.M * a.K * sizeof(bf16_t);
  static const hipError_t attr = hipFuncSetAttribute((const void*)gemv_bf16_kernel<PRO, MT, R, NB, NT, PF>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  if (attr != hipSuccess) {
    tv_set_error("gemv: cannot reserve 128 KiB of LDS: %s", hipGetErrorString(attr));
    return TV_ERR_LAUNCH;
  }
  // two work-groups per CU; fewer when there are not R rows for each
  int wgs = (a.N + R - 1) / R;
  const int most = gemv_wgs() * cu_count();
  if (wgs > most) wgs = most;
  gemv_bf16_kernel<PRO, MT, R, NB, NT, PF><<<dim3((unsigned)wgs), GV_THREADS, lds, st>>>(a);
  TV_LAUNCH_CHECK();
}
template <int PRO, int MT, bool NT, int R, int NB>
int launch_gemv_rows(const GemvArgs& a, hipStream_t st) {
  const size_t lds = (size_t)a.M * a.K * sizeof(bf16_t);
  static const hipError_t attr = hipFuncSetAttribute((const void*)gemv_rows_kernel<PRO, MT, NT, R, NB>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  if (attr != hipSuccess) {
    tv_set_error("gemv: cannot reserve 128 KiB of LDS: %s", hipGetErrorString(attr));
    return TV_ERR_LAUNCH;
  }
  int wgs = (a.N + R * GV_WAVES - 1) / (R * GV_WAVES);            // >= one row group per wave, two work-groups per CU
  const int most = gemv_wgs() * cu_count();
  if (wgs > most) wgs = most;
  gemv_rows_kernel<PRO, MT, NT, R, NB><<<dim3((unsigned)wgs), GV_THREADS, lds, st>>>(a);
  TV_LAUNCH_CHECK();
}
template <int PRO, bool NT>
int launch_splitk_reg(const GemvArgs& a, hipStream_t st) {
  int wgs = (a.N + 1) / 2;
  const int most = gemv_wgs() * cu_count();
  if (wgs > most) wgs = most;
  const int pw = a.K / 8 / GV_WAVES;                 // (reg_takes: K / 8 is a multiple of the four waves)
  if (pw == 64 * 5) gemv_splitk_reg_kernel<PRO, 2, 5, NT, true><<<dim3((unsigned)wgs), GV_THREADS, 0, st>>>(a);          // K = 10 240
  else if (pw == 64 * 8) gemv_splitk_reg_kernel<PRO, 2, 8, NT, true><<<dim3((unsigned)wgs), GV_THREADS, 0, st>>>(a);     // K = 16 384
  else gemv_splitk_reg_kernel<PRO, 2, 8, NT, false><<<dim3((unsigned)wgs), GV_THREADS, 0, st>>>(a);
  TV_LAUNCH_CHECK();
}
// One row of x (the batch-1 decode step), split-K, with f(x) in registers where the shape allows it; TV_GEMV_REG=0 (dev) keeps the LDS kernel
template <int PRO>
bool reg_takes(const GemvArgs& a) {
  static const int on = [] { const char* e = getenv("TV_GEMV_REG"); return e ? atoi(e) : 1; }();
  if (!on || a.M != 1 || a.conv_state) return false;
  const int nv = a.K / 8, pw = nv / GV_WAVES;
  if (nv % GV_WAVES || pw > 64 * 8) return false;
  if (PRO == PRO_GATED) return (pw * 8) % a.group == 0 && pw * 8 / a.group <= 4;       // whole norm groups a wave
  return PRO == PRO_NONE || PRO == PRO_RELU2;
}
template <int PRO, int MT>
int launch_gemv_m(const GemvArgs& a, hipStream_t st) {
  static const int force = [] { const char* e = getenv("TV_GEMV_SPLITK"); return e ? atoi(e) : -1; }();    // (dev: 0 / 1)
  const bool splitk = a.conv_state ? false : (force >= 0 ? force != 0 : a.K >= 8192);
  if constexpr (MT == 1 && PRO != PRO_RMSNORM) {
    if (splitk && reg_takes<PRO>(a)) return gemv_nt() ? launch_splitk_reg<PRO, true>(a, st) : launch_splitk_reg<PRO, false>(a, st);
  }
  // (the gated prologue holds two groups' operands in registers: no second set of W rows beside them)
  constexpr bool PF = PRO != PRO_GATED;
  if (!gemv_nt()) return splitk ? launch_gemv_r<PRO, MT, 2, 8, false, PF>(a, st) : launch_gemv_rows<PRO, MT, false, 2, 10>(a, st);
  return splitk ? launch_gemv_r<PRO, MT, 2, 8, true, PF>(a, st) : launch_gemv_rows<PRO, MT, true, 2, 10>(a, st);
}
template <int PRO>
int launch_gemv(const GemvArgs& a, hipStream_t st) {
  return a.M == 1 ? launch_gemv_m<PRO, 1>(a, st) : launch_gemv_m<PRO, GV_MAXM>(a, st);
}

}  // namespace

extern "C" int tv_gemv_bf16_fwd(const void* x, const void* W, const void* bias, void* y, int M, int N, int K,
                                int64_t x_stride, int64_t ldw, int64_t y_stride, int prologue, const void* delta,
                                int64_t delta_stride, void* sum_out, int64_t sum_stride, const void* norm_weight,
                                int norm_weight_dtype, float eps, const void* gate, int64_t gate_stride,
                                int group_size, void* conv_state, const void* conv_weight, const void* conv_bias,
                                int conv_row0, int conv_channels, void* stream) {
  TV_CHECK_ARG(x && W && y, "gemv: null pointer");
  TV_CHECK_ARG(M >= 1 && M <= GV_MAXM && N >= 1 && K >= 8, "gemv: M %d must be 1..%d, N %d >= 1, K %d >= 8", M, GV_MAXM, N, K);
  if (K % 8 || ldw % 8 || x_stride % 8 || ((uintptr_t)x & 15) || ((uintptr_t)W & 15))
    TV_UNSUPPORTED("gemv: K, the row strides of x and W must be multiples of 8 elements, x and W 16-byte aligned");
  if ((size_t)M * K * 2 > 128 * 1024) TV_UNSUPPORTED("gemv: M x K = %d x %d does not fit the 128 KiB of LDS f(x) is held in", M, K);
  TV_CHECK_ARG(prologue >= PRO_NONE && prologue <= PRO_GATED, "gemv: prologue %d", prologue);
  GemvArgs a;
  a.x = (const bf16_t*)x; a.W = (const bf16_t*)W; a.bias = (const bf16_t*)bias; a.y = (bf16_t*)y;
  a.delta = nullptr; a.gate = nullptr; a.norm_w = nullptr; a.sum_out = nullptr;
  a.M = M; a.N = N; a.K = K; a.group = K; a.norm_w_f32 = 0;
  a.xs = x_stride; a.ldw = ldw; a.ys = y_stride; a.ds = a.ss = a.gs = 0; a.eps = eps;
  if (prologue == PRO_RMSNORM || prologue == PRO_GATED) {
    TV_CHECK_ARG(norm_weight && (norm_weight_dtype == TV_F32 || norm_weight_dtype == TV_BF16), "gemv: norm weight (fp32 or bf16) needed");
    a.norm_w = norm_weight; a.norm_w_f32 = norm_weight_dtype == TV_F32;
  }
  if (prologue == PRO_RMSNORM) {
    if (K > 8 * GV_THREADS * 4) TV_UNSUPPORTED("gemv: rmsnorm prologue holds a row of <= %d channels", 8 * GV_THREADS * 4);
    if (delta && (delta_stride % 8 || ((uintptr_t)delta & 15))) TV_UNSUPPORTED("gemv: delta must be 16-byte aligned rows");
    if (sum_out && (sum_stride % 8 || ((uintptr_t)sum_out & 15))) TV_UNSUPPORTED("gemv: sum_out must be 16-byte aligned rows");
    a.delta = (const bf16_t*)delta; a.ds = delta_stride; a.sum_out = (bf16_t*)sum_out; a.ss = sum_stride;
  }
  if (prologue == PRO_GATED) {
    TV_CHECK_ARG(group_size > 0 && K % group_size == 0 && group_size % 8 == 0, "gemv: group size %d must divide K %d (multiple of 8)", group_size, K);
    if (group_size > 8 * 64 * 4) TV_UNSUPPORTED("gemv: gated-norm prologue holds groups of <= %d channels", 8 * 64 * 4);
    if (gate && (gate_stride % 8 || ((uintptr_t)gate & 15))) TV_UNSUPPORTED("gemv: gate must be 16-byte aligned rows");
    a.gate = (const bf16_t*)gate; a.gs = gate_stride; a.group = group_size;
  }
  a.conv_state = nullptr; a.conv_w = a.conv_b = nullptr; a.conv_lo = a.conv_hi = 0;
  if (conv_state) {
    TV_CHECK_ARG(conv_weight && conv_row0 >= 0 && conv_channels > 0 && conv_row0 + conv_channels <= N,
                 "gemv: conv epilogue rows [%d, %d) outside the %d outputs", conv_row0, conv_row0 + conv_channels, N);
    if (K >= 8192) TV_UNSUPPORTED("gemv: the conv epilogue is built into the row-per-wave kernel (K < 8192)");
    if (((uintptr_t)conv_state & 7) || ((uintptr_t)conv_weight & 7)) TV_UNSUPPORTED("gemv: conv state / weight must be 8-byte aligned");
    a.conv_state = (bf16_t*)conv_state; a.conv_w = (const bf16_t*)conv_weight; a.conv_b = (const bf16_t*)conv_bias;
    a.conv_lo = conv_row0; a.conv_hi = conv_row0 + conv_channels;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (prologue) {
    case PRO_NONE: return launch_gemv<PRO_NONE>(a, st);
    case PRO_RMSNORM: return launch_gemv<PRO_RMSNORM>(a, st);
    case PRO_RELU2: return launch_gemv<PRO_RELU2>(a, st);
    default: return launch_gemv<PRO_GATED>(a, st);
  }
}
