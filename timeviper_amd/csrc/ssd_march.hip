// S3 (fast path): Mamba-2 SSD selective scan as a single-pass "chunk march" on CDNA4.
//
// One 512-thread workgroup owns (batch, head, a <=48-column slice of head_dim) and walks
// the sequence in chunks of Q=64 tokens, carrying the running state X[n][p] (d_state x
// slice) in MFMA accumulators for the whole sequence: x, dt, B, C are read ONCE and y is
// written ONCE — no per-chunk states ever touch HBM (upstream's five-kernel pipeline moves
// ~2.5x the algorithmic bytes).  With H*slices >= 256 workgroups every CU streams; heads
// of one B/C group are mapped to one XCD (blockIdx % 8) so the group's B/C tiles are
// fetched from HBM once and served to its 32 workgroups by that XCD's L2.
//
// Per chunk (all products on v_mfma_f32_16x16x32_bf16, fp32 accumulate):
//   wave 0            dt = softplus(dt_raw + bias); cs = wave-prefix-sum(dt*A)   (64 lanes = Q)
//   CB^T[s][t]      = sum_n B[s][n] C[t][n]                    causal 16x16 tiles only
//   M[t][s]         = CB * exp(cs_t - cs_s) * dt_s * [s<=t]    -> LDS (bf16)
//   Yoff^T[p][t]    = sum_n S[p][n] C[t][n]                    S = bf16 copy of X in LDS
//   Ydiag^T[p][t]   = sum_s x[s][p] M[t][s]
//   y[t][p]         = Ydiag + exp(cs_t) * Yoff + D x[t][p]     -> LDS tile -> 16-byte row stores
//   X[n][p]         = exp(cs_Q) X[n][p] + sum_t B[t][n] (exp(cs_Q - cs_t) dt_t x[t][p])
// Waves 0-3 own one 16-token row block of y each; waves 4-7 own two 16-row blocks of the
// state each and the bulk of the CB tiles, so every SIMD carries one "y" and one "state"
// wave.  Next-chunk tiles are prefetched into registers at the top of a step and written to
// the other LDS buffer at its end (two barriers per step).  Operands that MFMA wants
// k-major come from the row-major tiles through ds_read_b64_tr_b16.
//
// Decay factors are only ever formed as exp(cs_i - cs_j) with i >= j inside one chunk
// (never a quotient of exponentials), like the reference's segment_sum
// (modeling_nano.py:159-186), so no input can overflow them.
// Reference semantics: mamba_chunk_scan_combined call modeling_nano.py:639-653;
// arithmetic :775-851.
#include "common.hpp"

namespace {

constexpr int MQ = 64;          // tokens per chunk (= wavefront width)
constexpr int MN = 128;         // d_state
constexpr int MTHREADS = 512;
constexpr int BSTR = MN + 8;    // B/C/S tile row stride (elements): +16 B against bank conflicts
constexpr int XSTR = 96;        // x tile row stride: 192 B == 48 dwords (mod 64) for tr-reads
constexpr int MSTR = MQ + 8;    // M tile row stride
constexpr int YSTR = 48;        // y tile row stride
constexpr int PMAX = 48;        // max head_dim columns per workgroup (3 MFMA tiles)

typedef __attribute__((address_space(3))) bf16x4 lds_v4;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 tr4(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)p);
}
// k-major 16x32 operand fragment from a row-major [k][col] LDS tile: element j of lane
// (c = lane&15, kq = lane>>4) is tile[k0 + 8kq + j][c0 + c].
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int stride, int k0, int c0, int lane) {
  const int kq = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const bf16_t* base = tile + (k0 + 8 * kq + q4) * stride + c0 + 4 * p4;
  const bf16x4 lo = tr4(base);
  const bf16x4 hi = tr4(base + 4 * stride);
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
  return r;
}
// row-major fragment: element j of lane (r = lane&15, kq) is tile[r0 + r][k0 + 8kq + j]
__device__ __forceinline__ bf16x8 row_frag(const bf16_t* tile, int stride, int r0, int k0, int lane) {
  return *(const bf16x8*)(tile + (r0 + (lane & 15)) * stride + k0 + 8 * (lane >> 4));
}

struct MarchArgs {
  const bf16_t *x, *dt, *Bm, *Cm;
  const float *A, *D, *dt_bias, *init;
  bf16_t* y;
  float *final_state, *total_decay;
  int L, H, P, G, nslices, pw;
  int64_t xsb, xsl, dsb, dsl, bsb, bsl, csb, csl, ysb, ysl;
  int softplus, group_map;
  float dt_min, dt_max;
};

struct __attribute__((aligned(16))) MarchSmem {
  bf16_t Bt[2][MQ * BSTR];
  bf16_t Ct[2][MQ * BSTR];
  bf16_t xt[2][MQ * XSTR];
  bf16_t S[2][PMAX * BSTR];
  bf16_t M[MQ * MSTR];
  bf16_t yt[MQ * YSTR];
  float cs[2][MQ];    // inclusive cumsum of dt*A inside the chunk
  float dtv[2][MQ];   // discretised dt
  float wts[2][MQ];   // exp(cs_last - cs_t) * dt_t
  float ecs[2][MQ];   // exp(cs_t)
  float dlast[2];     // exp(cs_last)
};

// causal CB tiles (t-tile, s-tile), s <= t
__constant__ int kCbT[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3};
__constant__ int kCbS[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};

template <int PT>
__global__ __launch_bounds__(MTHREADS) void ssd_march_kernel(MarchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  MarchSmem& sm = *reinterpret_cast<MarchSmem*>(smem_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lc = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y;
  // blockIdx.x -> (group, head in group, slice): blocks with equal (blockIdx.x % G) share an
  // XCD under round-robin dispatch when G == 8 (speed only, never correctness)
  const int hpg = a.H / a.G;
  int g, hig, slice;
  {
    const int bx = blockIdx.x;
    g = bx % a.G;
    const int rest = bx / a.G;
    hig = rest / a.nslices;
    slice = rest % a.nslices;
  }
  const int h = a.group_map ? (hig * a.G + g) : (g * hpg + hig);
  const int p_base = slice * a.pw;
  const int pw = a.pw;
  const int nchunks = (a.L + MQ - 1) / MQ;
  const float Ah = a.A[h];
  const float Dh = a.D ? a.D[h] : 0.f;
  const float bias = a.dt_bias ? a.dt_bias[h] : 0.f;

  const bf16_t* xg = a.x + (int64_t)b * a.xsb + (int64_t)h * a.P + p_base;
  const bf16_t* dtg = a.dt + (int64_t)b * a.dsb + h;
  const bf16_t* Bg = a.Bm + (int64_t)b * a.bsb + (int64_t)g * MN;
  const bf16_t* Cg = a.Cm + (int64_t)b * a.csb + (int64_t)g * MN;
  bf16_t* yg = a.y + (int64_t)b * a.ysb + (int64_t)h * a.P + p_base;

  // ---- zero the LDS regions that are read but never (fully) written ----
  {
    bf16x8 z = {};
    for (int i = tid; i < (int)(sizeof(MarchSmem) / 16); i += MTHREADS)
      reinterpret_cast<bf16x8*>(smem_raw)[i] = z;
  }
  __syncthreads();

  const bool ywave = wave < 4;
  // state accumulators X[n][p]: S-wave w owns n-tiles 2(w-4), 2(w-4)+1; col p = lane&15
  f32x4 xacc[2][PT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < PT; ++j) xacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!ywave && a.init) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        const int p = 16 * j + lc, n = 16 * (2 * (wave - 4) + i) + 4 * kq;
        if (p < pw) {
          const f32x4 v = *(const f32x4*)(a.init + (((int64_t)b * a.H + h) * a.P + p_base + p) * MN + n);
          xacc[i][j] = v;
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
          *(bf16x4*)(sm.S[0] + p * BSTR + n) = o;
        }
      }
  }

  // ---- staging registers for the next chunk ----
  bf16x8 rB[2], rC[2], rX;
  float rdt = 0.f;
  const int npc = pw >> 3;                   // 16-byte pieces per x / y row
  const int xrow = tid / npc, xch = tid % npc;
  auto issue_loads = [&](int c) {
    const int t0 = c * MQ;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * MTHREADS;      // 1024 16-byte pieces per tile
      const int row = i >> 4, ch = i & 15;
      const int t = t0 + row;
      bf16x8 z = {};
      rB[k] = t < a.L ? *(const bf16x8*)(Bg + (int64_t)t * a.bsl + ch * 8) : z;
      rC[k] = t < a.L ? *(const bf16x8*)(Cg + (int64_t)t * a.csl + ch * 8) : z;
    }
    {
      const int t = t0 + xrow;
      bf16x8 z = {};
      rX = (xrow < MQ && t < a.L) ? *(const bf16x8*)(xg + (int64_t)t * a.xsl + xch * 8) : z;
    }
    if (wave == 0) {
      const int t = t0 + lane;
      rdt = t < a.L ? (float)dtg[(int64_t)t * a.dsl] : -INFINITY;
    }
  };
  auto write_stage = [&](int buf) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * MTHREADS;
      const int row = i >> 4, ch = i & 15;
      *(bf16x8*)(sm.Bt[buf] + row * BSTR + ch * 8) = rB[k];
      *(bf16x8*)(sm.Ct[buf] + row * BSTR + ch * 8) = rC[k];
    }
    if (xrow < MQ) *(bf16x8*)(sm.xt[buf] + xrow * XSTR + xch * 8) = rX;
    if (wave == 0) {
      // discretise dt and prefix-sum dt*A across the 64 lanes of the wave
      float d = 0.f;
      if (rdt != -INFINITY) {
        d = rdt + bias;
        if (a.softplus) d = softplus_f(d);
        d = fminf(fmaxf(d, a.dt_min), a.dt_max);
      }
      const float cs = wave_incl_scan(d * Ah);
      const float cl = __shfl(cs, 63, 64);
      sm.cs[buf][lane] = cs;
      sm.dtv[buf][lane] = d;
      sm.wts[buf][lane] = __expf(cl - cs) * d;
      sm.ecs[buf][lane] = __expf(cs);
      if (lane == 0) sm.dlast[buf] = __expf(cl);
      return cl;
    }
    return 0.f;
  };

  float decay_total = 0.f;
  issue_loads(0);
  decay_total += write_stage(0);
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nchunks;
    if (more) issue_loads(c + 1);

    const bf16_t* Bt = sm.Bt[buf];
    const bf16_t* Ct = sm.Ct[buf];
    const bf16_t* xt = sm.xt[buf];
    const bf16_t* Sc = sm.S[buf];

    // ================= phase 1: Yoff (y-waves), CB^T -> M (mostly state-waves) ==========
    f32x4 yoff[PT];
    int ncb, cb0;
    if (ywave) {
#pragma unroll
      for (int j = 0; j < PT; ++j) yoff[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < MN / 32; ++ks) {
        const bf16x8 cf = row_frag(Ct, BSTR, 16 * wave, 32 * ks, lane);
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const bf16x8 sf = row_frag(Sc, BSTR, 16 * j, 32 * ks, lane);
          yoff[j] = mfma16(sf, cf, yoff[j]);
        }
      }
      ncb = wave < 2 ? 1 : 0;
      cb0 = 8 + wave;
    } else {
      ncb = 2;
      cb0 = 2 * (wave - 4);
    }
    for (int ic = 0; ic < ncb; ++ic) {
      const int ti = kCbT[cb0 + ic], si = kCbS[cb0 + ic];
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < MN / 32; ++ks) {
        const bf16x8 bf = row_frag(Bt, BSTR, 16 * si, 32 * ks, lane);   // rows s
        const bf16x8 cf = row_frag(Ct, BSTR, 16 * ti, 32 * ks, lane);   // cols t
        acc = mfma16(bf, cf, acc);
      }
      // acc[r] = CB^T[s = 16si + 4kq + r][t = 16ti + lc]
      const int t = 16 * ti + lc, s0 = 16 * si + 4 * kq;
      const float cst = sm.cs[buf][t];
      const f32x4 css = *(const f32x4*)(&sm.cs[buf][s0]);
      const f32x4 dts = *(const f32x4*)(&sm.dtv[buf][s0]);
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(fminf(cst - css[r], 0.f));
        o[r] = (s0 + r <= t) ? (bf16_t)(acc[r] * e * dts[r]) : (bf16_t)0.f;
      }
      *(bf16x4*)(sm.M + t * MSTR + s0) = o;
    }
    __syncthreads();

    // ================= phase 2: Ydiag + epilogue (y-waves), state update (state-waves) ===
    if (ywave) {
      f32x4 yd[PT];
#pragma unroll
      for (int j = 0; j < PT; ++j) yd[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int nks = (wave >> 1) + 1;   // s k-steps of 32 covering s <= 16*wave + 15
      for (int ks = 0; ks < nks; ++ks) {
        const bf16x8 mf = row_frag(sm.M, MSTR, 16 * wave, 32 * ks, lane);   // cols t, k = s
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const bf16x8 xf = tr_frag(xt, XSTR, 32 * ks, 16 * j, lane);       // rows p, k = s
          yd[j] = mfma16(xf, mf, yd[j]);
        }
      }
      // y^T[p = 16j + 4kq + r][t = 16wave + lc]
      const int t = 16 * wave + lc;
      const float e = sm.ecs[buf][t];
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        const int p0 = 16 * j + 4 * kq;
        const bf16x4 xv = *(const bf16x4*)(xt + t * XSTR + p0);
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          o[r] = (bf16_t)(yd[j][r] + e * yoff[j][r] + Dh * (float)xv[r]);
        *(bf16x4*)(sm.yt + t * YSTR + p0) = o;
      }
    } else {
      const float dl = sm.dlast[buf];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) xacc[i][j][r] *= dl;
#pragma unroll
      for (int ks = 0; ks < MQ / 32; ++ks) {
        // x~[t][p] = w_t x[t][p], k = t
        const f32x4 w0 = *(const f32x4*)(&sm.wts[buf][32 * ks + 8 * kq]);
        const f32x4 w1 = *(const f32x4*)(&sm.wts[buf][32 * ks + 8 * kq + 4]);
        bf16x8 xs[PT];
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const bf16x8 xf = tr_frag(xt, XSTR, 32 * ks, 16 * j, lane);       // k = t, cols p
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xs[j][e] = (bf16_t)((float)xf[e] * w0[e]);
            xs[j][4 + e] = (bf16_t)((float)xf[4 + e] * w1[e]);
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int n0 = 16 * (2 * (wave - 4) + i);
          const bf16x8 bf = tr_frag(Bt, BSTR, 32 * ks, n0, lane);           // rows n, k = t
#pragma unroll
          for (int j = 0; j < PT; ++j) xacc[i][j] = mfma16(bf, xs[j], xacc[i][j]);
        }
      }
      // bf16 copy of the new state for the next chunk's Yoff: S[p][n]
      bf16_t* Sn = sm.S[buf ^ 1];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const int p = 16 * j + lc, n = 16 * (2 * (wave - 4) + i) + 4 * kq;
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)xacc[i][j][r];
          *(bf16x4*)(Sn + p * BSTR + n) = o;
        }
    }
    if (more) decay_total += write_stage(buf ^ 1);
    __syncthreads();

    // ================= coalesced y store: 16 bytes per lane, whole rows =================
    {
      const int t = c * MQ + xrow;
      if (xrow < MQ && t < a.L)
        *(bf16x8*)(yg + (int64_t)t * a.ysl + xch * 8) = *(const bf16x8*)(sm.yt + xrow * YSTR + xch * 8);
    }
  }

  if (!ywave && a.final_state) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        const int p = 16 * j + lc, n = 16 * (2 * (wave - 4) + i) + 4 * kq;
        if (p < pw)
          *(f32x4*)(a.final_state + (((int64_t)b * a.H + h) * a.P + p_base + p) * MN + n) = xacc[i][j];
      }
  }
  if (a.total_decay && slice == 0 && tid == 0) a.total_decay[(int64_t)b * a.H + h] = decay_total;
}

bool pick_slices(int P, int* nslices, int* pw) {
  for (int ns = 1; ns <= 8; ++ns) {
    if (P % ns) continue;
    const int w = P / ns;
    if (w <= PMAX && w % 8 == 0) {
      // prefer >= 2 slices for P > 48 only; smaller heads take one slice
      *nslices = ns;
      *pw = w;
      return true;
    }
  }
  return false;
}

}  // namespace

bool tv_ssd_march_supported(int seqlen, int nheads, int headdim, int ngroups, int dstate,
                            int dtype, int64_t xsl, int64_t bsl, int64_t csl, int64_t ysl,
                            const void* x, const void* Bm, const void* Cm, const void* y) {
  int ns, pw;
  if (dtype != TV_BF16 || dstate != MN || seqlen < 1) return false;
  if (!pick_slices(headdim, &ns, &pw)) return false;
  if (xsl % 8 || bsl % 8 || csl % 8 || ysl % 8) return false;
  if (((uintptr_t)x & 15) || ((uintptr_t)Bm & 15) || ((uintptr_t)Cm & 15) || ((uintptr_t)y & 15))
    return false;
  if (headdim % 8) return false;
  (void)nheads; (void)ngroups;
  return true;
}

size_t tv_ssd_march_workspace_bytes(int, int, int, int, int, int) { return 0; }

int tv_ssd_march_launch(const void* x, const void* dt, const void* A, const void* Bm,
                        const void* Cm, const void* D, const void* dt_bias,
                        const void* init_state, void* y, void* final_state, void* total_decay,
                        int batch, int seqlen, int nheads, int headdim, int ngroups, int dstate,
                        int64_t xsb, int64_t xsl, int64_t dsb, int64_t dsl, int64_t bsb,
                        int64_t bsl, int64_t csb, int64_t csl, int64_t ysb, int64_t ysl,
                        int dtype, int dt_softplus, float dt_min, float dt_max, int group_map,
                        void* workspace, size_t workspace_bytes, hipStream_t st) {
  (void)workspace; (void)workspace_bytes; (void)dtype; (void)dstate;
  MarchArgs a;
  a.x = (const bf16_t*)x; a.dt = (const bf16_t*)dt; a.Bm = (const bf16_t*)Bm; a.Cm = (const bf16_t*)Cm;
  a.A = (const float*)A; a.D = (const float*)D; a.dt_bias = (const float*)dt_bias;
  a.init = (const float*)init_state; a.y = (bf16_t*)y; a.final_state = (float*)final_state;
  a.total_decay = (float*)total_decay;
  a.L = seqlen; a.H = nheads; a.P = headdim; a.G = ngroups;
  if (!pick_slices(headdim, &a.nslices, &a.pw)) TV_UNSUPPORTED("ssd_march: head_dim %d", headdim);
  a.xsb = xsb; a.xsl = xsl; a.dsb = dsb; a.dsl = dsl; a.bsb = bsb; a.bsl = bsl;
  a.csb = csb; a.csl = csl; a.ysb = ysb; a.ysl = ysl;
  a.softplus = dt_softplus; a.group_map = group_map; a.dt_min = dt_min; a.dt_max = dt_max;
  dim3 grid(nheads * a.nslices, batch);
  const size_t lds = sizeof(MarchSmem);
  const int pt = (a.pw + 15) / 16;
  hipError_t e = hipSuccess;
  switch (pt) {
    case 1:
      e = hipFuncSetAttribute((const void*)ssd_march_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess) ssd_march_kernel<1><<<grid, MTHREADS, lds, st>>>(a);
      break;
    case 2:
      e = hipFuncSetAttribute((const void*)ssd_march_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess) ssd_march_kernel<2><<<grid, MTHREADS, lds, st>>>(a);
      break;
    default:
      e = hipFuncSetAttribute((const void*)ssd_march_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess) ssd_march_kernel<3><<<grid, MTHREADS, lds, st>>>(a);
      break;
  }
  if (e != hipSuccess) {
    tv_set_error("ssd_march: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return TV_ERR_LAUNCH;
  }
  TV_LAUNCH_CHECK();
}
