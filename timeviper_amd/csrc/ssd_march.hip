// S3 (fast path): Mamba-2 SSD selective scan as a single-pass "chunk march" on CDNA4.
//
// One 512-thread workgroup owns (batch, head, a <=48-column slice of head_dim) and walks
// the sequence in chunks of Q=64 tokens, carrying the running state X[n][p] (d_state x
// slice) in MFMA accumulators for the whole sequence: x, dt, B, C are read ONCE and y is
// written ONCE — no per-chunk states, no C.B^T matrices and no dt/cumsum side arrays ever
// touch HBM (upstream's five-kernel pipeline moves ~2.5x the algorithmic bytes).  With
// H*slices >= 256 workgroups every CU streams; heads of one B/C group are mapped to one
// XCD (blockIdx % 8) so the group's B/C tiles are fetched from HBM once and served to its
// workgroups by that XCD's L2.
//
// Data movement: every input tile arrives by LDS-DMA (global_load_lds, 16 B per lane, no
// VGPR round trip) into a 3-slot ring, two chunks ahead of the math; waits are counted
// (s_waitcnt vmcnt(N), never 0 in the steady state) and the barriers are raw s_barrier, so
// the DMA stays in flight across them.  B/C tiles are stored unpadded and XOR-swizzled on
// the SOURCE address (chunk ^= row&15) so both the row-major (ds_read_b128) and the
// transposing (ds_read_b64_tr_b16) fragment reads are bank-conflict free.
//
// Per chunk (all products on v_mfma_f32_16x16x32_bf16, fp32 accumulate):
//   wave 0            dt = softplus(dt_raw + bias); cs = DPP wave-prefix-sum(dt*A) for the
//                     NEXT chunk (64 lanes = Q), off the critical path
//   y-waves 0..3      (16 tokens each)
//     Yoff^T[p][t]  = sum_n S[p][n] C[t][n]                    S = bf16 copy of X in LDS
//     CB^T[s][t]    = sum_n B[s][n] C[t][n]                    causal 16x16 tiles only
//     M^T[s][t]     = CB^T * exp(cs_t - cs_s) * dt_s * [s<=t]  in the accumulator registers,
//                     which ARE the B operand of the next product (k permuted, no LDS trip)
//     Ydiag^T[p][t] = sum_s x[s][p] M^T[s][t]
//     y[t][p]       = Ydiag + exp(cs_t) * Yoff + D x[t][p]     -> LDS tile -> 16-byte row stores
//   state-waves 4..7  (32 state rows each)
//     X[n][p]       = exp(cs_Q) X[n][p] + sum_t B[t][n] (exp(cs_Q - cs_t) dt_t x[t][p])
// Decay factors are only ever formed as exp(cs_i - cs_j) with i >= j inside one chunk
// (never a quotient of exponentials), like the reference's segment_sum
// (modeling_nano.py:159-186), so no input can overflow them.
// Reference semantics: mamba_chunk_scan_combined call modeling_nano.py:639-653;
// arithmetic :775-851.
#include <stdlib.h>
#include "common.hpp"

namespace {

constexpr int MQ = 64;          // tokens per chunk (= wavefront width)
constexpr int MN = 128;         // d_state
constexpr int MTHREADS = 512;
constexpr int SSTR = MN + 8;    // S tile row stride (elements): +16 B against bank conflicts
constexpr int YSTR = 48;        // y tile row stride
constexpr int MSTR = MQ + 8;    // M tile row stride (144 B: conflict-free b128 row reads)
constexpr int PMAX = 48;        // max head_dim columns per workgroup (3 MFMA tiles)
constexpr int NSLOT = 3;        // LDS ring depth (prefetch distance 2 chunks)

typedef __attribute__((address_space(3))) bf16x4 lds_v4;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 tr4(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)p);
}
__device__ __forceinline__ bf16x8 cat4(bf16x4 lo, bf16x4 hi) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
  return r;
}
// ---- swizzled [64][128] bf16 tile (B, C): element offset of 16-byte chunk cg of row r ----
__device__ __forceinline__ int swz(int r, int cg) { return r * MN + ((cg ^ (r & 15)) << 3); }
// row-major fragment: element j of lane (lc, kq) = tile[r0 + lc][k0 + 8kq + j]
__device__ __forceinline__ bf16x8 row_frag_swz(const bf16_t* tile, int r0, int k0, int lc, int kq) {
  return *(const bf16x8*)(tile + swz(r0 + lc, (k0 >> 3) + kq));
}
// k-major fragment: element j of lane (lc, kq) = tile[k0 + 8kq + j][c0 + lc]
__device__ __forceinline__ bf16x8 tr_frag_swz(const bf16_t* tile, int k0, int c0, int lane) {
  const int kq = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int col = c0 + 4 * p4;
  const int r_lo = k0 + 8 * kq + q4, r_hi = r_lo + 4;
  const bf16x4 lo = tr4(tile + swz(r_lo, col >> 3) + (col & 4));
  const bf16x4 hi = tr4(tile + swz(r_hi, col >> 3) + (col & 4));
  return cat4(lo, hi);
}
// k-major fragment from a plain row-major tile; first rows of the two 4-row blocks are given
__device__ __forceinline__ bf16x8 tr_frag_rows(const bf16_t* tile, int stride, int row_lo, int row_hi,
                                               int c0, int lane) {
  const int q4 = (lane & 15) >> 2, p4 = lane & 3;
  const bf16x4 lo = tr4(tile + (row_lo + q4) * stride + c0 + 4 * p4);
  const bf16x4 hi = tr4(tile + (row_hi + q4) * stride + c0 + 4 * p4);
  return cat4(lo, hi);
}

// LDS-DMA as inline asm: hipcc then does not know an LDS write is in flight, so it inserts
// no vmcnt(0) in front of the fragment reads (it does for the builtin + ds_read_tr); every
// wait on these copies is the hand-counted s_waitcnt vmcnt(N) + barrier below.  M0 (the LDS
// destination base) is saved and restored inside the statement.
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t*)p);
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// one dword per lane, written at lds_dst + 4*lane
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// inclusive prefix sum over the 64 lanes with DPP row shifts / broadcasts (no LDS)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_shift(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float wave_incl_scan_dpp(float v) {
  v += dpp_shift<0x111, 0xf>(v);   // row_shr:1
  v += dpp_shift<0x112, 0xf>(v);   // row_shr:2
  v += dpp_shift<0x114, 0xf>(v);   // row_shr:4
  v += dpp_shift<0x118, 0xf>(v);   // row_shr:8
  v += dpp_shift<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
  v += dpp_shift<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
  return v;
}

struct MarchArgs {
  const bf16_t *x, *dt, *Bm, *Cm;
  const float *A, *D, *dt_bias, *init;
  bf16_t* y;
  float *final_state, *total_decay;
  int L, H, P, G, nslices, pw;
  int64_t xsb, xsl, dsb, dsl, bsb, bsl, bsg, csb, csl, csg, ysb, ysl;
  int softplus, group_map;
  float dt_min, dt_max;
  int dbg;
};

struct __attribute__((aligned(16))) MarchSlot {
  bf16_t Bt[MQ * MN];            // swizzled
  bf16_t Ct[MQ * MN];            // swizzled
  bf16_t xt[MQ * PMAX + 64];     // [64][pw] linear (+ finite guard: tile PT-1 reads past pw)
};
struct __attribute__((aligned(16))) MarchSmem {
  MarchSlot slot[NSLOT];
  unsigned dtr[4][MQ];  // raw dt of heads (h&~1, h|1) per token, one chunk further ahead
                        // than the tiles (ring of 4)
  unsigned pad_[MQ];    // landing pad of the filler DMA
  bf16_t S[PMAX * SSTR];
  bf16_t yt[MQ * YSTR];
  bf16_t M[MQ * MSTR];          // decay-masked C.B^T of the chunk, [t][s]; s>t tiles stay 0
  float cs[2][MQ];    // inclusive cumsum of dt*A inside the chunk
  float dtv[2][MQ];   // discretised dt
  float wts[2][MQ];   // exp(cs_last - cs_t) * dt_t
  float dlast[2][4];  // exp(cs_last)
};

template <int PT>
__global__ __launch_bounds__(MTHREADS) void ssd_march_kernel(MarchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  MarchSmem& sm = *reinterpret_cast<MarchSmem*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y;
  // blockIdx.x -> (group, head in group, slice): blocks with equal (blockIdx.x % G) share an
  // XCD under round-robin dispatch when G == 8 (speed only, never correctness)
  const int hpg = a.H / a.G;
  const int g = blockIdx.x % a.G;
  const int rest = blockIdx.x / a.G;
  const int hig = rest / a.nslices, slice = rest % a.nslices;
  const int h = a.group_map ? (hig * a.G + g) : (g * hpg + hig);
  const int pw = a.pw, p_base = slice * pw;
  const int L = a.L;
  const int nchunks = (L + MQ - 1) / MQ;
  const float Ah = a.A[h];
  const float Dh = a.D ? a.D[h] : 0.f;
  const float bias = a.dt_bias ? a.dt_bias[h] : 0.f;

  const bf16_t* xg = a.x + (int64_t)b * a.xsb + (int64_t)h * a.P + p_base;
  const bf16_t* dtg = a.dt + (int64_t)b * a.dsb + (h & ~1);   // dword holding heads (h&~1, h|1)
  const bf16_t* Bg = a.Bm + (int64_t)b * a.bsb + (int64_t)g * a.bsg;
  const bf16_t* Cg = a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg;
  bf16_t* yg = a.y + (int64_t)b * a.ysb + (int64_t)h * a.P + p_base;

  // ---- zero LDS once: pad columns / guards must hold finite values ----
  {
    bf16x8 z = {};
    for (int i = tid; i < (int)(sizeof(MarchSmem) / 16); i += MTHREADS)
      reinterpret_cast<bf16x8*>(smem_raw)[i] = z;
  }
  __syncthreads();

  const bool ywave = wave < 4;
  const int sw = wave - 4;
  // state accumulators X[n][p]: state-wave sw owns n-tiles 2sw, 2sw+1; col p = lane&15
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
        const int p = 16 * j + lc, n = 16 * (2 * sw + i) + 4 * kq;
        if (p < pw) {
          const f32x4 v = *(const f32x4*)(a.init + (((int64_t)b * a.H + h) * a.P + p_base + p) * MN + n);
          xacc[i][j] = v;
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
          *(bf16x4*)(sm.S + p * SSTR + n) = o;
        }
      }
  }

  // ---- LDS-DMA issue: every wave moves 2 KB of B, 2 KB of C; waves 0..nxp-1 one x piece
  //      each, wave 5 the 64 raw dt values.  Rows past L are clamped to a valid row (their dt
  //      is forced to 0 in prep_chunk, so they contribute nothing).
  const int npc = pw >> 3;                    // 16-byte pieces per x / y row
  const int npx = MQ * npc;                   // 16-byte pieces per x / y tile
  const int nxp = (npx + 63) >> 6;            // x wave-instructions per chunk (<= 5 for pw <= 40)
  const int x_pc = wave < nxp ? wave : 0;
  const int x_i = min(x_pc * 64 + lane, npx - 1);
  const int x_row = x_i / npc, x_ch = x_i - x_row * npc;
  auto issue = [&](int c) {
    MarchSlot& s = sm.slot[c % NSLOT];
    const int t0 = c * MQ;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int piece = 2 * wave + k;         // 16 pieces of 4 rows
      const int row = 4 * piece + (lane >> 4);
      const int cg = (lane & 15) ^ (row & 15);
      const int64_t t = min(t0 + row, L - 1);
      glds16(Bg + t * a.bsl + cg * 8, lds_addr_of(s.Bt + piece * 512));
      glds16(Cg + t * a.csl + cg * 8, lds_addr_of(s.Ct + piece * 512));
    }
    // fifth DMA of this wave: an x piece (waves < nxp) or the raw dt (wave 6) — waves
    // that own neither re-fetch x piece 0 into the scratch pad so that EVERY wave issues
    // exactly 5 vm ops per step (uniform vmcnt bookkeeping)
    if (wave == 6) {   // raw dt of the chunk AFTER this one (prep runs one step early)
      const int64_t t = min(t0 + MQ + lane, L - 1);
      glds4(dtg + t * a.dsl, lds_addr_of(sm.dtr[(c + 1) & 3]));
    } else {
      const int64_t t = min(t0 + x_row, L - 1);
      if (wave < nxp) glds16(xg + t * a.xsl + x_ch * 8, lds_addr_of(s.xt + x_pc * 512));
      else glds4(dtg + min((int64_t)t0, (int64_t)L - 1) * a.dsl, lds_addr_of(sm.pad_));
    }
  };
  // discretise dt and prefix-sum dt*A for chunk c (one wave, 64 lanes = 64 tokens)
  float decay_total = 0.f;
  auto prep_chunk = [&](int c) {
    const int ab = c & 1;
    const int t = c * MQ + lane;
    float d = 0.f;
    if (t < L) {
      const unsigned w = sm.dtr[c & 3][lane];
      d = __uint_as_float((h & 1) ? (w & 0xffff0000u) : (w << 16)) + bias;
      if (a.softplus) d = softplus_f(d);
      d = fminf(fmaxf(d, a.dt_min), a.dt_max);
    }
    const float cs = wave_incl_scan_dpp(d * Ah);
    const float cl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), 63));
    sm.cs[ab][lane] = cs;
    sm.dtv[ab][lane] = d;
    sm.wts[ab][lane] = __expf(cl - cs) * d;
    if (lane == 0) sm.dlast[ab][0] = __expf(cl);
    decay_total += cl;
  };

  // ---- prologue: chunks 0 and 1 in flight, chunk 0 prepared ----
  if (wave == 6) glds4(dtg + min((int64_t)lane, (int64_t)L - 1) * a.dsl, lds_addr_of(sm.dtr[0]));
  issue(0);
  if (nchunks > 1) issue(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wave == 0) prep_chunk(0);
  __syncthreads();

  int yq[2], yc[2];                           // (row, 16-byte piece) of this lane's y stores
#pragma unroll
  for (int k = 0; k < 2; ++k) { yq[k] = (lane + 64 * k) / npc; yc[k] = (lane + 64 * k) - yq[k] * npc; }
  for (int c = 0; c < nchunks; ++c) {
    const int ab = c & 1;
    const MarchSlot& s = sm.slot[c % NSLOT];
    const bool do_issue = (c + 2 < nchunks) && !(a.dbg & 2);
    if (do_issue) issue(c + 2);

    const bool skip_y = (a.dbg & 1) || (a.dbg & 16), skip_s = (a.dbg & 1) || (a.dbg & 32);
    // ================= phase 1: Yoff (y-waves) | CB^T -> M (state-waves) ==================
    f32x4 yoff[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) yoff[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ywave) {
      if (!skip_y) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 cf = row_frag_swz(s.Ct, 16 * wave, 32 * ks, lc, kq);
#pragma unroll
          for (int j = 0; j < PT; ++j) {
            const bf16x8 sf = *(const bf16x8*)(sm.S + (16 * j + lc) * SSTR + 32 * ks + 8 * kq);
            yoff[j] = mfma16(sf, cf, yoff[j]);
          }
        }
      }
    } else if (!skip_s) {
      // causal CB^T tiles (t-tile, s-tile): sw0 (3,0)(3,1)(0,0) | sw1 (3,2)(3,3)(1,1) |
      // sw2 (2,0)(2,1) | sw3 (2,2)(1,0): two tiles share the C rows, chains interleave
      const int tA = sw < 2 ? 3 : 2;
      const int sA0 = sw == 0 ? 0 : sw == 1 ? 2 : sw == 2 ? 0 : 2;
      const int sA1 = sw == 3 ? -1 : sA0 + 1;             // second tile of the shared row
      const int tB = sw == 0 ? 0 : 1;                      // the odd tile (sw2 has none)
      const int sB = sw == 3 ? 0 : sw;                     // (0,0) (1,1) - (1,0)
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 cfa = row_frag_swz(s.Ct, 16 * tA, 32 * ks, lc, kq);
        c0 = mfma16(row_frag_swz(s.Bt, 16 * sA0, 32 * ks, lc, kq), cfa, c0);
        if (sA1 >= 0) c1 = mfma16(row_frag_swz(s.Bt, 16 * sA1, 32 * ks, lc, kq), cfa, c1);
        if (sw != 2)
          c2 = mfma16(row_frag_swz(s.Bt, 16 * sB, 32 * ks, lc, kq),
                      row_frag_swz(s.Ct, 16 * tB, 32 * ks, lc, kq), c2);
      }
      auto emit = [&](const f32x4& acc, int ti, int si) {
        // acc[r] = CB^T[s = 16si + 4kq + r][t = 16ti + lc]  ->  M[t][s] (bf16)
        const int t = 16 * ti + lc, s0 = 16 * si + 4 * kq;
        const float cst = sm.cs[ab][t];
        const f32x4 css = *(const f32x4*)(&sm.cs[ab][s0]);
        const f32x4 dts = *(const f32x4*)(&sm.dtv[ab][s0]);
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __expf(fminf(cst - css[r], 0.f));
          o[r] = (s0 + r <= t) ? (bf16_t)(acc[r] * e * dts[r]) : (bf16_t)0.f;
        }
        *(bf16x4*)(sm.M + t * MSTR + s0) = o;
      };
      emit(c0, tA, sA0);
      if (sA1 >= 0) emit(c1, tA, sA1);
      if (sw != 2) emit(c2, tB, sB);
    }
    // ---- barrier A: M visible; every Yoff read of S is done ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ================= phase 2: Ydiag + epilogue (y-waves) | state update (state-waves) ====
    if (ywave) {
      if (!skip_y) {
        const int t = 16 * wave + lc;
        f32x4 yd[PT];
#pragma unroll
        for (int j = 0; j < PT; ++j) yd[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (ks == 0 || wave >= 2) {   // s > t tiles of M are zero and never needed
            const bf16x8 mf = *(const bf16x8*)(sm.M + t * MSTR + 32 * ks + 8 * kq);   // k = s
#pragma unroll
            for (int j = 0; j < PT; ++j) {
              const int r0 = 32 * ks + 8 * kq;
              const bf16x8 xf = tr_frag_rows(s.xt, pw, r0, r0 + 4, 16 * j, lane);     // rows p
              yd[j] = mfma16(xf, mf, yd[j]);
            }
          }
        }
        // y^T[p = 16j + 4kq + r][t]
        const float e = __expf(sm.cs[ab][t]);
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const int p0 = 16 * j + 4 * kq;
          const bf16x4 xv = *(const bf16x4*)(s.xt + t * pw + p0);
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            o[r] = (bf16_t)(yd[j][r] + e * yoff[j][r] + Dh * (float)xv[r]);
          *(bf16x4*)(sm.yt + t * YSTR + p0) = o;
        }
      }
      // wave 0 prepares dt / cumsum of the NEXT chunk (its dt landed a step ago)
      if (wave == 0 && c + 1 < nchunks && !(a.dbg & 1)) prep_chunk(c + 1);
    } else if (!skip_s) {
      // X[n][p], n in [32sw, 32sw+32)
      const float dl = sm.dlast[ab][0];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) xacc[i][j][r] *= dl;
#pragma unroll
      for (int ks = 0; ks < MQ / 32; ++ks) {
        const f32x4 w0 = *(const f32x4*)(&sm.wts[ab][32 * ks + 8 * kq]);
        const f32x4 w1 = *(const f32x4*)(&sm.wts[ab][32 * ks + 8 * kq + 4]);
        bf16x8 xs[PT];
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const int r0 = 32 * ks + 8 * kq;
          const bf16x8 xf = tr_frag_rows(s.xt, pw, r0, r0 + 4, 16 * j, lane);   // k = t, cols p
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xs[j][e] = (bf16_t)((float)xf[e] * w0[e]);
            xs[j][4 + e] = (bf16_t)((float)xf[4 + e] * w1[e]);
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 bf = tr_frag_swz(s.Bt, 32 * ks, 16 * (2 * sw + i), lane);   // rows n, k = t
#pragma unroll
          for (int j = 0; j < PT; ++j) xacc[i][j] = mfma16(bf, xs[j], xacc[i][j]);
        }
      }
      // publish the new state (bf16) for the next chunk's Yoff: all reads of S finished
      // before barrier A
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const int p = 16 * j + lc, n = 16 * (2 * sw + i) + 4 * kq;
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)xacc[i][j][r];
          *(bf16x4*)(sm.S + p * SSTR + n) = o;
        }
    }
    // ---- barrier B: chunk c+1 has landed, S / cs visible.  Exactly 5 vm ops are younger
    // than chunk c+1's DMA when a group was issued this step (its 5 copies; the y stores of
    // the previous step are older than those) ----
    if (a.dbg & 2) {}   // ablation: nothing in flight
    else if (!do_issue) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- coalesced y store: each y-wave streams out the 16 rows it produced (whole rows,
    // 16 bytes per lane; no other wave touches them, so no barrier is needed) ----
    if (ywave && !(a.dbg & 4)) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = lane + 64 * k;
        if (i < 16 * npc) {
          const int row = 16 * wave + yq[k], ch = yc[k];
          const int t = c * MQ + row;
          if (t < L) {
            bf16_t* dstp = (a.dbg & 8) ? a.y + ((int64_t)(b * gridDim.x + blockIdx.x) * L + t) * pw + ch * 8
                                       : yg + (int64_t)t * a.ysl + ch * 8;
            *(bf16x8*)dstp = *(const bf16x8*)(sm.yt + row * YSTR + ch * 8);
          }
        }
      }
    }
  }

  if (!ywave && a.final_state) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        const int p = 16 * j + lc, n = 16 * (2 * sw + i) + 4 * kq;
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
      *nslices = ns;
      *pw = w;
      return true;
    }
  }
  return false;
}

}  // namespace

bool tv_ssd_march_supported(int seqlen, int nheads, int headdim, int ngroups, int dstate,
                            int dtype, int64_t xsl, int64_t bsl, int64_t bsg, int64_t csl, int64_t csg, int64_t ysl,
                            const void* x, const void* Bm, const void* Cm, const void* y) {
  int ns, pw;
  if (dtype != TV_BF16 || dstate != MN || seqlen < 1) return false;
  if (!pick_slices(headdim, &ns, &pw)) return false;
  if (xsl % 8 || bsl % 8 || csl % 8 || bsg % 8 || csg % 8 || ysl % 8 || nheads % 2) return false;
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
                        int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg, int64_t ysb, int64_t ysl,
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
  a.xsb = xsb; a.xsl = xsl; a.dsb = dsb; a.dsl = dsl; a.bsb = bsb; a.bsl = bsl; a.bsg = bsg;
  a.csb = csb; a.csl = csl; a.csg = csg; a.ysb = ysb; a.ysl = ysl;
  a.softplus = dt_softplus; a.group_map = group_map; a.dt_min = dt_min; a.dt_max = dt_max;
  { const char* e = getenv("TV_MARCH_DBG"); a.dbg = e ? atoi(e) : 0; }
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
