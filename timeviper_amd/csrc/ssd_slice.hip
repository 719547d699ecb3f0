// S3 (fast path, v2): Mamba-2 SSD selective scan as a "slice march" on CDNA4.
//
// Same single pass over HBM as ssd_march.hip (x, dt, B, C read once, y written once, the
// running state never leaves the chip), reorganised around what bounded that kernel: LDS
// operand traffic and the publish -> barrier -> re-read of the state every chunk.
//
//   * A workgroup = (batch, head, <=48-column slice of head_dim), 12 waves, one barrier per
//     64-token chunk.
//   * Slice-waves (one per 16 columns) keep X[n][16 cols] in MFMA accumulators for the whole
//     sequence and use those accumulators DIRECTLY as the B operand of Yoff = C.X (the k
//     index of an MFMA may be permuted freely as long as both operands agree, so C is read
//     with the permutation the accumulator layout dictates): the state is never written to
//     LDS.  Per chunk a slice-wave issues 16 (Yoff) + 16 (state update) + 6 (Ydiag) MFMAs
//     16x16x32 and touches only fragments.
//   * C.B^T is the same for the 32 workgroups of a B/C group: ssd_cb_kernel computes it once
//     per (chunk, group) — straight from global memory into MFMA fragments, no LDS — and
//     stores it in A-operand fragment order (6 causal fragments, 6 KiB per chunk and group,
//     bf16).  Mask waves turn it into M = CB .* exp(cs_t - cs_s) dt_s [s<=t] in LDS.
//   * Helper waves do everything that is not a state-dependent MFMA, one chunk or more
//     ahead: LDS-DMA of B/C (ring of 3, conflict-free swizzles for the transposing and the
//     row reads), of x and dt (ring of 4), dt -> softplus -> DPP prefix sum, x~ = w_t x,
//     the decay mask, and the coalesced y stores.
// Decay factors are only formed as exp(cs_i - cs_j), i >= j, inside a chunk (no quotient of
// exponentials), like the reference's segment_sum (modeling_nano.py:159-186).
// Reference semantics: mamba_chunk_scan_combined call modeling_nano.py:639-653; arithmetic
// :775-851.
#include <stdlib.h>
#include "ssd_common.hpp"

namespace {
using namespace ssdk;

constexpr int SQ = 64;           // tokens per chunk
constexpr int SN = 128;          // d_state
constexpr int STHREADS = 768;    // 12 waves
constexpr int NB = 3;            // B/C ring slots (prefetch distance 2 chunks)
constexpr int DXS = 3;           // x prefetch distance (chunks); dt runs one chunk further
constexpr int NXS = DXS + 1;     // x ring slots
constexpr int NDT = NXS + 2;     // raw-dt ring slots
constexpr int NV = 3;            // cs / ecs / dt / weight vector buffers
constexpr int NFRAG = 6;         // causal (t-tile, s-pair) fragments of a 64x64 chunk
constexpr int CB_ELEMS = NFRAG * 512;   // bf16 elements per (chunk, group)

// fragment f -> (t-tile, s-pair): (0,0) (1,0) (2,0) (2,1) (3,0) (3,1)
__device__ __forceinline__ int frag_ti(int f) { return f == 0 ? 0 : f == 1 ? 1 : f < 4 ? 2 : 3; }
__device__ __forceinline__ int frag_sp(int f) { return (f == 3 || f == 5) ? 1 : 0; }

// ------------------------------------------------------------------ C.B^T pre-pass
struct CbArgs {
  const bf16_t *Bm, *Cm;
  bf16_t* cb;
  int L, G, nchunks;
  int64_t bsb, bsl, bsg, csb, csl, csg;
};

// grid (nchunks, G, batch), 3 waves, 2 fragments each.  Fragment (ti, sp) holds, for lane
// (t = 16ti + lane%16, kq = lane/16), CB[t][s = 32sp + 8kq + 0..7]: the A operand of
// Ydiag[t][p] = sum_s M[t][s] x[s][p].  It is produced as two transposed tiles CB^T[s][t]
// whose s rows are chosen (rows of an A operand loaded from global memory can be any rows)
// so that each lane's eight accumulator values are those eight consecutive s.
__global__ __launch_bounds__(192) void ssd_cb_kernel(CbArgs a) {
  const int c = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lc = lane & 15, kq = lane >> 4;
  const int t0 = c * SQ;
  const int rmax = a.L - 1 - t0;   // rows past the sequence end repeat the last row (finite)
  const bf16_t* Bg = a.Bm + (int64_t)b * a.bsb + (int64_t)g * a.bsg + (int64_t)t0 * a.bsl;
  const bf16_t* Cg = a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg + (int64_t)t0 * a.csl;
#pragma unroll
  for (int ff = 0; ff < 2; ++ff) {
    const int f = 2 * wave + ff;
    const int ti = frag_ti(f), sp = frag_sp(f);
    const int tr = min(16 * ti + lc, rmax);
    const int s_lo = 32 * sp + 8 * (lc >> 2) + (lc & 3);
    const bf16_t* cp = Cg + (int64_t)tr * a.csl + 8 * kq;
    const bf16_t* bp0 = Bg + (int64_t)min(s_lo, rmax) * a.bsl + 8 * kq;
    const bf16_t* bp1 = Bg + (int64_t)min(s_lo + 4, rmax) * a.bsl + 8 * kq;
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
    bf16x8 cf[4], b0[4], b1[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      cf[ks] = *(const bf16x8*)(cp + 32 * ks);
      b0[ks] = *(const bf16x8*)(bp0 + 32 * ks);
      b1[ks] = *(const bf16x8*)(bp1 + 32 * ks);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      d0 = mfma16(b0[ks], cf[ks], d0);
      d1 = mfma16(b1[ks], cf[ks], d1);
    }
    bf16x8 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[r] = (bf16_t)d0[r];
      o[4 + r] = (bf16_t)d1[r];
    }
    bf16_t* dst = a.cb + ((((int64_t)b * a.G + g) * a.nchunks + c) * NFRAG + f) * 512 + lane * 8;
    *(bf16x8*)dst = o;
  }
}

// ------------------------------------------------------------------ main kernel
struct SliceArgs {
  const bf16_t *x, *dt, *Bm, *Cm, *cb;
  const float *A, *D, *dt_bias, *init;
  bf16_t* y;
  float *final_state, *total_decay;
  int L, H, P, G, nslices, pw, nchunks;
  int64_t xsb, xsl, dsb, dsl, bsb, bsl, bsg, csb, csl, csg, ysb, ysl;
  int softplus, group_map;
  float dt_min, dt_max;
};

template <int PW>
struct __attribute__((aligned(16))) SliceSmem {
  static constexpr int XSLOT = SQ * PW + 64;   // + finite guard (the last tile reads past PW)
  bf16_t bt[NB][SQ * SN];     // B tiles [t][n], 16-byte chunks XOR-swizzled for ds_read_b64_tr
  bf16_t ct[NB][SQ * SN];     // C tiles [t][n], chunks XOR-swizzled for row reads
  bf16_t xr[NXS][XSLOT];      // x tiles [t][PW]
  bf16_t xs[2][XSLOT];        // x~ = exp(cs_Q - cs_t) dt_t x
  bf16_t M[2][CB_ELEMS];      // decay-masked C.B^T fragments
  bf16_t yt[2][SQ * PW];      // y tiles [t][PW]
  unsigned dtr[NDT][SQ];      // raw dt of heads (h&~1, h|1)
  unsigned pad_[SQ];
  float cs[NV][SQ];           // inclusive cumsum of dt*A inside the chunk
  float ecs[NV][SQ];          // exp(cs)
  float dtv[NV][SQ];          // discretised dt
  float wts[NV][SQ];          // exp(cs_last - cs_t) * dt_t
  float dl[NV][4];            // exp(cs_last)
};

// wave roles
constexpr int W_XIO = 3;                         // x / dt DMA + y stores
__device__ __forceinline__ int bc_index(int w) { return w == 4 ? 0 : w == 5 ? 1 : w == 6 ? 2 : w == 8 ? 3 : -1; }
__device__ __forceinline__ int mask_index(int w) { return w == 7 ? 0 : w == 11 ? 1 : -1; }
__device__ __forceinline__ int scale_index(int w) { return w == 9 ? 0 : w == 10 ? 1 : -1; }

#define SLICE_BARRIER()                                   \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
  } while (0)

template <int PT, int PW>
__global__ __launch_bounds__(STHREADS) void ssd_slice_kernel(SliceArgs a) {
  typedef SliceSmem<PW> Smem;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
  constexpr int NPC = PW / 8;          // 16-byte pieces per x / y row
  constexpr int NPI = PW / 8;          // wave-instructions per x / y tile (64 rows * NPC / 64)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, kq = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int b = blockIdx.y;
  const int hpg = a.H / a.G;
  const int g = blockIdx.x % a.G;
  const int rest = blockIdx.x / a.G;
  const int hig = rest / a.nslices, slice = rest % a.nslices;
  const int h = a.group_map ? (hig * a.G + g) : (g * hpg + hig);
  const int p_base = slice * PW;
  const int L = a.L, nchunks = a.nchunks;

  {   // zero LDS once: guards / pad columns must hold finite values
    bf16x8 z = {};
    for (int i = tid; i < (int)(sizeof(Smem) / 16); i += STHREADS)
      reinterpret_cast<bf16x8*>(smem_raw)[i] = z;
  }
  __syncthreads();

  // The prologue fills the pipeline with three barriers (P1: first tiles landed, P2: chunks
  // 0/1 prepared, P3: x~_0 and M_0 built); then every role runs nchunks steps with one
  // barrier each.  At step c the slice-waves consume chunk c while the helpers produce
  // x~_{c+1}, M_{c+1}, the vectors of chunk c+2, issue B/C of chunk c+2, x of chunk c+3,
  // dt of chunk c+4 and store y_{c-1}.
  if (wave < PT) {
    // ============================================================ slice-wave (16 columns)
    const int j = wave;
    const float Dh = a.D ? a.D[h] : 0.f;
    f32x4 xacc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pcol = 16 * j + lc;
    const bool pvalid = pcol < PW;
    if (a.init && pvalid) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        xacc[i] = *(const f32x4*)(a.init + (((int64_t)b * a.H + h) * a.P + p_base + pcol) * SN + 16 * i + 4 * kq);
    }
    // C row fragments with the accumulator's k order: slots 0..3 = n 32m+4kq.., 4..7 = +16
    const int c_lo = lc * 256 + (kq & 1) * 8;
    int c_sw[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      c_sw[m][0] = c_lo + ((((4 * m + (kq >> 1)) ^ lc)) << 4);
      c_sw[m][1] = c_lo + ((((4 * m + 2 + (kq >> 1)) ^ lc)) << 4);
    }
    // transposing reads of B: row t = 32ks + 8kq + q4 (+4), chunk (2i + p4/2) ^ s(t)
    const int bsw = (2 * q4 + 8 * (kq & 1)) << 4;
    const int b_lo = (8 * kq + q4) * 256 + (p4 >> 1) * 16 + (p4 & 1) * 8;
    // transposing reads of x / x~ (B operand: k = token) and of x in accumulator layout
    const int trx = ((8 * kq + q4) * PW + 4 * p4) * 2 + j * 32;
    const int trd = ((4 * kq + q4) * PW + 4 * p4) * 2 + j * 32;
    SLICE_BARRIER();   // P1
    SLICE_BARRIER();   // P2
    SLICE_BARRIER();   // P3
    for (int c = 0; c < nchunks; ++c) {
      const int vb = c % NV;
      const unsigned char* Bt = reinterpret_cast<const unsigned char*>(sm.bt[c % NB]);
      const unsigned char* Ct = reinterpret_cast<const unsigned char*>(sm.ct[c % NB]);
      const unsigned char* xt = reinterpret_cast<const unsigned char*>(sm.xr[c % NXS]);
      const unsigned char* xw = reinterpret_cast<const unsigned char*>(sm.xs[c & 1]);
      const unsigned char* Mf = reinterpret_cast<const unsigned char*>(sm.M[c & 1]);
      // state at the chunk start as B operands (k = n in accumulator order)
      bf16x8 sb[4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sb[m][r] = (bf16_t)xacc[2 * m][r];
          sb[m][4 + r] = (bf16_t)xacc[2 * m + 1][r];
        }
      // x fragments (k = token), raw and weighted
      bf16x8 xf[2], xwf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        xf[ks] = cat4(tr4(xt + trx + ks * (32 * PW * 2)), tr4(xt + trx + ks * (32 * PW * 2) + 4 * PW * 2));
        xwf[ks] = cat4(tr4(xw + trx + ks * (32 * PW * 2)), tr4(xw + trx + ks * (32 * PW * 2) + 4 * PW * 2));
      }
      // X = exp(cs_Q) X + B^T x~
      const float dl = sm.dl[vb][0];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) xacc[i][r] *= dl;
        const int bo = b_lo + ((32 * i) ^ bsw);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 bf = cat4(tr4(Bt + bo + ks * 8192), tr4(Bt + bo + ks * 8192 + 1024));
          xacc[i] = mfma16(bf, xwf[ks], xacc[i]);
        }
      }
      // y tile: Yoff (state at chunk start) + Ydiag + D x
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        f32x4 yo = {0.f, 0.f, 0.f, 0.f}, yd = yo;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
          const u32x2 lo = *(const u32x2*)(Ct + ti * 4096 + c_sw[m][0]);
          const u32x2 hi = *(const u32x2*)(Ct + ti * 4096 + c_sw[m][1]);
          typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
          const u32x4 w = {lo[0], lo[1], hi[0], hi[1]};
          yo = mfma16(__builtin_bit_cast(bf16x8, w), sb[m], yo);
        }
        const int f0 = ti == 0 ? 0 : ti == 1 ? 1 : ti == 2 ? 2 : 4;
        yd = mfma16(ld8(Mf + f0 * 1024 + lane * 16), xf[0], yd);
        if (ti >= 2) yd = mfma16(ld8(Mf + (f0 + 1) * 1024 + lane * 16), xf[1], yd);
        const f32x4 e = *(const f32x4*)(&sm.ecs[vb][16 * ti + 4 * kq]);
        const bf16x4 xv = tr4(xt + trd + ti * (16 * PW * 2));
        if (pvalid) {
          bf16_t* yrow = sm.yt[c & 1] + (16 * ti + 4 * kq) * PW + pcol;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            yrow[r * PW] = (bf16_t)(yd[r] + e[r] * yo[r] + Dh * (float)xv[r]);
        }
      }
      SLICE_BARRIER();
    }
    if (a.final_state && pvalid) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        *(f32x4*)(a.final_state + (((int64_t)b * a.H + h) * a.P + p_base + pcol) * SN + 16 * i + 4 * kq) = xacc[i];
    }
    SLICE_BARRIER();   // final (y of the last chunk is stored after it)
  } else if (wave == W_XIO) {
    // ============================================================ x / dt DMA, y stores
    const bf16_t* xg = a.x + (int64_t)b * a.xsb + (int64_t)h * a.P + p_base;
    const bf16_t* dtg = a.dt + (int64_t)b * a.dsb + (h & ~1);
    bf16_t* yg = a.y + (int64_t)b * a.ysb + (int64_t)h * a.P + p_base;
    unsigned x_off[NPI];
    int64_t y_off[NPI];
    int prow[NPI];
#pragma unroll
    for (int k = 0; k < NPI; ++k) {
      const int i = lane + 64 * k;
      prow[k] = i / NPC;
      x_off[k] = (unsigned)((prow[k] * a.xsl + (i % NPC) * 8) * 2);
      y_off[k] = (int64_t)prow[k] * a.ysl + (i % NPC) * 8;
    }
    auto issue_x = [&](int c) {          // NPI x pieces of chunk c
      const int t0 = c * SQ;
      const void* sx = uniform_ptr(xg + (int64_t)t0 * a.xsl);
      const bool full = t0 + SQ <= L;
#pragma unroll
      for (int k = 0; k < NPI; ++k) {
        unsigned o = x_off[k];
        if (!full) o = (unsigned)((min(prow[k], L - 1 - t0) * a.xsl + ((lane + 64 * k) % NPC) * 8) * 2);
        glds16(sx, o, lds_addr_of(sm.xr[c % NXS] + 512 * k));
      }
    };
    auto issue_dt = [&](int c) {         // one piece: raw dt of chunk c (rows clamped to L-1)
      const int t0 = min(c * SQ, L - 1);
      const void* sd = uniform_ptr(dtg + (int64_t)t0 * a.dsl);
      const unsigned od = (unsigned)(min(lane, L - 1 - t0) * a.dsl * 2);
      glds4(sd, od, lds_addr_of(sm.dtr[c % NDT]));
    };
    auto store_y = [&](int c) {
      const int t0 = c * SQ;
      const bool full = t0 + SQ <= L;
      const unsigned char* ytb = reinterpret_cast<const unsigned char*>(sm.yt[c & 1]);
      bf16_t* yc = yg + (int64_t)t0 * a.ysl;
#pragma unroll
      for (int k = 0; k < NPI; ++k)
        if (full || t0 + prow[k] < L) *(bf16x8*)(yc + y_off[k]) = *(const bf16x8*)(ytb + (lane + 64 * k) * 16);
    };
    for (int c = 0; c < DXS; ++c) issue_x(min(c, nchunks - 1));
    for (int c = 0; c <= DXS; ++c) issue_dt(c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SLICE_BARRIER();   // P1
    SLICE_BARRIER();   // P2
    SLICE_BARRIER();   // P3
    for (int c = 0; c < nchunks; ++c) {
      if (c > 0) store_y(c - 1);
      const bool issued = c + DXS < nchunks;
      if (issued) {
        issue_x(c + DXS);
        issue_dt(c + DXS + 1);
      }
      // x of chunk c+2 and dt of chunk c+3 (issued a step ago) must have landed: since then
      // this wave issued NPI stores + NPI + 1 copies
      if (issued && c > 0 && (c + 1) * SQ <= L) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPI + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SLICE_BARRIER();
    }
    SLICE_BARRIER();   // final: the last chunk's y tile is complete
    store_y(nchunks - 1);
  } else if (bc_index(wave) >= 0) {
    // ============================================================ B / C DMA
    const int q = bc_index(wave);
    const bf16_t* Bg = a.Bm + (int64_t)b * a.bsb + (int64_t)g * a.bsg;
    const bf16_t* Cg = a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg;
    int brow[4];
    unsigned off_b[4], off_c[4];
    int cg_b[4], cg_c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = 16 * q + 4 * k + (lane >> 4);
      brow[k] = row;
      cg_b[k] = (lane & 15) ^ (2 * (row & 3) + 8 * ((row >> 3) & 1));
      cg_c[k] = (lane & 15) ^ (row & 15);
      off_b[k] = (unsigned)((row * a.bsl + cg_b[k] * 8) * 2);
      off_c[k] = (unsigned)((row * a.csl + cg_c[k] * 8) * 2);
    }
    auto issue_bc = [&](int c) {
      const int slot = c % NB;
      const int t0 = c * SQ;
      const void* sb = uniform_ptr(Bg + (int64_t)t0 * a.bsl);
      const void* sc = uniform_ptr(Cg + (int64_t)t0 * a.csl);
      const bool full = t0 + SQ <= L;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        unsigned ob = off_b[k], oc = off_c[k];
        if (!full) {
          const int rr = min(brow[k], L - 1 - t0);
          ob = (unsigned)((rr * a.bsl + cg_b[k] * 8) * 2);
          oc = (unsigned)((rr * a.csl + cg_c[k] * 8) * 2);
        }
        glds16(sb, ob, lds_addr_of(sm.bt[slot] + (4 * q + k) * 512));
        glds16(sc, oc, lds_addr_of(sm.ct[slot] + (4 * q + k) * 512));
      }
    };
    issue_bc(0);
    if (nchunks > 1) issue_bc(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SLICE_BARRIER();   // P1
    SLICE_BARRIER();   // P2
    SLICE_BARRIER();   // P3
    for (int c = 0; c < nchunks; ++c) {
      const bool issued = c + 2 < nchunks;
      if (issued) {
        issue_bc(c + 2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // chunk c+1 landed, c+2 in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      SLICE_BARRIER();
    }
    SLICE_BARRIER();   // final
  } else if (mask_index(wave) >= 0) {
    // ============================================================ decay mask M = CB .* L
    const int mi = mask_index(wave);
    const bf16_t* cbg = a.cb + (((int64_t)b * a.G + g) * nchunks) * CB_ELEMS + lane * 8;
    bf16x8 cbv[3];
    auto load_cb = [&](int c) {
#pragma unroll
      for (int ff = 0; ff < 3; ++ff) cbv[ff] = *(const bf16x8*)(cbg + (int64_t)c * CB_ELEMS + (3 * mi + ff) * 512);
    };
    auto build = [&](int c) {
      const int vb = c % NV;
#pragma unroll
      for (int ff = 0; ff < 3; ++ff) {
        const int f = 3 * mi + ff;
        const int t = 16 * frag_ti(f) + lc, s0 = 32 * frag_sp(f) + 8 * kq;
        const float cst = sm.cs[vb][t];
        const f32x4 ca = *(const f32x4*)(&sm.cs[vb][s0]), cb2 = *(const f32x4*)(&sm.cs[vb][s0 + 4]);
        const f32x4 da = *(const f32x4*)(&sm.dtv[vb][s0]), db = *(const f32x4*)(&sm.dtv[vb][s0 + 4]);
        bf16x8 o;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const float css = jj < 4 ? ca[jj & 3] : cb2[jj & 3];
          const float dts = jj < 4 ? da[jj & 3] : db[jj & 3];
          const float e = __expf(fminf(cst - css, 0.f));
          o[jj] = (s0 + jj <= t) ? (bf16_t)((float)cbv[ff][jj] * e * dts) : (bf16_t)0.f;
        }
        *(bf16x8*)(sm.M[c & 1] + f * 512 + lane * 8) = o;
      }
    };
    load_cb(0);
    SLICE_BARRIER();   // P1
    SLICE_BARRIER();   // P2 (chunks 0/1 prepared)
    build(0);
    if (nchunks > 1) load_cb(1);
    SLICE_BARRIER();   // P3
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) build(c + 1);
      if (c + 2 < nchunks) load_cb(c + 2);
      SLICE_BARRIER();
    }
    SLICE_BARRIER();   // final
  } else if (scale_index(wave) >= 0) {
    // ============================================================ x~ tiles, dt / cumsum prep
    const int xi = scale_index(wave);
    constexpr int KA = (NPI + 1) / 2;
    const int k0 = xi == 0 ? 0 : KA, k1 = xi == 0 ? KA : NPI;
    const float Ah = a.A[h];
    const float bias = a.dt_bias ? a.dt_bias[h] : 0.f;
    float decay_total = 0.f;
    auto scale_x = [&](int c) {
      const int vb = c % NV;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(sm.xr[c % NXS]);
      unsigned char* dst = reinterpret_cast<unsigned char*>(sm.xs[c & 1]);
      for (int k = k0; k < k1; ++k) {
        const int i = lane + 64 * k;
        const float w = sm.wts[vb][i / NPC];
        const uint4 v = *(const uint4*)(src + i * 16);
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[2 * e] = (bf16_t)(bf16_lo(u[e]) * w);
          o[2 * e + 1] = (bf16_t)(bf16_hi(u[e]) * w);
        }
        *(bf16x8*)(dst + i * 16) = o;
      }
    };
    auto prep = [&](int c) {             // one wave: lane = token
      const int vb = c % NV;
      const int t = c * SQ + lane;
      float d = 0.f;
      if (t < L) {
        const unsigned w = sm.dtr[c % NDT][lane];
        d = ((h & 1) ? bf16_hi(w) : bf16_lo(w)) + bias;
        if (a.softplus) d = softplus_fast(d);
        d = fminf(fmaxf(d, a.dt_min), a.dt_max);
      }
      const float cs = wave_incl_scan_dpp(d * Ah);
      const float cl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), 63));
      sm.cs[vb][lane] = cs;
      sm.ecs[vb][lane] = __expf(cs);
      sm.dtv[vb][lane] = d;
      sm.wts[vb][lane] = __expf(cl - cs) * d;
      if (lane == 0) sm.dl[vb][0] = __expf(cl);
      decay_total += cl;
    };
    SLICE_BARRIER();   // P1
    if (xi == 1) {
      prep(0);
      if (nchunks > 1) prep(1);
    }
    SLICE_BARRIER();   // P2
    scale_x(0);
    SLICE_BARRIER();   // P3
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) scale_x(c + 1);
      if (xi == 1 && c + 2 < nchunks) prep(c + 2);
      SLICE_BARRIER();
    }
    SLICE_BARRIER();   // final
    if (xi == 1 && a.total_decay && slice == 0 && lane == 0) a.total_decay[(int64_t)b * a.H + h] = decay_total;
  } else {
    // idle waves (slice-wave slots of narrower slices)
    SLICE_BARRIER();
    SLICE_BARRIER();
    SLICE_BARRIER();
    for (int c = 0; c < nchunks; ++c) SLICE_BARRIER();
    SLICE_BARRIER();
  }
}

bool pick_slices(int P, int* nslices, int* pw) {
  for (int ns = 1; ns <= 8; ++ns) {
    if (P % ns) continue;
    const int w = P / ns;
    if (w <= 40 && w % 8 == 0) {   // 48-column slices would not fit the LDS budget
      *nslices = ns;
      *pw = w;
      return true;
    }
  }
  return false;
}

template <int PT, int PW>
hipError_t launch_slice(const SliceArgs& a, dim3 grid, hipStream_t st) {
  const size_t lds = sizeof(SliceSmem<PW>);
  static_assert(sizeof(SliceSmem<PW>) <= 160 * 1024, "LDS budget");
  hipError_t e = hipFuncSetAttribute((const void*)ssd_slice_kernel<PT, PW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  ssd_slice_kernel<PT, PW><<<grid, STHREADS, lds, st>>>(a);
  return hipSuccess;
}

}  // namespace

bool tv_ssd_slice_supported(int seqlen, int nheads, int headdim, int ngroups, int dstate,
                            int dtype, int64_t xsl, int64_t bsl, int64_t bsg, int64_t csl,
                            int64_t csg, int64_t ysl, const void* x, const void* Bm,
                            const void* Cm, const void* y) {
  int ns, pw;
  if (dtype != TV_BF16 || dstate != SN || seqlen < 1) return false;
  if (!pick_slices(headdim, &ns, &pw)) return false;
  if (xsl % 8 || bsl % 8 || csl % 8 || bsg % 8 || csg % 8 || ysl % 8 || nheads % 2) return false;
  if (((uintptr_t)x & 15) || ((uintptr_t)Bm & 15) || ((uintptr_t)Cm & 15) || ((uintptr_t)y & 15))
    return false;
  if (64 * xsl * 2 >= (1ll << 31) || 64 * bsl * 2 >= (1ll << 31) || 64 * csl * 2 >= (1ll << 31))
    return false;
  (void)ngroups;
  return true;
}

size_t tv_ssd_slice_workspace_bytes(int batch, int seqlen, int, int, int ngroups, int) {
  const size_t nchunks = (size_t)(seqlen + SQ - 1) / SQ;
  return (size_t)batch * ngroups * nchunks * CB_ELEMS * sizeof(bf16_t);
}

int tv_ssd_slice_launch(const void* x, const void* dt, const void* A, const void* Bm,
                        const void* Cm, const void* D, const void* dt_bias,
                        const void* init_state, void* y, void* final_state, void* total_decay,
                        int batch, int seqlen, int nheads, int headdim, int ngroups, int dstate,
                        int64_t xsb, int64_t xsl, int64_t dsb, int64_t dsl, int64_t bsb,
                        int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg,
                        int64_t ysb, int64_t ysl, int dtype, int dt_softplus, float dt_min,
                        float dt_max, int group_map, void* workspace, size_t workspace_bytes,
                        hipStream_t st) {
  (void)dtype; (void)dstate;
  const size_t need = tv_ssd_slice_workspace_bytes(batch, seqlen, nheads, headdim, ngroups, dstate);
  TV_CHECK_ARG(workspace && workspace_bytes >= need && (((uintptr_t)workspace) & 15) == 0,
               "ssd_slice: workspace of %zu bytes (16-byte aligned) required, got %zu", need,
               workspace_bytes);
  SliceArgs a;
  a.x = (const bf16_t*)x; a.dt = (const bf16_t*)dt; a.Bm = (const bf16_t*)Bm; a.Cm = (const bf16_t*)Cm;
  a.cb = (const bf16_t*)workspace;
  a.A = (const float*)A; a.D = (const float*)D; a.dt_bias = (const float*)dt_bias;
  a.init = (const float*)init_state; a.y = (bf16_t*)y; a.final_state = (float*)final_state;
  a.total_decay = (float*)total_decay;
  a.L = seqlen; a.H = nheads; a.P = headdim; a.G = ngroups;
  a.nchunks = (seqlen + SQ - 1) / SQ;
  if (!pick_slices(headdim, &a.nslices, &a.pw)) TV_UNSUPPORTED("ssd_slice: head_dim %d", headdim);
  a.xsb = xsb; a.xsl = xsl; a.dsb = dsb; a.dsl = dsl; a.bsb = bsb; a.bsl = bsl; a.bsg = bsg;
  a.csb = csb; a.csl = csl; a.csg = csg; a.ysb = ysb; a.ysl = ysl;
  a.softplus = dt_softplus; a.group_map = group_map; a.dt_min = dt_min; a.dt_max = dt_max;

  CbArgs ca;
  ca.Bm = a.Bm; ca.Cm = a.Cm; ca.cb = (bf16_t*)workspace;
  ca.L = seqlen; ca.G = ngroups; ca.nchunks = a.nchunks;
  ca.bsb = bsb; ca.bsl = bsl; ca.bsg = bsg; ca.csb = csb; ca.csl = csl; ca.csg = csg;
  ssd_cb_kernel<<<dim3(a.nchunks, ngroups, batch), 192, 0, st>>>(ca);

  dim3 grid(nheads * a.nslices, batch);
  hipError_t e = hipSuccess;
  switch (a.pw) {
    case 8: e = launch_slice<1, 8>(a, grid, st); break;
    case 16: e = launch_slice<1, 16>(a, grid, st); break;
    case 24: e = launch_slice<2, 24>(a, grid, st); break;
    case 32: e = launch_slice<2, 32>(a, grid, st); break;
    case 40: e = launch_slice<3, 40>(a, grid, st); break;
    default: TV_UNSUPPORTED("ssd_slice: slice width %d", a.pw);
  }
  if (e != hipSuccess) {
    tv_set_error("ssd_slice: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return TV_ERR_LAUNCH;
  }
  TV_LAUNCH_CHECK();
}
