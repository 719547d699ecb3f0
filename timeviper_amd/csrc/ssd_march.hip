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
// Data movement.  Every input tile arrives by LDS-DMA (global_load_lds, 16 B per lane, no
// VGPR round trip; scalar base + per-lane 32-bit offset, so stepping to the next chunk is
// two scalar adds).  The B/C tiles (L2 hits, shared by the group's workgroups) are fetched
// by the state-waves 2 chunks ahead into a 3-slot ring; the x tiles (HBM, private) by the
// y-waves 4 chunks ahead into a 5-slot ring — vmcnt retires in order per wave, so the two
// latency classes live in different waves and the y stores cannot stall the B/C stream.
// Waits are counted (s_waitcnt vmcnt(N), never 0 in the steady state) and barriers are raw
// s_barrier, so the DMA stays in flight across them.  B/C tiles are stored unpadded and
// XOR-swizzled on the SOURCE address (chunk ^= row&15): the row-major (ds_read_b128) reads
// are bank-conflict free, the transposing (ds_read_b64_tr_b16) ones at most 2-way.
//
// Per chunk (all products on v_mfma_f32_16x16x32_bf16, fp32 accumulate):
//   y-waves 0..3 (16 tokens each)            state-waves 4..7 (32 state rows each)
//   -- phase 1 ---------------------------------------------------------------------------
//   Yoff^T[p][t] = sum_n S[p][n] C[t][n]     CB^T[s][t] = sum_n B[s][n] C[t][n]  (causal tiles)
//   x~[t][p] = exp(cs_Q-cs_t) dt_t x[t][p]   M[t][s] = CB^T exp(cs_t-cs_s) dt_s [s<=t] -> LDS
//   wave 0: dt = softplus(.), cs = DPP wave-prefix-sum(dt*A) of the NEXT chunk
//   -- barrier ---------------------------------------------------------------------------
//   Ydiag^T[p][t] = sum_s x[s][p] M[t][s]    X[n][p] = exp(cs_Q) X[n][p] + sum_t B[t][n] x~[t][p]
//   y = Ydiag + exp(cs_t) Yoff + D x  -> LDS -> 16-byte row stores        S = bf16(X) -> LDS
//   -- barrier (counted DMA waits) ---------------------------------------------------------
// Decay factors are only ever formed as exp(cs_i - cs_j) with i >= j inside one chunk
// (never a quotient of exponentials), like the reference's segment_sum
// (modeling_nano.py:159-186), so no input can overflow them.
// Reference semantics: mamba_chunk_scan_combined call modeling_nano.py:639-653;
// arithmetic :775-851.
#include <stdlib.h>
#include "ssd_common.hpp"

namespace {
using namespace ssdk;

constexpr int MQ = 64;          // tokens per chunk (= wavefront width)
constexpr int MN = 128;         // d_state
constexpr int MTHREADS = 512;
constexpr int SSTR = MN + 8;    // S tile row stride (elements): +16 B against bank conflicts
constexpr int YSTR = 48;        // y tile row stride
constexpr int MSTR = MQ + 8;    // M tile row stride (144 B: conflict-free b128 row reads)
constexpr int PMAX = 48;        // max head_dim columns per workgroup (3 MFMA tiles)
constexpr int NBC = 3;          // B/C ring slots  (prefetch distance 2 chunks)

// -DTV_MARCH_ABLATE builds the ablation switches (env TV_MARCH_DBG) into the kernel; the
// production build has none of those branches.
#ifdef TV_MARCH_ABLATE
#define DBG(a, bit) ((a).dbg & (bit))
#else
#define DBG(a, bit) 0
#endif

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

template <int PW>
struct __attribute__((aligned(16))) MarchSmem {
  static constexpr int DX = PW > 40 ? 3 : 4;   // x / dt prefetch distance (chunks)
  static constexpr int NXS = DX + 1;           // x ring slots
  static constexpr int XSLOT = MQ * PW + 64;   // elements; + finite guard (tile PT-1 reads past PW)
  bf16_t bc[NBC][2][MQ * MN];   // [slot][B|C], swizzled, 16 KiB each
  bf16_t xr[NXS][XSLOT];        // x tiles [64][PW] linear
  bf16_t xs[XSLOT];             // x~ = exp(cs_Q - cs_t) dt_t x, same layout
  bf16_t S[PMAX * SSTR];        // bf16 copy of the state, [p][n]
  bf16_t yt[MQ * YSTR];         // y tile
  bf16_t M[MQ * MSTR];          // decay-masked C.B^T of the chunk, [t][s]; s>t tiles stay 0
  unsigned dtr[NXS + 1][MQ];    // raw dt of heads (h&~1, h|1) per token, one chunk further ahead
  unsigned pad_[MQ];            // landing pad of the filler DMA
  float cs[2][MQ];              // inclusive cumsum of dt*A inside the chunk
  float dtv[2][MQ];             // discretised dt
  float wts[2][MQ];             // exp(cs_last - cs_t) * dt_t
  float dlast[2][4];            // exp(cs_last)
};

template <int PT, int PW>
__global__ __launch_bounds__(MTHREADS) void ssd_march_kernel(MarchArgs a) {
  typedef MarchSmem<PW> Smem;
  constexpr int DX = Smem::DX, NXS = Smem::NXS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
  constexpr int NPC = PW / 8;                     // 16-byte pieces per x / y row
  constexpr int YPIECES = 16 * NPC;               // pieces of the 16 rows a y-wave owns
  constexpr int NXI = (YPIECES + 63) / 64;        // x DMA (and y store) instructions per step
  constexpr int YWAIT = (DX - 1) * (NXI + 1 + NXI);   // see barrier B
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, kq = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int b = blockIdx.y;
  // blockIdx.x -> (group, head in group, slice): blocks with equal (blockIdx.x % G) share an
  // XCD under round-robin dispatch when G == 8 (speed only, never correctness)
  const int hpg = a.H / a.G;
  const int g = blockIdx.x % a.G;
  const int rest = blockIdx.x / a.G;
  const int hig = rest / a.nslices, slice = rest % a.nslices;
  const int h = a.group_map ? (hig * a.G + g) : (g * hpg + hig);
  const int p_base = slice * PW;
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

  // ---- zero LDS once: pad columns / guards / untouched M tiles must hold finite values ----
  {
    bf16x8 z = {};
    for (int i = tid; i < (int)(sizeof(Smem) / 16); i += MTHREADS)
      reinterpret_cast<bf16x8*>(smem_raw)[i] = z;
  }
  __syncthreads();

  const bool ywave = wave < 4;
  const int sw = wave & 3;     // index inside the wave's role group
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
        if (p < PW) {
          const f32x4 v = *(const f32x4*)(a.init + (((int64_t)b * a.H + h) * a.P + p_base + p) * MN + n);
          xacc[i][j] = v;
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
          *(bf16x4*)(sm.S + p * SSTR + n) = o;
        }
      }
  }

  // ------------------------------------------------------------------ per-lane constants
  // byte offsets inside a swizzled [64][128] bf16 tile: row_frag(r0, ks) = r0*256 + rfo[ks]
  int rfo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rfo[ks] = lc * 256 + ((((4 * ks + kq) ^ lc)) << 4);
  const int s_rd = (lc * SSTR + 8 * kq) * 2;                 // S tile A operand (+ j, ks)
  const int m_rd = (lc * MSTR + 8 * kq) * 2;                 // M tile B operand (+ t-tile, ks)
  const int m_wr = (lc * MSTR + 4 * kq) * 2;                 // M tile writes   (+ t-tile, s-tile)
  const int trx = ((8 * kq + q4) * PW + 4 * p4) * 2;         // tr-read of x / x~ (+ ks, hi, j)
  int trb[2][2];                                             // tr-read of B rows n (state-waves)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int r = 8 * kq + q4 + 4 * hh;                    // (+32 ks: same r & 15)
      const int cg = 2 * (2 * sw + i) + (p4 >> 1);
      trb[i][hh] = r * 256 + ((cg ^ (r & 15)) << 4) + (p4 & 1) * 8;
    }
  // DMA source offsets (bytes from the chunk's scalar base)
  //   state-waves: B and C pieces 4sw..4sw+3 (4 rows each)      y-waves: x rows 16w..16w+15
  unsigned bc_off_b[4], bc_off_c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = 4 * (4 * sw + k) + (lane >> 4);
    const int cg = (lane & 15) ^ (row & 15);
    bc_off_b[k] = (unsigned)((row * a.bsl + cg * 8) * 2);
    bc_off_c[k] = (unsigned)((row * a.csl + cg * 8) * 2);
  }
  unsigned x_off[NXI];
  bool x_act[NXI];
#pragma unroll
  for (int k = 0; k < NXI; ++k) {
    const int i = lane + 64 * k;
    x_act[k] = i < YPIECES;
    const int ii = min(i, YPIECES - 1);
    x_off[k] = (unsigned)(((16 * sw + ii / NPC) * a.xsl + (ii % NPC) * 8) * 2);
  }
  const unsigned dt_off = (unsigned)(lane * a.dsl * 2);
  // y stores: this lane's (row, piece) inside the wave's 16 rows, as running pointers
  bf16_t* ypt[NXI];
  int y_lds[NXI], y_row[NXI];
#pragma unroll
  for (int k = 0; k < NXI; ++k) {
    const int ii = min(lane + 64 * k, YPIECES - 1);
    y_row[k] = 16 * sw + ii / NPC;
    ypt[k] = yg + (int64_t)y_row[k] * a.ysl + (ii % NPC) * 8;
    y_lds[k] = (y_row[k] * YSTR + (ii % NPC) * 8) * 2;
  }
  const int64_t ystep = (int64_t)MQ * a.ysl;

  // ---- DMA issue.  Chunks whose 64 rows are all < L use the precomputed offsets; the last
  // partial chunk clamps its rows to L-1 (their dt is forced to 0, so they add nothing).
  auto issue_bc = [&](int c) {                     // state-waves: 8 DMA ops
    const int slot = c % NBC;
    const int t0 = c * MQ;
    const void* sb = uniform_ptr(Bg + (int64_t)t0 * a.bsl);
    const void* sc = uniform_ptr(Cg + (int64_t)t0 * a.csl);
    const bool full = t0 + MQ <= L;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned ob = bc_off_b[k], oc = bc_off_c[k];
      if (!full) {
        const int row = 4 * (4 * sw + k) + (lane >> 4);
        const int cg = (lane & 15) ^ (row & 15);
        const int rr = min(row, L - 1 - t0);
        ob = (unsigned)((rr * a.bsl + cg * 8) * 2);
        oc = (unsigned)((rr * a.csl + cg * 8) * 2);
      }
      glds16(sb, ob, lds_addr_of(sm.bc[slot][0] + (4 * sw + k) * 512));
      glds16(sc, oc, lds_addr_of(sm.bc[slot][1] + (4 * sw + k) * 512));
    }
  };
  auto issue_x = [&](int c) {                      // y-waves: NXI x pieces + 1 dt (or filler)
    const int slot = c % NXS;
    const int t0 = c * MQ;
    const void* sx = uniform_ptr(xg + (int64_t)t0 * a.xsl);
    const void* sd = uniform_ptr(dtg + (int64_t)t0 * a.dsl);
    const bool full = t0 + MQ <= L;
    bf16_t* xdst = sm.xr[slot] + 16 * sw * PW;
#pragma unroll
    for (int k = 0; k < NXI; ++k) {
      unsigned o = x_off[k];
      if (!full) {
        const int ii = min(lane + 64 * k, YPIECES - 1);
        const int row = min(16 * sw + ii / NPC, L - 1 - t0);
        o = (unsigned)((row * a.xsl + (ii % NPC) * 8) * 2);
      }
      if (x_act[k]) glds16(sx, o, lds_addr_of(xdst + 512 * k));   // EXEC masks the tail lanes
    }
    // raw dt of the chunk AFTER this one (prep_chunk runs a step early)
    unsigned od = dt_off + (unsigned)(MQ * a.dsl * 2);
    if (t0 + 2 * MQ > L) od = (unsigned)(min(MQ + lane, L - 1 - t0) * a.dsl * 2);
    glds4(sd, od, lds_addr_of(sw == 0 ? sm.dtr[(c + 1) % (NXS + 1)] : sm.pad_));
  };

  // discretise dt and prefix-sum dt*A for chunk c (one wave, 64 lanes = 64 tokens)
  float decay_total = 0.f;
  auto prep_chunk = [&](int c) {
    const int ab = c & 1;
    const int t = c * MQ + lane;
    float d = 0.f;
    if (t < L) {
      const unsigned w = sm.dtr[c % (NXS + 1)][lane];
      d = ((h & 1) ? bf16_hi(w) : bf16_lo(w)) + bias;
      if (a.softplus) d = softplus_fast(d);
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

  // ---- prologue: B/C of chunks 0,1 and x/dt of chunks 0..DX-1 in flight; chunk 0 prepared ----
  if (ywave) {
    if (wave == 0) glds4(uniform_ptr(dtg), (unsigned)(min(lane, L - 1) * a.dsl * 2), lds_addr_of(sm.dtr[0]));
    for (int c = 0; c < DX && c < nchunks; ++c) issue_x(c);
  } else {
    issue_bc(0);
    if (nchunks > 1) issue_bc(1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wave == 0) prep_chunk(0);
  __syncthreads();

  const unsigned char* Sb = reinterpret_cast<const unsigned char*>(sm.S);
  unsigned char* Mb = reinterpret_cast<unsigned char*>(sm.M);
  unsigned char* xs = reinterpret_cast<unsigned char*>(sm.xs);
  const unsigned char* ytb = reinterpret_cast<const unsigned char*>(sm.yt);
  for (int c = 0; c < nchunks; ++c) {
    const int ab = c & 1;
    const unsigned char* Bt = reinterpret_cast<const unsigned char*>(sm.bc[c % NBC][0]);
    const unsigned char* Ct = reinterpret_cast<const unsigned char*>(sm.bc[c % NBC][1]);
    const unsigned char* xt = reinterpret_cast<const unsigned char*>(sm.xr[c % NXS]);
    const bool noload = DBG(a, 2);
    const bool issued_bc = (c + 2 < nchunks) && !noload;
    const bool issued_x = (c + DX < nchunks) && !noload;
    if (ywave) { if (issued_x) issue_x(c + DX); }
    else if (issued_bc) issue_bc(c + 2);

    const bool skip_y = DBG(a, 1) || DBG(a, 16), skip_s = DBG(a, 1) || DBG(a, 32);
    // ================= phase 1 ============================================================
    f32x4 yoff[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) yoff[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ywave) {
      if (!skip_y) {
        // Yoff^T[p][t] = sum_n S[p][n] C[t][n],  t-tile = wave
        // all 4 + 4*PT fragment reads are issued before the first MFMA: one LDS round trip
        bf16x8 cf[4], sf[4][PT];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          cf[ks] = ld8(Ct + sw * 4096 + rfo[ks]);
#pragma unroll
          for (int j = 0; j < PT; ++j) sf[ks][j] = ld8(Sb + s_rd + j * (16 * SSTR * 2) + ks * 64);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < PT; ++j) yoff[j] = mfma16(sf[ks][j], cf[ks], yoff[j]);
        // x~ tile for the state update: rows 16w..16w+15, 16-byte pieces
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
          const int i = lane + 64 * k;
          if (i < YPIECES) {
            const int off = (16 * sw * PW) * 2 + i * 16;
            const float w = sm.wts[ab][16 * sw + i / NPC];
            const uint4 v = *(const uint4*)(xt + off);
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o[2 * e] = (bf16_t)(bf16_lo(u[e]) * w);
              o[2 * e + 1] = (bf16_t)(bf16_hi(u[e]) * w);
            }
            *(bf16x8*)(xs + off) = o;
          }
        }
      }
      // wave 0 prepares dt / cumsum of the NEXT chunk (its dt landed steps ago)
      if (wave == 0 && c + 1 < nchunks && !DBG(a, 1)) prep_chunk(c + 1);
    } else if (!skip_s) {
      // causal CB^T tiles (t-tile, s-tile), three per wave, no branches:
      //   sw0 (3,0)(3,1)|(0,0)   sw1 (3,2)(3,3)|(1,1)   sw2 (2,0)(2,1)|(1,0)   sw3 (2,2)(2,3)*|(0,1)*
      // (* = above the diagonal: the causal mask turns them into the zeros M must hold there).
      // Two tiles share the C rows; the three accumulator chains interleave.
      const int tA = sw < 2 ? 3 : 2, sA = (sw & 1) * 2;
      const int tB = (sw == 1 || sw == 2) ? 1 : 0, sB = sw & 1;
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
      bf16x8 fca[4], fcb[4], fb0[4], fb1[4], fb2[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {   // 20 fragment reads in flight before the first MFMA
        fca[ks] = ld8(Ct + tA * 4096 + rfo[ks]);
        fb0[ks] = ld8(Bt + sA * 4096 + rfo[ks]);
        fb1[ks] = ld8(Bt + (sA + 1) * 4096 + rfo[ks]);
        fb2[ks] = ld8(Bt + sB * 4096 + rfo[ks]);
        fcb[ks] = ld8(Ct + tB * 4096 + rfo[ks]);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        c0 = mfma16(fb0[ks], fca[ks], c0);
        c1 = mfma16(fb1[ks], fca[ks], c1);
        c2 = mfma16(fb2[ks], fcb[ks], c2);
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
        *(bf16x4*)(Mb + m_wr + ti * (16 * MSTR * 2) + si * 32) = o;
      };
      emit(c0, tA, sA);
      emit(c1, tA, sA + 1);
      emit(c2, tB, sB);
    }
    // ---- barrier A: M and x~ visible; every Yoff read of S is done ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ================= phase 2 ============================================================
    if (ywave) {
      if (!skip_y) {
        // Ydiag^T[p][t] = sum_s x[s][p] M[t][s]
        f32x4 yd[PT];
#pragma unroll
        for (int j = 0; j < PT; ++j) yd[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (s > t tiles of M hold zeros, so both k-steps run for every wave: no branch)
        bf16x8 mf[2], xf[2][PT];
        uint2 xv[PT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          mf[ks] = ld8(Mb + m_rd + sw * (16 * MSTR * 2) + ks * 64);
#pragma unroll
          for (int j = 0; j < PT; ++j) {
            const unsigned char* xp = xt + trx + ks * (32 * PW * 2) + j * 32;
            xf[ks][j] = cat4(tr4(xp), tr4(xp + 4 * PW * 2));
          }
        }
#pragma unroll
        for (int j = 0; j < PT; ++j)
          xv[j] = *(const uint2*)(xt + ((16 * sw + lc) * PW + 16 * j + 4 * kq) * 2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < PT; ++j) yd[j] = mfma16(xf[ks][j], mf[ks], yd[j]);
        // y^T[p = 16j + 4kq + r][t = 16w + lc]
        const int t = 16 * sw + lc;
        const float e = __expf(sm.cs[ab][t]);
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const int p0 = 16 * j + 4 * kq;
          bf16x4 o;
          o[0] = (bf16_t)(yd[j][0] + e * yoff[j][0] + Dh * bf16_lo(xv[j].x));
          o[1] = (bf16_t)(yd[j][1] + e * yoff[j][1] + Dh * bf16_hi(xv[j].x));
          o[2] = (bf16_t)(yd[j][2] + e * yoff[j][2] + Dh * bf16_lo(xv[j].y));
          o[3] = (bf16_t)(yd[j][3] + e * yoff[j][3] + Dh * bf16_hi(xv[j].y));
          *(bf16x4*)(sm.yt + t * YSTR + p0) = o;
        }
      }
    } else if (!skip_s) {
      // X[n][p] = exp(cs_Q) X[n][p] + sum_t B[t][n] x~[t][p],  n in [32sw, 32sw+32)
      const float dl = sm.dlast[ab][0];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) xacc[i][j][r] *= dl;
      bf16x8 xf[2][PT], bf[2][2];
#pragma unroll
      for (int ks = 0; ks < MQ / 32; ++ks) {   // all transposing reads first
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const unsigned char* xp = xs + trx + ks * (32 * PW * 2) + j * 32;
          xf[ks][j] = cat4(tr4(xp), tr4(xp + 4 * PW * 2));                   // k = t, cols p
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
          bf[ks][i] = cat4(tr4(Bt + trb[i][0] + ks * 8192), tr4(Bt + trb[i][1] + ks * 8192));
      }
#pragma unroll
      for (int ks = 0; ks < MQ / 32; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < PT; ++j) xacc[i][j] = mfma16(bf[ks][i], xf[ks][j], xacc[i][j]);   // rows n
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
    // ---- barrier B: the next chunk's tiles have landed; S / cs visible.
    // vm ops younger than the DMA group that must have landed now:
    //   state-waves: B/C of chunk c+1 were issued a step ago; younger = this step's 8 copies
    //   y-waves: x/dt of chunk c+1 were issued DX-1 steps ago; since then (DX-1) x (NXI+1
    //            copies + NXI y stores) were issued
    if (noload) {
    } else if (ywave) {
      if (issued_x) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YWAIT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (issued_bc) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- coalesced y store: each y-wave streams out the 16 rows it produced (whole rows,
    // 16 bytes per lane; no other wave touches them).  Exactly NXI store instructions per
    // y-wave and step while full chunks remain (vmcnt bookkeeping above). ----
    if (ywave) {
      const bool full = (c + 1) * MQ <= L;
#pragma unroll
      for (int k = 0; k < NXI; ++k) {
        if (x_act[k] && (full || c * MQ + y_row[k] < L) && !DBG(a, 4))
          *(bf16x8*)ypt[k] = *(const bf16x8*)(ytb + y_lds[k]);
        ypt[k] += ystep;
      }
    }
  }

  if (!ywave && a.final_state) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        const int p = 16 * j + lc, n = 16 * (2 * sw + i) + 4 * kq;
        if (p < PW)
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

template <int PT, int PW>
hipError_t launch_march(const MarchArgs& a, dim3 grid, hipStream_t st) {
  const size_t lds = sizeof(MarchSmem<PW>);
  static_assert(sizeof(MarchSmem<PW>) <= 160 * 1024, "LDS budget");
  hipError_t e = hipFuncSetAttribute((const void*)ssd_march_kernel<PT, PW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  ssd_march_kernel<PT, PW><<<grid, MTHREADS, lds, st>>>(a);
  return hipSuccess;
}

}  // namespace

bool tv_ssd_march_supported(int seqlen, int nheads, int headdim, int ngroups, int dstate,
                            int dtype, int64_t xsl, int64_t bsl, int64_t bsg, int64_t csl,
                            int64_t csg, int64_t ysl, const void* x, const void* Bm,
                            const void* Cm, const void* y) {
  int ns, pw;
  if (dtype != TV_BF16 || dstate != MN || seqlen < 1) return false;
  if (!pick_slices(headdim, &ns, &pw)) return false;
  if (xsl % 8 || bsl % 8 || csl % 8 || bsg % 8 || csg % 8 || ysl % 8 || nheads % 2) return false;
  if (((uintptr_t)x & 15) || ((uintptr_t)Bm & 15) || ((uintptr_t)Cm & 15) || ((uintptr_t)y & 15))
    return false;
  // per-lane DMA offsets are 32-bit byte offsets from the chunk base
  if (64 * xsl * 2 >= (1ll << 31) || 64 * bsl * 2 >= (1ll << 31) || 64 * csl * 2 >= (1ll << 31))
    return false;
  (void)ngroups;
  return true;
}

size_t tv_ssd_march_workspace_bytes(int, int, int, int, int, int) { return 0; }

int tv_ssd_march_launch(const void* x, const void* dt, const void* A, const void* Bm,
                        const void* Cm, const void* D, const void* dt_bias,
                        const void* init_state, void* y, void* final_state, void* total_decay,
                        int batch, int seqlen, int nheads, int headdim, int ngroups, int dstate,
                        int64_t xsb, int64_t xsl, int64_t dsb, int64_t dsl, int64_t bsb,
                        int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg,
                        int64_t ysb, int64_t ysl, int dtype, int dt_softplus, float dt_min,
                        float dt_max, int group_map, void* workspace, size_t workspace_bytes,
                        hipStream_t st) {
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
  hipError_t e = hipSuccess;
  switch (a.pw) {
    case 8: e = launch_march<1, 8>(a, grid, st); break;
    case 16: e = launch_march<1, 16>(a, grid, st); break;
    case 24: e = launch_march<2, 24>(a, grid, st); break;
    case 32: e = launch_march<2, 32>(a, grid, st); break;
    case 40: e = launch_march<3, 40>(a, grid, st); break;
    case 48: e = launch_march<3, 48>(a, grid, st); break;
    default: TV_UNSUPPORTED("ssd_march: slice width %d", a.pw);
  }
  if (e != hipSuccess) {
    tv_set_error("ssd_march: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return TV_ERR_LAUNCH;
  }
  TV_LAUNCH_CHECK();
}
