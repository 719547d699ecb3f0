// S3 (fast path, v2): Mamba-2 SSD selective scan as a "slice march" on CDNA4.
//
// A single pass over HBM (x, dt, B, C read once, y written once, the running state never leaves the
// chip), organised around what bounded round 1's chunk march (state published to LDS, a barrier and a
// re-read every chunk; removed in round 6, docs/history.md): LDS operand traffic and that round trip.
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
#include <type_traits>
#include "ssd_common.hpp"

// ssd_correct.hip
size_t tv_ssd_correct_workspace_bytes(int batch, int seqlen, int nheads);
int tv_ssd_correct_launch(void* y, const void* dt, const void* A, const void* Cm, const void* dt_bias,
                          const void* state_in, int batch, int seqlen, int nheads, int headdim,
                          int ngroups, int64_t ysb, int64_t ysl, int64_t dsb, int64_t dsl, int64_t csb,
                          int64_t csl, int64_t csg, int dt_softplus, float dt_min, float dt_max,
                          int group_map, void* workspace, const float* chunk_tot, int64_t chunk_tot_stride,
                          hipStream_t st);

// TV_BSCALE (wide layouts): 1 = the B/C waves scale the B tile in place (B~ = w_t B) and the slice-waves run the
// state update on raw x fragments; 0 = the slice-waves form x~ = w_t x on their own fragments (round 2)
#ifndef TV_BSCALE
#define TV_BSCALE 0
#endif
// TV_YDIRECT (wide layouts): 1 = the slice-waves store y from their registers (no y tiles in LDS: room for a B/C ring
// of 3); 0 = y tiles in LDS, stored by the B/C waves, B/C ring of 2 (round 2)
#ifndef TV_YDIRECT
#define TV_YDIRECT 0
#endif
// TV_YDIAG_FIRST (wide layouts): 1 = a step OPENS with Ydiag + D x (M and the raw x fragments are the first reads after the
// barrier; their twelve MFMAs run while the C / B fragments of quarter 0 arrive) and ends with y = e^{cs_t} Yoff + that,
// vector work only; 0 = Ydiag accumulates onto the scaled Yoff at the end of the step (round 2)
#ifndef TV_YDIAG_FIRST
#define TV_YDIAG_FIRST 0
#endif
// TV_SLICE_PRIO: static s_setprio level of the slice-waves (0 = none)
#ifndef TV_SLICE_PRIO
#define TV_SLICE_PRIO 0
#endif
namespace {
using namespace ssdk;
constexpr bool BSCALE = TV_BSCALE != 0, YDIRECT = TV_YDIRECT != 0, YDF = TV_YDIAG_FIRST != 0;
static_assert(YDIRECT || !BSCALE, "TV_BSCALE needs the ring of 3 (TV_YDIRECT)");

constexpr int SQ = 64;           // tokens per chunk
constexpr int SN = 128;          // d_state
constexpr int NV = 3;            // cs / ecs / dt / weight vector buffers
// Two layouts of a work-group.
//   narrow (slices of <= 48 columns, two work-groups per head at Nano dims; round 1): 12 waves, one
//     16-column tile per slice-wave, B/C ring of 3 (prefetch distance 2), x issued 4 chunks ahead,
//     x~ built by helper waves.
//   wide (a whole head, <= 80 columns): 8 waves with up to 256 registers each; a slice-wave owns TWO
//     16-column tiles and uses every C / B^T fragment it reads from LDS for both (the fragment reads
//     were what bounded the narrow slice-waves: 39 KB per 16 columns and step); x~ = w_t x is formed by
//     the slice-waves on their own fragments; B/C ring of 2 (distance 1: the y tiles need the LDS),
//     x issued 3 chunks ahead.
template <int PW> struct Rings {
  static constexpr bool WIDE = PW > 48;
  static constexpr int NB = (WIDE && !YDIRECT) ? 2 : 3;      // B/C ring slots
  static constexpr int BD = NB - 1;            // B/C prefetch distance (chunks)
  static constexpr int DXS = WIDE ? 3 : 4;     // x prefetch distance
  static constexpr int NXS = DXS + 1;          // x ring slots
  static constexpr int NDT = NXS + 2;          // raw-dt ring slots
};
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
  // wave 0: fragments (2,0) (2,1); wave 1: (3,0) (3,1) — one C t-tile against both s-pairs;
  // wave 2: (0,0) (1,0) — two C t-tiles against s-pair 0.  Every operand row is loaded once
  // per wave (20 / 20 / 16 loads of 16 bytes per lane).
  bf16x8 cf[2][4], b0[2][4], b1[2][4];
  const int nct = wave == 2 ? 2 : 1, nsp = wave == 2 ? 1 : 2;
  const int ti0 = wave == 0 ? 2 : wave == 1 ? 3 : 0;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (u < nct) {
      const bf16_t* cp = Cg + (int64_t)min(16 * (ti0 + u) + lc, rmax) * a.csl + 8 * kq;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) cf[u][ks] = *(const bf16x8*)(cp + 32 * ks);
    }
    if (u < nsp) {
      const int s_lo = 32 * u + 8 * (lc >> 2) + (lc & 3);
      const bf16_t* bp0 = Bg + (int64_t)min(s_lo, rmax) * a.bsl + 8 * kq;
      const bf16_t* bp1 = Bg + (int64_t)min(s_lo + 4, rmax) * a.bsl + 8 * kq;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        b0[u][ks] = *(const bf16x8*)(bp0 + 32 * ks);
        b1[u][ks] = *(const bf16x8*)(bp1 + 32 * ks);
      }
    }
  }
#pragma unroll
  for (int ff = 0; ff < 2; ++ff) {
    // fragment index in the (0,0) (1,0) (2,0) (2,1) (3,0) (3,1) order
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
  // sequence segments (blockIdx.z): segment s marches chunks [s * seg_chunks, ...) from a zero state
  // (segment 0 from `init`) and leaves its final state / total log-decay in seg_state / seg_decay;
  // nseg == 1: one march over the whole sequence, final_state / total_decay written directly
  float *seg_state, *seg_decay;
  float* chunk_tot;                // (B, H, nchunks) log2-decay of every chunk, for the correction's prefix (nseg > 1)
  int nseg, seg_chunks;
  int L, H, P, G, nslices, pw, nchunks;
  int64_t xsb, xsl, dsb, dsl, bsb, bsl, bsg, csb, csl, csg, ysb, ysl;
  int softplus, group_map;
  float dt_min, dt_max;
  int dbg;
};

// -DTV_MARCH_ABLATE builds ablation switches (env TV_MARCH_DBG) into the kernel:
// 1 slice-waves idle, 2 no B/C copies, 4 no x copies / y stores, 8 no mask build,
// 16 no x~ / prep, 32 slice-waves: no state update, 64: no Yoff, 128: no Ydiag/epilogue
#ifdef TV_MARCH_ABLATE
#define SDBG(a, bit) ((a).dbg & (bit))
#else
#define SDBG(a, bit) 0
#endif

template <int PW>
struct __attribute__((aligned(16))) SliceSmem {
  static constexpr int XSLOT = SQ * PW + 32;   // + finite guard (column tiles past the slice width read <= 48 B past the last row)
  static constexpr int NB = Rings<PW>::NB, NXS = Rings<PW>::NXS, NDT = Rings<PW>::NDT;
  bf16_t bt[NB][SQ * SN];     // B tiles [t][n], 16-byte chunk index ^ 4(t & 3) (ds_read_b64_tr)
  bf16_t ct[NB][SQ * SN];     // C tiles [t][n], chunks XOR-swizzled for row reads
  bf16_t xr[NXS][XSLOT];      // x tiles [t][PW]
  static constexpr bool WIDE = Rings<PW>::WIDE;
  bf16_t xs[WIDE ? 1 : 2][WIDE ? 8 : XSLOT];        // x~ = exp(cs_Q - cs_t) dt_t x   (narrow layout only)
  bf16_t M[2][CB_ELEMS];      // decay-masked C.B^T fragments
  bf16_t yt[(WIDE && YDIRECT) ? 1 : 2][(WIDE && YDIRECT) ? 8 : SQ * PW];      // y tiles [t][PW]   (not with TV_YDIRECT)
  unsigned dtr[NDT][SQ];      // raw dt of heads (h&~1, h|1)
  float cs[NV][SQ];           // inclusive cumsum of dt*A inside the chunk, times log2(e)
  float ecs[NV][SQ];          // exp(cs)
  float dtv[NV][SQ];          // discretised dt
  float wts[NV][SQ];          // exp(cs_last - cs_t) * dt_t
  float dl[NV][4];            // exp(cs_last)
  // off-diagonal 16x16 blocks of the decay mask are separable around the first token of
  // their t-tile: 2^(cs2_t - cs2_s) dt_s = ut[t] * ws[ti][s],  both factors <= 1 resp. dt
  float ut[2][SQ];            // 2^(cs2_t - cs2_{16 (t/16)})
  float ws[2][96];            // t-tile 1: s < 16 at [0,16); tile 2: s < 32 at [16,48); tile 3: [48,96)
};

__device__ __forceinline__ unsigned lds_lane_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) void*)p;
}
// (a ^ k) + b in one VALU op (k: wave-uniform)
__device__ __forceinline__ int xad(int a, int k, int b) {
  int d;
  asm("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(b));
  return d;
}

// wave roles (waves w, w+4, w+8 share a SIMD).
//   narrow: 0-2 slices, 3 x/dt/y, 4-6 + 11 B/C (+ one x~ piece each), 7-8 mask, 9 prep, 10 x~ pieces —
//     the two mask waves, the heaviest VALU helpers, sit on different SIMDs; SIMD 3 has no slice-wave;
//   wide:   0-2 slices (two column tiles each), 3 x/dt copies + prep, 4-5 B/C copies + y stores, 6-7 mask.
//   (round 3's 12-wave whole-head layout with one column tile per slice-wave, "impl 5", measured slower than this one and was
//   removed in round 6: docs/history.md.)
template <int PW> struct Roles {
  static constexpr bool WIDE = PW > 48;
  static constexpr int NWAVES = WIDE ? 8 : 12;
  static constexpr int NC = WIDE ? 2 : 1;          // 16-column tiles per slice-wave
  static constexpr int XIO = 3;                    // x / dt DMA (+ y stores, narrow; + prep, wide)
  static constexpr int PREP = WIDE ? XIO : 9;      // dt -> softplus -> prefix sum, mask factors
  static constexpr int SCALE = WIDE ? -1 : 10;     // x~ pieces the four B/C waves do not take (narrow)
  static constexpr int NBCW = WIDE ? 2 : 4;        // B/C copy waves
  static __device__ __forceinline__ int bc(int w) {
    if (WIDE) return (w == 4 || w == 5) ? w - 4 : -1;
    return w == 4 ? 0 : w == 5 ? 1 : w == 6 ? 2 : w == 11 ? 3 : -1;
  }
  static __device__ __forceinline__ int mask(int w) {
    if (WIDE) return w == 6 ? 0 : w == 7 ? 1 : -1;
    return w == 7 ? 0 : w == 8 ? 1 : -1;
  }
};

// -DTV_SLICE_STAMP: every wave of workgroup 0 sums the cycles it spends parked at the step
// barrier (s_memtime); tv_ssd_slice_debug_stamps() returns {wait[16], total[16]}.
#ifdef TV_SLICE_STAMP
__device__ unsigned long long g_slice_stamps[32];
#define SLICE_BARRIER()                                   \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    const unsigned long long ta__ = clock64();            \
    __builtin_amdgcn_s_barrier();                         \
    stamp_wait += clock64() - ta__;                       \
  } while (0)
#define PSTAMP(slot) do { const unsigned long long n__ = clock64(); ph_acc[slot] += n__ - ph_last; ph_last = n__; } while (0)
__device__ unsigned long long g_slice_phases[8];
#else
#define PSTAMP(slot) do {} while (0)
#define SLICE_BARRIER()                                   \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
  } while (0)
#endif

template <int PT, int PW>
__global__ __launch_bounds__((Roles<PW>::NWAVES * 64)) void ssd_slice_kernel(SliceArgs a) {
  typedef SliceSmem<PW> Smem;
  typedef Rings<PW> RG;
  typedef Roles<PW> RL;
  constexpr int STHREADS = RL::NWAVES * 64, NC = RL::NC;
  constexpr int NB = RG::NB, BD = RG::BD, DXS = RG::DXS, NXS = RG::NXS, NDT = RG::NDT;
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
  // this work-group's segment of the sequence: chunks [c_first, c_first + nchunks), tokens [t_first, t_first + L)
  const int seg = blockIdx.z;
  const int c_first = seg * a.seg_chunks;
  const int t_first = c_first * SQ;
  const int nchunks = min(a.seg_chunks, a.nchunks - c_first);
  const int L = min(a.L - t_first, nchunks * SQ);

#ifdef TV_SLICE_STAMP
  unsigned long long stamp_wait = 0;
  const unsigned long long stamp_t0 = clock64();
  unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = stamp_t0;
#endif
  {   // zero LDS once: guards / pad columns must hold finite values
    bf16x8 z = {};
    for (int i = tid; i < (int)(sizeof(Smem) / 16); i += STHREADS)
      reinterpret_cast<bf16x8*>(smem_raw)[i] = z;
  }
  __syncthreads();

  // The prologue fills the pipeline with three barriers (P1: first tiles landed, P2: chunks
  // 0/1 prepared, P3: x~_0 and M_0 built); then every role runs nchunks steps with one
  // barrier each.  At step c the slice-waves consume chunk c while the helpers produce
  // x~_{c+1}, M_{c+1}, the vectors of chunk c+2, issue B/C of chunk c+2, x of chunk c+4,
  // dt of chunk c+5 and store y_{c-2} (the slice-waves write y_{c-1} to LDS at the start of step c).
  // x~ = w_t x for one 1 KiB piece (64 lanes x 16 B) of chunk c: done by the B/C waves (one
  // piece each, they have VALU and LDS slots to spare) and by W_SCALE
  auto scale_piece = [&](int c, int k) {
    const int i = lane + 64 * k;
    const float w = sm.wts[c % NV][i / NPC];
    const uint4 v = *(const uint4*)(reinterpret_cast<const unsigned char*>(sm.xr[c % NXS]) + i * 16);
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[2 * e] = (bf16_t)(bf16_lo(u[e]) * w);
      o[2 * e + 1] = (bf16_t)(bf16_hi(u[e]) * w);
    }
    *(bf16x8*)(reinterpret_cast<unsigned char*>(sm.xs[c & 1]) + i * 16) = o;
  };
  // dt -> discretised dt, chunk-local cumulative log-decay and everything derived from it (one wave,
  // lane = token): run by the prep wave, or by the x/dt/y wave of a whole-head work-group
  const float Ah = a.A[h];
  const float bias = a.dt_bias ? a.dt_bias[h] : 0.f;
  float decay_total = 0.f;
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
    const float cs2 = cs * 1.4426950408889634f, cl2 = cl * 1.4426950408889634f;
    sm.cs[vb][lane] = cs2;                                    // log2 domain (v_exp_f32 is 2^x)
    sm.ecs[vb][lane] = __builtin_amdgcn_exp2f(cs2);
    sm.dtv[vb][lane] = d;
    sm.wts[vb][lane] = __builtin_amdgcn_exp2f(cl2 - cs2) * d;
    if (lane == 0) sm.dl[vb][0] = __builtin_amdgcn_exp2f(cl2);
    if (a.chunk_tot && slice == 0 && lane == 0) a.chunk_tot[((int64_t)b * a.H + h) * a.nchunks + c_first + c] = cl2;
    // separable factors of the off-diagonal mask blocks (pivot = first token of a t-tile)
    const float p1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs2), 16));
    const float p2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs2), 32));
    const float p3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs2), 48));
    const float p0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs2), 0));
    const float pv = lane < 16 ? p0 : lane < 32 ? p1 : lane < 48 ? p2 : p3;
    sm.ut[c & 1][lane] = __builtin_amdgcn_exp2f(fminf(cs2 - pv, 0.f));
    if (lane < 16) sm.ws[c & 1][lane] = __builtin_amdgcn_exp2f(fminf(p1 - cs2, 0.f)) * d;
    if (lane < 32) sm.ws[c & 1][16 + lane] = __builtin_amdgcn_exp2f(fminf(p2 - cs2, 0.f)) * d;
    if (lane < 48) sm.ws[c & 1][48 + lane] = __builtin_amdgcn_exp2f(fminf(p3 - cs2, 0.f)) * d;
    decay_total += cl;
  };
  if (wave < PT) {
    // ============================================================ slice-wave (NC tiles of 16 columns)
    if (TV_SLICE_PRIO) __builtin_amdgcn_s_setprio(TV_SLICE_PRIO);
    const float Dh = a.D ? a.D[h] : 0.f;
    f32x4 xacc[NC][8];
    int pcol[NC];
    bool pvalid[NC];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      pcol[ct] = 16 * (wave * NC + ct) + lc;
      pvalid[ct] = pcol[ct] < PW;
#pragma unroll
      for (int i = 0; i < 8; ++i) xacc[ct][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.init && seg == 0 && pvalid[ct]) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          xacc[ct][i] = *(const f32x4*)(a.init + (((int64_t)b * a.H + h) * a.P + p_base + pcol[ct]) * SN + 32 * (i >> 1) + 8 * kq + 4 * (i & 1));
      }
    }
    // State tiles 2m / 2m+1 hold the state rows n = 32m + 8kq + r / + 4 + r (r = accumulator
    // register), so the pair is, as a B operand, the k slots n = 32m + 8kq + 0..7 — the order
    // of a plain 16-byte row read of C.  16-byte chunk (4m + kq) ^ (t & 15) = 4m ^ c_z.
    const int c_lo = lc * 256;
    const int c_z = (kq ^ lc) << 4;
    // transposing reads of B for tiles 2m, 2m+1: row t = 32ks + 8kq + q4 (+4), 8 bytes at
    // chunk (4m + p4) ^ 4(t & 3), half = tile parity  ->  ((m ^ q4) << 6) + 16 p4 + 8 odd
    const int bsw = q4 << 6;
    const int b_lo = (8 * kq + q4) * 256 + p4 * 16;
    // transposing reads of x / x~ (B operand: k = token) and of x in accumulator layout; column tile
    // ct of this wave adds 32 ct bytes.  (A tile past the slice width reads finite neighbouring data —
    // the x slots carry a guard — and its results are never written.)
    const int trx = ((8 * kq + q4) * PW + 4 * p4) * 2 + wave * NC * 32;
    const int trd = (lc * PW + 4 * kq) * 2 + wave * NC * 32;     // x in the (transposed) y accumulator layout
    // The y accumulators are kept TRANSPOSED (the MFMA operands of Yoff and Ydiag swapped: same registers,
    // same products, same k order): a lane then holds FOUR CONSECUTIVE COLUMNS 16 tile + 4 kq + r of ONE token
    // 16 ti + lc — 8 contiguous bytes of the [t][PW] y tile, one ds_write_b64 per (tile, ti).  (With tokens on the
    // registers it took four ds_write_b16 each: 32 LDS writes per step, 680 of a slice-wave's 3 950 cycles.)
    typedef __attribute__((ext_vector_type(2))) unsigned ypk_t;
    ypk_t ypk[NC][4] = {};
    bool cvalid[NC];                // this lane's four columns exist (PW is a multiple of 8)
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) cvalid[ct] = 16 * (wave * NC + ct) + 4 * kq < PW;
    auto write_y = [&](int c) {     // y tile of chunk c (stored by the x/dt/y wave, narrow, or the B/C waves, wide)
      if (RL::WIDE && YDIRECT) return;
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) {
        if (!cvalid[ct]) continue;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
          const unsigned ya = lds_lane_addr(sm.yt[c & 1] + (16 * ti + lc) * PW + 16 * (wave * NC + ct) + 4 * kq);
          asm volatile("ds_write_b64 %0, %1" :: "v"(ya), "v"(ypk[ct][ti]) : "memory");
        }
      }
    };
    // Wide layout: y leaves the wave straight from these registers — 8 bytes per lane and (tile, t-tile),
    // NC * 4 global_store_dwordx2 per step, nobody waits for them (round 3; the y tiles in LDS, their
    // ds_write_b64 / ds_read_b128 round trip and the helper waves' stores are gone, and the 20 KiB they
    // took hold the third B/C ring slot).  The 32-byte pieces of a row that the five tiles write meet in L2.
    bf16_t* const ygs = a.y + (int64_t)b * a.ysb + (int64_t)t_first * a.ysl + (int64_t)h * a.P + p_base;
    unsigned yoff[4];               // byte offset of this lane's 8 bytes inside a chunk's rows, tile 0 of the wave
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) yoff[ti] = (unsigned)(((16 * ti + lc) * a.ysl + 16 * wave * NC + 4 * kq) * 2);
    auto store_y = [&](int c) {
      const int t0 = c * SQ;
      const void* yc = uniform_ptr(ygs + (int64_t)t0 * a.ysl);
      const bool full = t0 + SQ <= L;
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) {
        if (!cvalid[ct]) continue;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
          if (full || t0 + 16 * ti + lc < L)      // scalar base + 32-bit lane offset (a generic pointer store would be a flat_store, which also counts in lgkmcnt)
            asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" :: "v"(yoff[ti]), "v"(ypk[ct][ti]), "s"(yc), "n"(32 * ct) : "memory");
      }
    };
    bf16x8 cq[2][4];
    bf16x4 bq[2][2][4];
    auto read_cq_at = [&](const unsigned char* Ct, int q, bf16x8 (&cf)[4]) {     // [t-tile]
      const unsigned char* cp = Ct + xad(c_z, 64 * q, c_lo);
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) cf[ti] = ld8(cp + ti * 4096);
    };
    auto read_b2_at = [&](const unsigned char* Bt, int i0, bf16x4 (&dst)[2][4]) {   // tiles i0 = 2m, i0 + 1
      const unsigned char* bp = Bt + xad(bsw, 32 * i0, b_lo);
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          dst[ii][2 * ks] = tr4(bp + ii * 8 + ks * 8192);
          dst[ii][2 * ks + 1] = tr4(bp + ii * 8 + ks * 8192 + 1024);
        }
    };
    SLICE_BARRIER();   // P1
    SLICE_BARRIER();   // P2
    SLICE_BARRIER();   // P3
    for (int c = 0; c < nchunks; ++c) {
      const int vb = c % NV;
      const unsigned char* Bt = reinterpret_cast<const unsigned char*>(sm.bt[c % NB]);
      const unsigned char* Ct = reinterpret_cast<const unsigned char*>(sm.ct[c % NB]);
      const unsigned char* xt = reinterpret_cast<const unsigned char*>(sm.xr[c % NXS]);
      const unsigned char* xw = reinterpret_cast<const unsigned char*>(sm.xs[RL::WIDE ? 0 : (c & 1)]);
      const unsigned char* Mf = reinterpret_cast<const unsigned char*>(sm.M[c & 1]);
      if (SDBG(a, 1)) { SLICE_BARRIER(); continue; }
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      typedef __attribute__((ext_vector_type(2))) float f32x2;
      // Software pipeline in quarters.  Quarter q takes the 32 state rows n in [32q, 32q+32)
      // (accumulator tiles 2q, 2q+1) of every column tile: it snapshots them as bf16 B operands,
      // adds their share C[:, block q] . X[block q] to the Yoff tiles, then advances them with
      // B^T x~.  The C columns and B tiles of quarter q+1 are read while quarter q computes
      // (LDS returns in order, waits are counted) and serve all column tiles of the wave; the
      // scheduling fences keep hipcc from sinking those reads back next to their uses.
      auto read_cq = [&](int q, bf16x8 (&cf)[4]) { read_cq_at(Ct, q, cf); };
      auto read_b2 = [&](int i0, bf16x4 (&dst)[2][4]) { read_b2_at(Bt, i0, dst); };
      f32x4 yo[NC][4];
#pragma unroll
      for (int ct = 0; ct < NC; ++ct)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) yo[ct][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto quarter = [&](int q, const bf16x8 (&cf)[4], const bf16x4 (&bt2)[2][4],
                         const bf16x8 (&xwf)[NC][2], float dl) {
        bf16x8 sbq[NC];
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sbq[ct][r] = (bf16_t)xacc[ct][2 * q][r];
            sbq[ct][4 + r] = (bf16_t)xacc[ct][2 * q + 1][r];
          }
        if (!SDBG(a, 64))
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
          for (int ct = 0; ct < NC; ++ct) yo[ct][ti] = mfma16(sbq[ct], cf[ti], yo[ct][ti]);   // Yoff^T: [column][token]
        if (SDBG(a, 32)) return;
        const f32x2 dl2 = {dl, dl};
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {   // v_pk_mul_f32: two state values per instruction
            f32x2 a0 = {xacc[ct][2 * q + ii][0], xacc[ct][2 * q + ii][1]}, a1 = {xacc[ct][2 * q + ii][2], xacc[ct][2 * q + ii][3]};
            a0 *= dl2;
            a1 *= dl2;
            xacc[ct][2 * q + ii] = f32x4{a0[0], a0[1], a1[0], a1[1]};
          }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const bf16x8 bfrag = cat4(bt2[ii][2 * ks], bt2[ii][2 * ks + 1]);
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) xacc[ct][2 * q + ii] = mfma16(bfrag, xwf[ct][ks], xacc[ct][2 * q + ii]);
          }
      };
      PSTAMP(7);
      // ---- first reads
      constexpr bool YF = YDF && RL::WIDE;
      bf16x8 mf[NFRAG];
      bf16x4 xv[NC][4];               // x[t][16 tile + 4 kq + 0..3]: a plain 8-byte row read
      if (YF) {
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) mf[f] = ld8(Mf + f * 1024 + lane * 16);
      } else {
        read_cq(0, cq[0]);
      }
      bf16x4 xq[NC][2][2], xwq[NC][2][2];
      f32x4 wq[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (RL::WIDE) {      // raw x fragments (kept for Ydiag); BSCALE: the state update runs on them against B~ = w_t B
#pragma unroll
          for (int ct = 0; ct < NC; ++ct) {
            xq[ct][ks][0] = tr4(xt + trx + 32 * ct + ks * (32 * PW * 2));
            xq[ct][ks][1] = tr4(xt + trx + 32 * ct + ks * (32 * PW * 2) + 4 * PW * 2);
          }
          if (!BSCALE) {     // the weights of their eight tokens
            wq[ks][0] = *(const f32x4*)(&sm.wts[vb][32 * ks + 8 * kq]);
            wq[ks][1] = *(const f32x4*)(&sm.wts[vb][32 * ks + 8 * kq + 4]);
          }
        } else {
#pragma unroll
          for (int ct = 0; ct < NC; ++ct) {
            xwq[ct][ks][0] = tr4(xw + trx + 32 * ct + ks * (32 * PW * 2));
            xwq[ct][ks][1] = tr4(xw + trx + 32 * ct + ks * (32 * PW * 2) + 4 * PW * 2);
          }
        }
      }
      const float dl = sm.dl[vb][0];
      if (YF) {
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
          for (int ct = 0; ct < NC; ++ct) xv[ct][ti] = *(const bf16x4*)(xt + trd + 32 * ct + ti * (16 * PW * 2));
        read_cq(0, cq[0]);
      }
      read_b2(0, bq[0]);
      PSTAMP(6);
      if (c > 0) write_y(c - 1);        // previous chunk's results, under the reads just issued
      __builtin_amdgcn_sched_barrier(0);
      f32x4 yd[NC][4];                  // Ydiag + D x (YF)
      if (YF && !SDBG(a, 128)) {
        const f32x2 dh2 = {Dh, Dh};
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
          const bf16x8 xf[2] = {cat4(xq[ct][0][0], xq[ct][0][1]), cat4(xq[ct][1][0], xq[ct][1][1])};
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) {
            const unsigned x01 = __builtin_bit_cast(u32x2, xv[ct][ti])[0], x23 = __builtin_bit_cast(u32x2, xv[ct][ti])[1];
            const f32x2 d0 = dh2 * f32x2{bf16_lo(x01), bf16_hi(x01)}, d1 = dh2 * f32x2{bf16_lo(x23), bf16_hi(x23)};
            yd[ct][ti] = f32x4{d0[0], d0[1], d1[0], d1[1]};
          }
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) {
            const int f0 = ti == 0 ? 0 : ti == 1 ? 1 : ti == 2 ? 2 : 4;
            yd[ct][ti] = mfma16(xf[0], mf[f0], yd[ct][ti]);
            if (ti >= 2) yd[ct][ti] = mfma16(xf[1], mf[f0 + 1], yd[ct][ti]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      PSTAMP(0);
      read_cq(1, cq[1]);
      read_b2(2, bq[1]);
      bf16x8 xwf[NC][2];
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) {
        if (RL::WIDE && BSCALE) {
          xwf[ct][0] = cat4(xq[ct][0][0], xq[ct][0][1]);
          xwf[ct][1] = cat4(xq[ct][1][0], xq[ct][1][1]);
        } else if (RL::WIDE) {
          // x~ = w_t x on this wave's own fragments: element j of fragment ks is token 32 ks + 8 kq + j
          // (same products and roundings as the helper waves' scale_piece of the narrow layout)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              const u32x2 u = __builtin_bit_cast(u32x2, xq[ct][ks][hf]);
              xwf[ct][ks][4 * hf + 0] = (bf16_t)(bf16_lo(u[0]) * wq[ks][hf][0]);
              xwf[ct][ks][4 * hf + 1] = (bf16_t)(bf16_hi(u[0]) * wq[ks][hf][1]);
              xwf[ct][ks][4 * hf + 2] = (bf16_t)(bf16_lo(u[1]) * wq[ks][hf][2]);
              xwf[ct][ks][4 * hf + 3] = (bf16_t)(bf16_hi(u[1]) * wq[ks][hf][3]);
            }
        } else {
          xwf[ct][0] = cat4(xwq[ct][0][0], xwq[ct][0][1]);
          xwf[ct][1] = cat4(xwq[ct][1][0], xwq[ct][1][1]);
        }
      }
      quarter(0, cq[0], bq[0], xwf, dl);
      __builtin_amdgcn_sched_barrier(0);
      PSTAMP(1);
      read_cq(2, cq[0]);
      read_b2(4, bq[0]);
      quarter(1, cq[1], bq[1], xwf, dl);
      __builtin_amdgcn_sched_barrier(0);
      PSTAMP(2);
      read_cq(3, cq[1]);
      read_b2(6, bq[1]);
      quarter(2, cq[0], bq[0], xwf, dl);
      __builtin_amdgcn_sched_barrier(0);
      PSTAMP(3);
      // epilogue operands in flight under the last quarter
      if (!YF) {
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) mf[f] = ld8(Mf + f * 1024 + lane * 16);
      }
      float ev[4];                    // exp(cs_t) of this lane's token 16 ti + lc
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        ev[ti] = sm.ecs[vb][16 * ti + lc];
        if (!YF) {
#pragma unroll
          for (int ct = 0; ct < NC; ++ct) xv[ct][ti] = *(const bf16x4*)(xt + trd + 32 * ct + ti * (16 * PW * 2));
        }
      }
      if (!RL::WIDE) {       // (wide: the raw fragments were read at the top of the step)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            xq[ct][ks][0] = tr4(xt + trx + 32 * ct + ks * (32 * PW * 2));
            xq[ct][ks][1] = tr4(xt + trx + 32 * ct + ks * (32 * PW * 2) + 4 * PW * 2);
          }
      }
      quarter(3, cq[1], bq[1], xwf, dl);
      __builtin_amdgcn_sched_barrier(0);
      PSTAMP(4);
      // ---- y = exp(cs_t) Yoff + M x + D x  (Ydiag accumulates onto the scaled Yoff)
      if (!SDBG(a, 128)) {
        const f32x2 dh2 = {Dh, Dh};
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
          if (YF) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
              f32x2 y0 = {yo[ct][ti][0], yo[ct][ti][1]}, y1 = {yo[ct][ti][2], yo[ct][ti][3]};
              y0 = __builtin_elementwise_fma(y0, f32x2{ev[ti], ev[ti]}, f32x2{yd[ct][ti][0], yd[ct][ti][1]});
              y1 = __builtin_elementwise_fma(y1, f32x2{ev[ti], ev[ti]}, f32x2{yd[ct][ti][2], yd[ct][ti][3]});
              yo[ct][ti] = f32x4{y0[0], y0[1], y1[0], y1[1]};
            }
          } else {
          const bf16x8 xf[2] = {cat4(xq[ct][0][0], xq[ct][0][1]), cat4(xq[ct][1][0], xq[ct][1][1])};
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) {
            const unsigned x01 = __builtin_bit_cast(u32x2, xv[ct][ti])[0], x23 = __builtin_bit_cast(u32x2, xv[ct][ti])[1];
            f32x2 y0 = {yo[ct][ti][0], yo[ct][ti][1]}, y1 = {yo[ct][ti][2], yo[ct][ti][3]};
            y0 = __builtin_elementwise_fma(y0, f32x2{ev[ti], ev[ti]}, dh2 * f32x2{bf16_lo(x01), bf16_hi(x01)});
            y1 = __builtin_elementwise_fma(y1, f32x2{ev[ti], ev[ti]}, dh2 * f32x2{bf16_lo(x23), bf16_hi(x23)});
            yo[ct][ti] = f32x4{y0[0], y0[1], y1[0], y1[1]};
          }
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) {
            const int f0 = ti == 0 ? 0 : ti == 1 ? 1 : ti == 2 ? 2 : 4;
            yo[ct][ti] = mfma16(xf[0], mf[f0], yo[ct][ti]);
            if (ti >= 2) yo[ct][ti] = mfma16(xf[1], mf[f0 + 1], yo[ct][ti]);
          }
          }
          // packed bf16 results stay in registers across the barrier; they are written to the y
          // tile at the start of the next step, beside that step's first fragment reads
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            const bf16x2 p01 = {(bf16_t)yo[ct][ti][0], (bf16_t)yo[ct][ti][1]}, p23 = {(bf16_t)yo[ct][ti][2], (bf16_t)yo[ct][ti][3]};
            ypk[ct][ti] = ypk_t{__builtin_bit_cast(unsigned, p01), __builtin_bit_cast(unsigned, p23)};
          }
        }
        if (RL::WIDE && YDIRECT) store_y(c);
      }
      PSTAMP(5);
      SLICE_BARRIER();
    }
#ifdef TV_SLICE_STAMP
    if (blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && lane == 0)
      for (int i = 0; i < 8; ++i) g_slice_phases[i] = ph_acc[i];
#endif
    write_y(nchunks - 1);
    float* fin = a.nseg > 1 ? a.seg_state + (int64_t)seg * gridDim.y * a.H * a.P * SN : a.final_state;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
      if (fin && pvalid[ct]) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          *(f32x4*)(fin + (((int64_t)b * a.H + h) * a.P + p_base + pcol[ct]) * SN + 32 * (i >> 1) + 8 * kq + 4 * (i & 1)) = xacc[ct][i];
      }
    SLICE_BARRIER();   // final (y of the last chunk is stored after it)
  } else if (wave == RL::XIO) {
    // ============================================================ x / dt DMA, y stores
    const bf16_t* xg = a.x + (int64_t)b * a.xsb + (int64_t)t_first * a.xsl + (int64_t)h * a.P + p_base;
    const bf16_t* dtg = a.dt + (int64_t)b * a.dsb + (int64_t)t_first * a.dsl + (h & ~1);
    bf16_t* yg = a.y + (int64_t)b * a.ysb + (int64_t)t_first * a.ysl + (int64_t)h * a.P + p_base;
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
    constexpr int DDT = RL::WIDE ? DXS + 2 : DXS + 1;      // dt copy distance (chunks)
    for (int c = 0; c < DXS; ++c) issue_x(min(c, nchunks - 1));
    for (int c = 0; c < DDT; ++c) issue_dt(c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SLICE_BARRIER();   // P1
    if (RL::WIDE) {    // wide: this wave prepares the vectors too (no y stores here)
      prep(0);
      if (nchunks > 1) prep(1);
    }
    SLICE_BARRIER();   // P2
    SLICE_BARRIER();   // P3
    // Narrow: x of chunk c+2 and dt of chunk c+3 must have landed at the end of step c (the helpers
    // build x~_{c+2} at step c+1); they were issued two steps ago, and since then this wave issued
    // 2 x (NPI stores + NPI + 1 copies) — two steps of flight time.  Wide (no x~ helpers, x issued 3
    // chunks ahead): x of chunk c+1 must have landed at the end of step c (the slice-waves read it at
    // step c+1) and dt of chunk c+3 (prep runs two chunks ahead); both were issued two steps ago (x 3
    // and dt 5 chunks ahead), the 2 (NPI + 1) copies issued since then may still be in flight.
    constexpr int FLY = RL::WIDE ? (DXS - 1) * (NPI + 1) : (DXS - 2) * (2 * NPI + 1);
    for (int c = 0; c < nchunks; ++c) {
      if (!RL::WIDE && c > 1 && !SDBG(a, 4)) store_y(c - 2);   // written by the slice-waves at the start of step c-1
      const bool issued = c + DXS < nchunks && !SDBG(a, 4);
      if (issued) {
        issue_x(c + DXS);
        issue_dt(c + DDT);
      }
      if (RL::WIDE && c + 2 < nchunks && !SDBG(a, 16)) prep(c + 2);     // under the copies just issued
      if constexpr (FLY > 0) {
        if (issued && c > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FLY) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      SLICE_BARRIER();
    }
    if (!RL::WIDE && nchunks > 1) store_y(nchunks - 2);
    SLICE_BARRIER();   // final: the slice-waves have flushed the last chunk's y tile
    if (!RL::WIDE) store_y(nchunks - 1);
    if (RL::WIDE) {
      float* td = a.nseg > 1 ? a.seg_decay + (int64_t)seg * gridDim.y * a.H : a.total_decay;
      if (td && slice == 0 && lane == 0) td[(int64_t)b * a.H + h] = decay_total;
    }
  } else if (RL::bc(wave) >= 0) {
    // ============================================================ B / C DMA
    const int q = RL::bc(wave);
    const bf16_t* Bg = a.Bm + (int64_t)b * a.bsb + (int64_t)g * a.bsg + (int64_t)t_first * a.bsl;
    const bf16_t* Cg = a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg + (int64_t)t_first * a.csl;
    constexpr int KP = 16 / RL::NBCW;        // 1 KiB pieces (4 token rows) of B per wave and chunk; as many of C
    int brow[KP];
    unsigned off_b[KP], off_c[KP];
    int cg_b[KP], cg_c[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int row = 4 * KP * q + 4 * k + (lane >> 4);
      brow[k] = row;
      cg_b[k] = (lane & 15) ^ (4 * (row & 3));
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
      if (full && KP % 4 == 0) {     // four pieces per M0 set-up (piece k's row is >= 4 k: the offsets stay >= 0)
#pragma unroll
        for (int k = 0; k + 3 < KP; k += 4) {
          glds16x4(sb, off_b[k], off_b[k + 1] - 1024u, off_b[k + 2] - 2048u, off_b[k + 3] - 3072u,
                   lds_addr_of(sm.bt[slot] + (KP * q + k) * 512));
          glds16x4(sc, off_c[k], off_c[k + 1] - 1024u, off_c[k + 2] - 2048u, off_c[k + 3] - 3072u,
                   lds_addr_of(sm.ct[slot] + (KP * q + k) * 512));
        }
        return;
      }
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        unsigned ob = off_b[k], oc = off_c[k];
        if (!full) {
          const int rr = min(brow[k], L - 1 - t0);
          ob = (unsigned)((rr * a.bsl + cg_b[k] * 8) * 2);
          oc = (unsigned)((rr * a.csl + cg_c[k] * 8) * 2);
        }
        glds16(sb, ob, lds_addr_of(sm.bt[slot] + (KP * q + k) * 512));
        glds16(sc, oc, lds_addr_of(sm.ct[slot] + (KP * q + k) * 512));
      }
    };
    // Wide layout: B~[t][n] = w_t B[t][n] (w_t = exp(cs_last - cs_t) dt_t) IN PLACE, each B/C wave on the pieces it
    // copied itself — the state update X = e^{cs_Q} X + B~^T x then runs on the raw x fragments, and the slice-waves
    // (the pole of the step) lose the unpack / multiply / repack of x~ = w_t x: ~110 of their ~400 vector instructions
    // per step.  The raw B tile has no other reader (C.B^T comes from the pre-pass).  Rows past the sequence end have
    // w = 0.
    auto scale_b = [&](int c) {
      if (!RL::WIDE || !BSCALE) return;
      unsigned char* bp = reinterpret_cast<unsigned char*>(sm.bt[c % NB]) + KP * q * 1024 + lane * 16;
      const float* wv = sm.wts[c % NV] + 4 * KP * q + (lane >> 4);
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      typedef __attribute__((ext_vector_type(2))) float f32x2;
      u32x4 v[KP];
      float w[KP];
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        v[k] = *(const u32x4*)(bp + k * 1024);
        w[k] = wv[4 * k];
      }
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x2 pr = f32x2{bf16_lo(v[k][e]), bf16_hi(v[k][e])} * f32x2{w[k], w[k]};
          o[2 * e] = (bf16_t)pr[0];
          o[2 * e + 1] = (bf16_t)pr[1];
        }
        *(bf16x8*)(bp + k * 1024) = o;
      }
    };
    // x~ pieces (narrow layout only): this wave takes one of the first four, the W_SCALE wave the rest
    auto scale_mine = [&](int c) {
      if (!RL::WIDE && q < NPI) scale_piece(c, q);
    };
    // wide without TV_YDIRECT: the two B/C waves store the y tiles (every other 1 KiB piece each).  (Measured: moved to
    // the lighter mask wave the stores' completion sits in front of that wave's compiler-placed waits
    // and makes it the pole of the step: 4 380 against 3 930 ticks.)
    bf16_t* ygw = a.y + (int64_t)b * a.ysb + (int64_t)t_first * a.ysl + (int64_t)h * a.P + p_base;
    auto store_y_half = [&](int c) {
      if (!RL::WIDE || YDIRECT) return;
      const int t0 = c * SQ;
      const unsigned char* ytb = reinterpret_cast<const unsigned char*>(sm.yt[c & 1]);
      bf16_t* yc = ygw + (int64_t)t0 * a.ysl;
#pragma unroll
      for (int k = 0; k < (NPI + 1) / 2; ++k) {
        const int i = lane + 64 * (2 * k + q);
        const int row = i / NPC;
        if (2 * k + q < NPI && t0 + row < L)
          *(bf16x8*)(yc + (int64_t)row * a.ysl + (i % NPC) * 8) = *(const bf16x8*)(ytb + i * 16);
      }
    };
    issue_bc(0);
    if (BD > 1 && nchunks > 1) issue_bc(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SLICE_BARRIER();   // P1
    SLICE_BARRIER();   // P2 (chunks 0 / 1 prepared)
    scale_mine(0);
    scale_b(0);
    SLICE_BARRIER();   // P3
    for (int c = 0; c < nchunks; ++c) {
      const bool issued = c + BD < nchunks && !SDBG(a, 2);
      if (issued) issue_bc(c + BD);
      if (RL::WIDE && BSCALE) {
        // chunk c+1 (issued a step ago) must have landed: it is scaled now and read after this step's barrier;
        // this step's copies stay in flight
        if (issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * KP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (c + 1 < nchunks && !SDBG(a, 16)) scale_b(c + 1);
      } else {
        if (c > 1 && !SDBG(a, 4)) store_y_half(c - 2);   // written by the slice-waves at the start of step c-1
        if (c + 1 < nchunks && !SDBG(a, 16)) scale_mine(c + 1);
        // chunk c+1 must have landed; chunk c+2 (this step's 2 KP copies) stays in flight
        if (BD > 1 && issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * KP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      SLICE_BARRIER();
    }
    if (nchunks > 1) store_y_half(nchunks - 2);
    SLICE_BARRIER();   // final: the slice-waves have flushed the last chunk's y tile
    store_y_half(nchunks - 1);
  } else if (RL::mask(wave) >= 0) {
    // ============================================================ decay mask M = CB .* L
    // M[t][s] = CB[t][s] 2^(cs2_t - cs2_s) dt_s for s <= t (cs2 = cs log2 e), else 0, built
    // in five wave-wide units of 16x16 blocks so that every lane of an instruction does the
    // same kind of work:
    //   unit 0 / 1: diagonal blocks (0,0)+(1,1) / (2,2)+(3,3): one v_exp per element
    //     (exponent <= 0 where s <= t, -inf -> factor 0 above the diagonal);
    //   units 2, 3, 4: off-diagonal blocks {(2,0),(2,1)}, {(3,0),(3,1)}, {(1,0),(3,2)}:
    //     ut[t] * ws[ti][s];
    //   the fragment halves above the diagonal are never written (LDS was zeroed).
    // Mask wave 0 takes units 0, 2, 3; mask wave 1 units 1, 4.
    auto run_mask = [&](auto MI) {
      constexpr int mi = decltype(MI)::value;
      constexpr int NS = mi == 0 ? 2 : 1;           // separable units of this wave
      const bf16_t* cbg = a.cb + (((int64_t)b * a.G + g) * a.nchunks + c_first) * CB_ELEMS;
      typedef __attribute__((ext_vector_type(2))) float f32x2;
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      const int hi = kq >> 1;
      // element offset of this lane's 8 values inside the chunk's CB / M image, its t, first s
      const int d_off = mi == 0 ? hi * 512 + lane * 8 : (3 + 2 * hi) * 512 + lane * 8;
      const int d_t = 32 * mi + 16 * hi + lc, d_s0 = 32 * mi + 8 * kq;
      int s_off[NS], s_t[NS], s_w[NS];
      if (mi == 0) {
        s_off[0] = 2 * 512 + lane * 8; s_t[0] = 32 + lc; s_w[0] = 16 + 8 * kq;            // frag (2,0)
        s_off[NS - 1] = 4 * 512 + lane * 8; s_t[NS - 1] = 48 + lc; s_w[NS - 1] = 48 + 8 * kq;   // frag (3,0)
      } else {
        s_off[0] = hi ? 5 * 512 + (lane - 32) * 8 : 512 + lane * 8;                        // (1,0) | (3,2)
        s_t[0] = hi ? 48 + lc : 16 + lc;
        s_w[0] = hi ? 48 + 16 + 8 * kq : 8 * kq;
      }
      auto load_cb = [&](int c, bf16x8 (&cbv)[1 + NS]) {
        cbv[0] = *(const bf16x8*)(cbg + (int64_t)c * CB_ELEMS + d_off);
#pragma unroll
        for (int u = 0; u < NS; ++u) cbv[1 + u] = *(const bf16x8*)(cbg + (int64_t)c * CB_ELEMS + s_off[u]);
      };
      auto build = [&](int c, const bf16x8 (&cbv)[1 + NS]) {
        const int vb = c % NV, pb = c & 1;
        bf16_t* Mo = sm.M[pb];
        // all LDS reads first
        float utv[NS];
        f32x4 cs_s[2], dt_s[2], wv[NS][2];
        const float cst = sm.cs[vb][d_t];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          cs_s[hh] = *(const f32x4*)(&sm.cs[vb][d_s0 + 4 * hh]);
          dt_s[hh] = *(const f32x4*)(&sm.dtv[vb][d_s0 + 4 * hh]);
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          utv[u] = sm.ut[pb][s_t[u]];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) wv[u][hh] = *(const f32x4*)(&sm.ws[pb][s_w[u] + 4 * hh]);
        }
        {   // diagonal unit
          const u32x4 cw = __builtin_bit_cast(u32x4, cbv[0]);
          bf16x8 o;
#pragma unroll
          for (int jp = 0; jp < 4; ++jp) {
            const int j0 = 2 * jp, j1 = 2 * jp + 1;
            const float a0 = (d_s0 + j0 <= d_t) ? cst - cs_s[j0 >> 2][j0 & 3] : -__builtin_inff();
            const float a1 = (d_s0 + j1 <= d_t) ? cst - cs_s[j1 >> 2][j1 & 3] : -__builtin_inff();
            const f32x2 e = {__builtin_amdgcn_exp2f(a0), __builtin_amdgcn_exp2f(a1)};
            const f32x2 v = f32x2{bf16_lo(cw[jp]), bf16_hi(cw[jp])} *
                            f32x2{dt_s[j0 >> 2][j0 & 3], dt_s[j1 >> 2][j1 & 3]} * e;
            o[j0] = (bf16_t)v[0];
            o[j1] = (bf16_t)v[1];
          }
          *(bf16x8*)(Mo + d_off) = o;
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) {          // separable units
          const u32x4 cw = __builtin_bit_cast(u32x4, cbv[1 + u]);
          const f32x2 u2 = {utv[u], utv[u]};
          bf16x8 o;
#pragma unroll
          for (int jp = 0; jp < 4; ++jp) {
            const int j0 = 2 * jp, j1 = 2 * jp + 1;
            const f32x2 v = f32x2{bf16_lo(cw[jp]), bf16_hi(cw[jp])} *
                            (u2 * f32x2{wv[u][j0 >> 2][j0 & 3], wv[u][j1 >> 2][j1 & 3]});
            o[j0] = (bf16_t)v[0];
            o[j1] = (bf16_t)v[1];
          }
          *(bf16x8*)(Mo + s_off[u]) = o;
        }
      };
      // CB fragments are fetched two steps ahead into two register sets (A: even chunks,
      // B: odd chunks), so a fetch has two steps of flight time
      bf16x8 cbA[1 + NS], cbB[1 + NS];
      load_cb(0, cbA);
      load_cb(min(1, nchunks - 1), cbB);
      SLICE_BARRIER();   // P1
      SLICE_BARRIER();   // P2 (chunks 0/1 prepared)
      build(0, cbA);
      load_cb(min(2, nchunks - 1), cbA);
      SLICE_BARRIER();   // P3
      for (int c = 0; c < nchunks; c += 2) {
        if (c + 1 < nchunks && !SDBG(a, 8)) {
          build(c + 1, cbB);
          load_cb(min(c + 3, nchunks - 1), cbB);
        }
        SLICE_BARRIER();
        if (c + 1 < nchunks) {
          if (c + 2 < nchunks && !SDBG(a, 8)) {
            build(c + 2, cbA);
            load_cb(min(c + 4, nchunks - 1), cbA);
          }
          SLICE_BARRIER();
        }
      }
      SLICE_BARRIER();   // final
    };
    if (RL::mask(wave) == 0) run_mask(std::integral_constant<int, 0>{});
    else run_mask(std::integral_constant<int, 1>{});
  } else if (wave == RL::SCALE) {
    // ============================================================ remaining x~ pieces
    SLICE_BARRIER();   // P1
    SLICE_BARRIER();   // P2
#pragma unroll
    for (int k = 4; k < NPI; ++k) scale_piece(0, k);
    SLICE_BARRIER();   // P3
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks && !SDBG(a, 16)) {
#pragma unroll
        for (int k = 4; k < NPI; ++k) scale_piece(c + 1, k);
      }
      SLICE_BARRIER();
    }
    SLICE_BARRIER();   // final
  } else if (!RL::WIDE && wave == RL::PREP) {
    // ============================================================ dt / cumsum prep
    SLICE_BARRIER();   // P1
    prep(0);
    if (nchunks > 1) prep(1);
    SLICE_BARRIER();   // P2
    SLICE_BARRIER();   // P3
    for (int c = 0; c < nchunks; ++c) {
      if (c + 2 < nchunks && !SDBG(a, 16)) prep(c + 2);
      SLICE_BARRIER();
    }
    SLICE_BARRIER();   // final
    float* td = a.nseg > 1 ? a.seg_decay + (int64_t)seg * gridDim.y * a.H : a.total_decay;
    if (td && slice == 0 && lane == 0) td[(int64_t)b * a.H + h] = decay_total;
  } else {
    // idle waves (slice-wave slots of narrower slices)
    SLICE_BARRIER();
    SLICE_BARRIER();
    SLICE_BARRIER();
    for (int c = 0; c < nchunks; ++c) SLICE_BARRIER();
    SLICE_BARRIER();
  }
#ifdef TV_SLICE_STAMP
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
    g_slice_stamps[wave] = stamp_wait;
    g_slice_stamps[16 + wave] = clock64() - stamp_t0;
  }
#endif
}

// wide = whole-head work-groups (one slice of up to 80 columns, 5 slice-waves)
bool pick_slices(int P, int* nslices, int* pw, int wide = 0) {
  if (wide) {
    if (P > 48 && P <= 80 && P % 8 == 0) {
      *nslices = 1;
      *pw = P;
      return true;
    }
    return false;
  }
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
  ssd_slice_kernel<PT, PW><<<grid, Roles<PW>::NWAVES * 64, lds, st>>>(a);
  return hipSuccess;
}

// In how many segments a sequence is marched concurrently: whole-head work-groups need
// batch * heads * segments of them to cover the 256 CUs (Nano-9B: 128 heads -> 2 segments).
int pick_segments(int batch, int nheads, int nchunks) {
  const int wg = batch * nheads;
  int nseg = wg >= 192 ? 1 : 256 / wg;
  if (nseg > 4) nseg = 4;
  if (const char* e = getenv("TV_SSD_NSEG")) nseg = atoi(e);      // dev tool: correction cost against the segment count
  while (nseg > 1 && nchunks / nseg < 16) --nseg;      // short sequences: not worth the fix-up passes
  return nseg < 1 ? 1 : nseg;
}

struct SegLayout {
  size_t cb, seg_state, seg_decay, sin, corr, ctot, total;
  int nseg, seg_chunks;
};
SegLayout seg_layout(int batch, int seqlen, int nheads, int headdim, int ngroups, int wide) {
  SegLayout l;
  const size_t nchunks = (size_t)(seqlen + SQ - 1) / SQ;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  l.nseg = wide ? pick_segments(batch, nheads, (int)nchunks) : 1;
  l.seg_chunks = (int)((nchunks + l.nseg - 1) / l.nseg);
  l.cb = 0;
  l.seg_state = up((size_t)batch * ngroups * nchunks * CB_ELEMS * sizeof(bf16_t));
  const size_t st = (size_t)batch * nheads * headdim * SN * sizeof(float);
  l.seg_decay = l.seg_state + (l.nseg > 1 ? up(l.nseg * st) : 0);
  l.sin = l.seg_decay + (l.nseg > 1 ? up((size_t)l.nseg * batch * nheads * sizeof(float)) : 0);
  l.corr = l.sin + (l.nseg > 2 ? up((l.nseg - 1) * st) : 0);       // 2 segments: S_in(1) = seg_state[0]
  l.ctot = l.corr + (l.nseg > 1 ? up(tv_ssd_correct_workspace_bytes(batch, l.seg_chunks * SQ, nheads)) : 0);
  l.total = l.ctot + (l.nseg > 1 ? up((size_t)batch * nheads * nchunks * sizeof(float)) : 0);
  return l;
}

// S_in of every segment > 0, the final state and the total log-decay from the per-segment results:
//   run = seg_state[0];  for s >= 1:  S_in(s) = run;  run = exp(decay[s]) run + seg_state[s]
__global__ __launch_bounds__(256) void ssd_seg_combine_kernel(const float* __restrict__ seg_state,
                                                              const float* __restrict__ seg_decay,
                                                              float* __restrict__ sin, float* __restrict__ final_state,
                                                              float* __restrict__ total_decay, int nseg,
                                                              int64_t bh, int64_t per_head) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // float4 index
  const int64_t n4 = bh * per_head / 4;
  if (i < n4) {
    const int64_t head = (i * 4) / per_head;
    f32x4 run = ((const f32x4*)seg_state)[i];
    for (int s = 1; s < nseg; ++s) {
      if (sin && nseg > 2) ((f32x4*)sin)[(int64_t)(s - 1) * n4 + i] = run;
      const float e = __expf(seg_decay[(int64_t)s * bh + head]);
      const f32x4 cur = ((const f32x4*)seg_state)[(int64_t)s * n4 + i];
      run = f32x4{e * run[0] + cur[0], e * run[1] + cur[1], e * run[2] + cur[2], e * run[3] + cur[3]};
    }
    if (final_state) ((f32x4*)final_state)[i] = run;
  }
  if (total_decay && i < bh) {
    float t = 0.f;
    for (int s = 0; s < nseg; ++s) t += seg_decay[(int64_t)s * bh + i];
    total_decay[i] = t;
  }
}

}  // namespace

// the C.B^T pre-pass on its own (ssd_head.hip; callers that got no fragments from the conv kernel)
int tv_ssd_cb_prepass_launch(const void* Bm, const void* Cm, void* cb, int batch, int seqlen, int ngroups,
                             int64_t bsb, int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg,
                             hipStream_t st) {
  CbArgs ca;
  ca.Bm = (const bf16_t*)Bm; ca.Cm = (const bf16_t*)Cm; ca.cb = (bf16_t*)cb;
  ca.L = seqlen; ca.G = ngroups; ca.nchunks = (seqlen + SQ - 1) / SQ;
  ca.bsb = bsb; ca.bsl = bsl; ca.bsg = bsg; ca.csb = csb; ca.csl = csl; ca.csg = csg;
  ssd_cb_kernel<<<dim3(ca.nchunks, ngroups, batch), 192, 0, st>>>(ca);
  TV_LAUNCH_CHECK();
}

#ifdef TV_SLICE_STAMP
extern "C" int tv_ssd_slice_debug_stamps(unsigned long long* out) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slice_stamps), sizeof(g_slice_stamps));
  if (e == hipSuccess) e = hipMemcpyFromSymbol(out + 32, HIP_SYMBOL(g_slice_phases), sizeof(g_slice_phases));
  return (int)e;
}
#endif

bool tv_ssd_slice_supported(int seqlen, int nheads, int headdim, int ngroups, int dstate,
                            int dtype, int64_t xsl, int64_t bsl, int64_t bsg, int64_t csl,
                            int64_t csg, int64_t ysl, const void* x, const void* Bm,
                            const void* Cm, const void* y, int wide) {
  int ns, pw;
  if (dtype != TV_BF16 || dstate != SN || seqlen < 1) return false;
  if (!pick_slices(headdim, &ns, &pw, wide)) return false;
  if (xsl % 8 || bsl % 8 || csl % 8 || bsg % 8 || csg % 8 || ysl % 8 || nheads % 2) return false;
  if (((uintptr_t)x & 15) || ((uintptr_t)Bm & 15) || ((uintptr_t)Cm & 15) || ((uintptr_t)y & 15))
    return false;
  if (64 * xsl * 2 >= (1ll << 31) || 64 * bsl * 2 >= (1ll << 31) || 64 * csl * 2 >= (1ll << 31) ||
      64 * ysl * 2 >= (1ll << 31))
    return false;
  (void)ngroups;
  return true;
}

size_t tv_ssd_slice_workspace_bytes(int batch, int seqlen, int nheads, int headdim, int ngroups, int,
                                    int wide) {
  return seg_layout(batch, seqlen, nheads, headdim, ngroups, wide).total;
}

int tv_ssd_slice_launch(const void* x, const void* dt, const void* A, const void* Bm,
                        const void* Cm, const void* D, const void* dt_bias,
                        const void* init_state, void* y, void* final_state, void* total_decay,
                        int batch, int seqlen, int nheads, int headdim, int ngroups, int dstate,
                        int64_t xsb, int64_t xsl, int64_t dsb, int64_t dsl, int64_t bsb,
                        int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg,
                        int64_t ysb, int64_t ysl, int dtype, int dt_softplus, float dt_min,
                        float dt_max, int group_map, void* workspace, size_t workspace_bytes,
                        int wide, const void* cb_pre, hipStream_t st) {
  (void)dtype; (void)dstate;
  const SegLayout lay = seg_layout(batch, seqlen, nheads, headdim, ngroups, wide);
  const size_t need = lay.total;
  TV_CHECK_ARG(workspace && workspace_bytes >= need && (((uintptr_t)workspace) & 15) == 0,
               "ssd_slice: workspace of %zu bytes (16-byte aligned) required, got %zu", need,
               workspace_bytes);
  SliceArgs a;
  a.x = (const bf16_t*)x; a.dt = (const bf16_t*)dt; a.Bm = (const bf16_t*)Bm; a.Cm = (const bf16_t*)Cm;
  a.cb = cb_pre ? (const bf16_t*)cb_pre : (const bf16_t*)workspace;     // C.B^T fragments: the caller's (tv_causal_conv1d_xbc_cb_fwd) or the pre-pass's
  a.A = (const float*)A; a.D = (const float*)D; a.dt_bias = (const float*)dt_bias;
  a.init = (const float*)init_state; a.y = (bf16_t*)y; a.final_state = (float*)final_state;
  a.total_decay = (float*)total_decay;
  a.L = seqlen; a.H = nheads; a.P = headdim; a.G = ngroups;
  a.nchunks = (seqlen + SQ - 1) / SQ;
  if (!pick_slices(headdim, &a.nslices, &a.pw, wide)) TV_UNSUPPORTED("ssd_slice: head_dim %d", headdim);
  unsigned char* wsb = (unsigned char*)workspace;
  a.nseg = lay.nseg; a.seg_chunks = lay.seg_chunks;
  a.seg_state = lay.nseg > 1 ? (float*)(wsb + lay.seg_state) : nullptr;
  a.seg_decay = lay.nseg > 1 ? (float*)(wsb + lay.seg_decay) : nullptr;
  a.chunk_tot = lay.nseg > 1 ? (float*)(wsb + lay.ctot) : nullptr;
  a.xsb = xsb; a.xsl = xsl; a.dsb = dsb; a.dsl = dsl; a.bsb = bsb; a.bsl = bsl; a.bsg = bsg;
  a.csb = csb; a.csl = csl; a.csg = csg; a.ysb = ysb; a.ysl = ysl;
  a.softplus = dt_softplus; a.group_map = group_map; a.dt_min = dt_min; a.dt_max = dt_max;
  { const char* e = getenv("TV_MARCH_DBG"); a.dbg = e ? atoi(e) : 0; }

  CbArgs ca;
  ca.Bm = a.Bm; ca.Cm = a.Cm; ca.cb = (bf16_t*)workspace;
  ca.L = seqlen; ca.G = ngroups; ca.nchunks = a.nchunks;
  ca.bsb = bsb; ca.bsl = bsl; ca.bsg = bsg; ca.csb = csb; ca.csl = csl; ca.csg = csg;
  if (!cb_pre) ssd_cb_kernel<<<dim3(a.nchunks, ngroups, batch), 192, 0, st>>>(ca);

  dim3 grid(nheads * a.nslices, batch, a.nseg);
  hipError_t e = hipSuccess;
  switch (a.pw) {
    case 8: e = launch_slice<1, 8>(a, grid, st); break;
    case 16: e = launch_slice<1, 16>(a, grid, st); break;
    case 24: e = launch_slice<2, 24>(a, grid, st); break;
    case 32: e = launch_slice<2, 32>(a, grid, st); break;
    case 40: e = launch_slice<3, 40>(a, grid, st); break;
    // wide: PT slice-waves x 2 column tiles, 8 waves
    case 56: e = launch_slice<2, 56>(a, grid, st); break;
    case 64: e = launch_slice<2, 64>(a, grid, st); break;
    case 72: e = launch_slice<3, 72>(a, grid, st); break;
    case 80: e = launch_slice<3, 80>(a, grid, st); break;
    default: TV_UNSUPPORTED("ssd_slice: slice width %d", a.pw);
  }
  if (e != hipSuccess) {
    tv_set_error("ssd_slice: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return TV_ERR_LAUNCH;
  }
  if (a.nseg > 1) {
    // segments > 0 marched from a zero state: chain the segment states, then add the carried-in
    // term to their outputs (it stops at each head's decay horizon)
    const int64_t bh = (int64_t)batch * nheads, per_head = (int64_t)headdim * SN;
    float* sin = a.nseg > 2 ? (float*)(wsb + lay.sin) : nullptr;
    const int64_t n4 = bh * per_head / 4;
    ssd_seg_combine_kernel<<<dim3((unsigned)((n4 + 255) / 256)), 256, 0, st>>>(
        a.seg_state, a.seg_decay, sin, (float*)final_state, (float*)total_decay, a.nseg, bh, per_head);
    for (int s = 1; s < a.nseg; ++s) {
      const int64_t t0 = (int64_t)s * a.seg_chunks * SQ;
      if (t0 >= seqlen) break;
      const int len = (int)(seqlen - t0 < (int64_t)a.seg_chunks * SQ ? seqlen - t0 : (int64_t)a.seg_chunks * SQ);
      const float* s_in = a.nseg > 2 ? sin + (int64_t)(s - 1) * bh * per_head : a.seg_state;
      const int rc = tv_ssd_correct_launch((bf16_t*)y + t0 * ysl, (const bf16_t*)dt + t0 * dsl, A,
                                           (const bf16_t*)Cm + t0 * csl, dt_bias, s_in, batch, len, nheads,
                                           headdim, ngroups, ysb, ysl, dsb, dsl, csb, csl, csg, dt_softplus,
                                           dt_min, dt_max, group_map, wsb + lay.corr,
                                           a.chunk_tot + (int64_t)s * a.seg_chunks, a.nchunks, st);
      if (rc != TV_OK) return rc;
    }
  }
  TV_LAUNCH_CHECK();
}
