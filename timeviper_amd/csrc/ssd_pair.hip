// S3 (fast path, v4, "impl 7"): the head-per-wave march of ssd_head.hip at TWO WAVES PER SIMD.
//
// ssd_head.hip gives a head to one 512-register wave, one wave per SIMD: nothing hides a stall, and a 64-token step of 190
// MFMAs (3 040 cycles of matrix pipe) takes ~14 600 cycles — the kernel is bound by its ~1 200 vector instructions and by
// exposed LDS / MFMA -> VALU latencies, not by HBM (with every global access switched off it still takes 85 % of its
// time: profiles/r04_ssd_scan_read_attribution.json).  Here the 80 columns of a head are split over TWO waves of <= 256
// registers that share a SIMD's issue slots — columns 0..47 (3 tiles: 96 + 48 accumulation registers) and 48..79 (2 tiles:
// 64 + 32) — so that one wave's waits sit under the other's MFMAs and vector work (MI355X guide, "two waves per SIMD:
// pair matrix with memory").  A work-group = 8 waves = the 4 heads of a B/C group x 2 column halves:
//   * B / C tiles: one LDS ring of 2 for all eight waves (each wave copies an eighth of the pieces);
//   * x tile [64 tokens][80 columns]: one ring of 2 per HEAD, copied by the head's two waves (6 + 5 LDS-DMA instructions, the
//     last one EXEC-masked to the tile's 64 rows: no overhang, the LDS budget is 512 bytes short of it);
//   * the per-chunk vectors of a head (decay sums, weights, row factors, mode of the step) are prepared ONCE, by the
//     2-tile wave (it has a third fewer MFMAs), into LDS by chunk parity; its partner reads them behind the step's barrier;
//   * fragments are single-buffered (a quarter = C / B^T fragment reads, the bf16 copy of 32 state rows, 8 PT MFMAs):
//     112 vector registers beside 144 accumulation registers; the in-wave latency this exposes is what the partner hides.
// Arithmetic, frames (floating / re-based / reset / standard steps), layouts of B, C, C.B^T and of the state, sequence
// segments and their carried-in correction are those of ssd_head.hip — see there.  head_dim 80, d_state 128, bf16,
// heads per group a multiple of 4.
// Reference semantics: mamba_chunk_scan_combined call modeling_nano.py:639-653; arithmetic :775-851.
#include <stdlib.h>
#include <type_traits>
#include "ssd_common.hpp"

// ssd_slice.hip
int tv_ssd_cb_prepass_launch(const void* Bm, const void* Cm, void* cb, int batch, int seqlen, int ngroups,
                             int64_t bsb, int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg,
                             hipStream_t st);
// ssd_correct.hip
size_t tv_ssd_correct_all_workspace_bytes(int batch, int nheads, int nchunks, int nseg, int headdim);
int tv_ssd_correct_all_launch(void* y, const void* dt, const void* A, const void* Cm, const void* dt_bias,
                              const float* seg_state, const float* seg_decay, float* final_state,
                              float* total_decay, const float* chunk_tot, int batch, int seqlen, int nheads,
                              int headdim, int ngroups, int nseg, int seg_chunks, int64_t ysb, int64_t ysl,
                              int64_t dsb, int64_t dsl, int64_t dsh, int64_t csb, int64_t csl, int64_t csg, int dt_softplus,
                              float dt_min, float dt_max, int group_map, void* workspace, hipStream_t st);
// ssd_head.hip
void tv_ssd_dt_transpose_launch(const void* dt, void* out, int batch, int seqlen, int nheads, int64_t dsb, int64_t dsl,
                                hipStream_t st);

namespace {
using namespace ssdk;

constexpr int HQ = 64;            // tokens per chunk
constexpr int HN = 128;           // d_state
constexpr int HP = 80;            // head_dim
constexpr int NFR = 6;            // causal (t-tile, s-pair) fragments of a 64x64 chunk
constexpr int CBE = NFR * 512;    // bf16 elements of C.B^T per (chunk, group)
constexpr float RMAX = 100.f;
constexpr float RESET_THR = 64.f;
constexpr int XROW = 2 * HP;      // bytes per x / y row of a head
constexpr int NPC = HP / 8;       // 16-byte pieces per row (10)
constexpr int RPI = 64 / NPC;     // whole rows per copy instruction (6)
constexpr int NXI = (HQ + RPI - 1) / RPI;   // copy instructions per x tile (11)

struct PairArgs {
  const bf16_t *x, *dt, *Bm, *Cm, *cb;
  const float *A, *D, *dt_bias, *init;
  bf16_t* y;
  float *final_state, *total_decay;
  float *seg_state, *seg_decay, *chunk_tot;
  int nseg, seg_chunks;
  int L, H, G, nchunks;
  int64_t xsb, xsl, dsb, dsl, dsh, bsb, bsl, bsg, csb, csl, csg, ysb, ysl;
  int softplus, group_map;
  float dt_min, dt_max;
};

struct __attribute__((aligned(16))) PairVec {   // per head and chunk parity
  float cs[HQ];       // inclusive cumsum of dt A inside the chunk, times log2(e)
  float dtv[HQ];      // discretised dt
  float ut[HQ];       // standard steps: row factor of the separable off-diagonal mask blocks
  float wts[HQ];      // weight of token s in the state update (frame-dependent)
  float wtd[HQ];      // reset steps: weight of token s in Ydiag (the old frame's)
  float ecs[HQ];      // 2^(cs_t + E): row factor of Yoff
  float ws[96];       // standard steps: column factors of the separable blocks: t-tile 1 at [0,16), 2 at [16,48), 3 at [48,96)
  int shift;          // != 0: X' *= 2^shift before the step (re-basing of the frame)
  int reset;          // the chunk builds its state anew
  int stdstep;        // ... as a standard step
  float e_after;      // frame after this chunk
  int pad[12];
};
static_assert(sizeof(PairVec) == 1984, "PairVec");

struct __attribute__((aligned(16))) PairSmem {
  bf16_t bt[2][HQ * HN];       // B tiles [t][n], 16-byte chunk index ^ 4 (t & 3)
  bf16_t ct[2][HQ * HN];       // C tiles [t][n], chunk ^ (t & 15)
  bf16_t xr[4][2][HQ * HP];    // x tiles [t][80] of the 4 heads, ring of 2
  PairVec v[4][2];
};
static_assert(sizeof(PairSmem) <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int xad(int a, int k, int b) {     // (a ^ k) + b, k wave-uniform
  int d;
  asm("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(b));
  return d;
}
__device__ __forceinline__ float rdlane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// One wave's march.  PT column tiles starting at tile CT0 of head (g, hig); PREP: this wave prepares the head's vectors.
template <int PT, int CT0, bool PREP>
__device__ __forceinline__ void pair_march(const PairArgs& a, PairSmem& sm, unsigned lds0, int wave, int lane) {
  const int hi4 = wave & 3;                // head of the work-group
  const int lc = lane & 15, kq = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int b = blockIdx.y;
  const int hpg = a.H / a.G;
  const int g = blockIdx.x % a.G;
  const int hig = (blockIdx.x / a.G) * 4 + hi4;
  const int h = a.group_map ? (hig * a.G + g) : (g * hpg + hig);
  const int seg = blockIdx.z;
  const int c_first = seg * a.seg_chunks;
  const int t_first = c_first * HQ;
  const int nchunks = min(a.seg_chunks, a.nchunks - c_first);
  const int L = min(a.L - t_first, nchunks * HQ);
  const unsigned lds_bt = lds0 + (unsigned)offsetof(PairSmem, bt), lds_ct = lds0 + (unsigned)offsetof(PairSmem, ct);
  const unsigned lds_xr = lds0 + (unsigned)offsetof(PairSmem, xr) + hi4 * (unsigned)sizeof(sm.xr[0]);
  constexpr unsigned XSLOT = sizeof(sm.xr[0][0]);

  // ------------------------------------------------------------------ B / C copies: 2 pieces of each per wave and chunk
  const bf16_t* Bg = a.Bm + (int64_t)b * a.bsb + (int64_t)g * a.bsg + (int64_t)t_first * a.bsl;
  const bf16_t* Cg = a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg + (int64_t)t_first * a.csl;
  // piece k of this wave (k = 0, 1) = token rows 8 wave + 4 k + (lane >> 4), 16-byte chunk lane & 15 of the row, stored
  // swizzled: B chunk ^ 4 (row & 3) (the same for both k), C chunk ^ (row & 15) (k flips bit 2 of it)
  // (the lane offsets of the copies are re-derived from the lane index at every use: kept across the step they are the
  // first registers the allocator spills, and a reload in front of a copy waits for every copy before it)
  auto issue_bc = [&](int c, int which) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int bc_row0 = 8 * wave + (ln >> 4);
    const unsigned off_b0 = (unsigned)((bc_row0 * (int)a.bsl + ((ln & 15) ^ (4 * ((ln >> 4) & 3))) * 8) * 2);
    const unsigned off_c0 = (unsigned)((bc_row0 * (int)a.csl + ((ln & 15) ^ (bc_row0 & 15)) * 8) * 2);
    const int t0 = c * HQ;
    const bf16_t* Tc = which ? Cg + (int64_t)t0 * a.csl : Bg + (int64_t)t0 * a.bsl;
    const int64_t rl = which ? a.csl : a.bsl;
    const unsigned dst = (which ? lds_ct : lds_bt) + (c & 1) * (HQ * HN * 2) + 2 * wave * 1024;
    if (t0 + HQ <= L) {       // whole chunk: ONE M0 set-up and scalar base for the two pieces; the instruction offset of the
                              // second moves the source and the LDS address alike, its lane offset makes up the difference
      const unsigned r4 = (unsigned)(4 * rl * 2) - 1024u;      // bytes of 4 rows in memory - bytes of a piece (rows >= 256 B)
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %3\n\t"
                   "global_load_lds_dwordx4 %2, %3 offset:1024\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(which ? off_c0 : off_b0), "v"((which ? (off_c0 ^ 64u) : off_b0) + r4),
                     "s"(uniform_ptr(Tc)), "s"(dst) : "memory");
      return;
    }
    const void* sp = uniform_ptr(Tc);
#pragma unroll
    for (int k = 0; k < 2; ++k) {      // last, partial chunk: rows past the end repeat the last row (finite)
      const int row = 8 * wave + 4 * k + (ln >> 4);
      const int rr = min(row, L - 1 - t0);
      const int cg = which ? (ln & 15) ^ (row & 15) : (ln & 15) ^ (4 * (row & 3));
      glds16(sp, (unsigned)((rr * rl + cg * 8) * 2), dst + k * 1024);
    }
  };

  // ------------------------------------------------------------------ x: the head's tile, copied by its two waves
  const bf16_t* xg = a.x + (int64_t)b * a.xsb + (int64_t)t_first * a.xsl + (int64_t)h * HP;
  constexpr int XI0 = CT0 == 0 ? 0 : 6, XI1 = CT0 == 0 ? 6 : NXI;       // this wave's copy instructions
  // whole chunks: instruction k = XI0 + j copies rows RPI k .. of the tile; groups of up to four share ONE M0 set-up and scalar
  // base, the instruction offset (j RPI XROW <= 2 880) moves both addresses, lane offset j dj makes up the difference
  auto issue_x = [&](int c) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int x_lrow = ln / NPC;
    const unsigned x_off = (unsigned)((x_lrow * (int)a.xsl + (ln - NPC * x_lrow) * 8) * 2);
    const int t0 = c * HQ;
    const bf16_t* xc = xg + (int64_t)t0 * a.xsl;
    const unsigned dst = lds_xr + (c & 1) * XSLOT;
    if (t0 + HQ <= L) {
      const unsigned dj = (unsigned)(RPI * a.xsl * 2) - (unsigned)(RPI * XROW);
      const void* s0 = uniform_ptr(xc + (int64_t)RPI * XI0 * a.xsl);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %5\n\t"
                   "global_load_lds_dwordx4 %2, %5 offset:%7\n\t"
                   "global_load_lds_dwordx4 %3, %5 offset:%8\n\t"
                   "global_load_lds_dwordx4 %4, %5 offset:%9\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(x_off), "v"(x_off + dj), "v"(x_off + 2 * dj), "v"(x_off + 3 * dj), "s"(s0),
                     "s"(dst + RPI * XI0 * XROW), "n"(RPI * XROW), "n"(2 * RPI * XROW), "n"(3 * RPI * XROW) : "memory");
      const void* s1 = uniform_ptr(xc + (int64_t)RPI * (XI0 + 4) * a.xsl);
      if (CT0 == 0)       // instructions 4, 5
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\t"
                     "global_load_lds_dwordx4 %2, %3 offset:%5\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(x_off), "v"(x_off + dj), "s"(s1), "s"(dst + RPI * (XI0 + 4) * XROW), "n"(RPI * XROW) : "memory");
      else                // instruction 10: rows 60 .. 63 only (lanes 0 .. 39): nothing lands behind the tile
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_mov_b32 exec_hi, 0xff\n\ts_nop 1\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(x_off), "s"(s1), "s"(dst + RPI * (XI0 + 4) * XROW) : "memory");
      return;
    }
    const void* sp = uniform_ptr(xc);
#pragma unroll
    for (int k = XI0; k < XI1; ++k) {      // last, partial chunk: the rows are clamped per lane
      const unsigned vo = (unsigned)((min(RPI * k + x_lrow, L - 1 - t0) * (int)a.xsl + (ln - NPC * x_lrow) * 8) * 2);
      if (k < NXI - 1) glds16(sp, vo, dst + RPI * k * XROW);
      else {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_mov_b32 exec_hi, 0xff\n\ts_nop 1\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(sp), "s"(dst + RPI * k * XROW) : "memory");
      }
    }
  };
  // transposing reads (MFMA operand with k = token): lane (lc = column, kq) gets tokens 32 ks + 8 kq + 0..7 of column
  // 16 ct + lc (ct: tile of the HEAD)
  const int xr_lo = (8 * kq + q4) * XROW + 8 * p4;
  auto read_xf = [&](unsigned xt, int ct, int ks) __attribute__((always_inline)) {
    const unsigned p = xt + xr_lo + 32 * (CT0 + ct) + ks * (32 * XROW);
    return cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(size_t)p), __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(size_t)(p + 4 * XROW)));
  };
  const int xv_lo = lc * XROW + 8 * kq;
  bf16_t* const ygs = a.y + (int64_t)b * a.ysb + (int64_t)t_first * a.ysl + (int64_t)h * HP;
  const unsigned yoff16 = (unsigned)(((16 * (kq & 1) + lc) * a.ysl + 8 * (kq >> 1)) * 2);

  // ------------------------------------------------------------------ per-chunk vectors (the PREP wave; lane = token)
  const float Ah = a.A[h];
  const float bias = a.dt_bias ? a.dt_bias[h] : 0.f;
  const float Dh = a.D ? a.D[h] : 0.f;
  const bf16_t* dtg = (const bf16_t*)uniform_ptr(a.dt + (int64_t)b * a.dsb + (int64_t)t_first * a.dsl + (int64_t)h * a.dsh);
  float decay_total = 0.f;
  float E = 0.f;                  // X = 2^E X' (tracked by the PREP wave; the partner reads e_after)
  auto load_dt = [&](int c) __attribute__((always_inline)) {
    const int t = min(c * HQ + lane, L - 1);
    return (unsigned)*(const unsigned short*)(dtg + (int64_t)t * a.dsl);
  };
  auto prep = [&](int c, unsigned raw_bits) __attribute__((always_inline)) {
    PairVec& vec = sm.v[hi4][c & 1];
    const int t = c * HQ + lane;
    float d = 0.f;
    if (t < L) {
      d = bf16_lo(raw_bits) + bias;
      if (a.softplus) d = softplus_fast(d);
      d = fminf(fmaxf(d, a.dt_min), a.dt_max);
    }
    const float cs = wave_incl_scan_dpp(d * Ah);
    const float cl = rdlane(cs, 63);
    const float cs2 = cs * 1.4426950408889634f, cl2 = cl * 1.4426950408889634f;
    const int mode = -(E + cl2) <= RMAX ? 0 : -cl2 <= 2.f * RMAX - 1.f ? 1 : 2;
    const float mshift = mode == 1 ? __builtin_floorf(RMAX - E) : 0.f;
    const float Euse = E + mshift;
    vec.cs[lane] = cs2;
    vec.dtv[lane] = d;
    vec.ecs[lane] = __builtin_amdgcn_exp2f(cs2 + Euse);
    const bool rst = mode != 2 && cl2 <= -RESET_THR;
    const bool ustd = mode == 2;
    vec.wts[lane] = __builtin_amdgcn_exp2f(mode == 2 ? cl2 - cs2 : rst ? cl2 - cs2 - RMAX : -cs2 - Euse) * d;
    vec.wtd[lane] = __builtin_amdgcn_exp2f(-cs2 - Euse) * d;
    if (a.chunk_tot && lane == 0) a.chunk_tot[((int64_t)b * a.H + h) * a.nchunks + c_first + c] = cl2;
    if (__builtin_expect(mode == 2, 0)) {
      const float p0 = rdlane(cs2, 0), p1 = rdlane(cs2, 16), p2 = rdlane(cs2, 32), p3 = rdlane(cs2, 48);
      const float pv = lane < 16 ? p0 : lane < 32 ? p1 : lane < 48 ? p2 : p3;
      vec.ut[lane] = __builtin_amdgcn_exp2f(fminf(cs2 - pv, 0.f));
      if (lane < 16) vec.ws[lane] = __builtin_amdgcn_exp2f(fminf(p1 - cs2, 0.f)) * d;
      if (lane < 32) vec.ws[16 + lane] = __builtin_amdgcn_exp2f(fminf(p2 - cs2, 0.f)) * d;
      if (lane < 48) vec.ws[48 + lane] = __builtin_amdgcn_exp2f(fminf(p3 - cs2, 0.f)) * d;
    }
    decay_total += cl;
    E = mode == 2 ? 0.f : rst ? RMAX : Euse + cl2;
    if (lane == 0) {
      vec.shift = mode == 1 ? -(int)mshift : 0;
      vec.reset = (rst || ustd) ? 1 : 0;
      vec.stdstep = ustd ? 1 : 0;
      vec.e_after = E;
    }
  };

  // ------------------------------------------------------------------ state (accumulation registers; see ssd_head.hip)
  f32x4 xacc[PT][8];
#pragma unroll
  for (int ct = 0; ct < PT; ++ct) {
#pragma unroll
    for (int i = 0; i < 8; ++i) xacc[ct][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.init && seg == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        xacc[ct][i] = *(const f32x4*)(a.init + (((int64_t)b * a.H + h) * HP + 16 * (CT0 + ct) + lc) * HN + 32 * (i >> 1) + 8 * kq + 4 * (i & 1));
    }
  }
  auto snap_tile = [&](int q, int ct) __attribute__((always_inline)) {
    bf16x8 sb;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      float t0, t1, t2, t3;
      asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                   : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3)
                   : "a"(xacc[ct][2 * q + ii][0]), "a"(xacc[ct][2 * q + ii][1]), "a"(xacc[ct][2 * q + ii][2]), "a"(xacc[ct][2 * q + ii][3]));
      sb[4 * ii + 0] = (bf16_t)t0;
      sb[4 * ii + 1] = (bf16_t)t1;
      sb[4 * ii + 2] = (bf16_t)t2;
      sb[4 * ii + 3] = (bf16_t)t3;
    }
    return sb;
  };
  auto rebase_state = [&](int sh) __attribute__((always_inline)) {
#pragma unroll
    for (int ct = 0; ct < PT; ++ct)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float e0 = xacc[ct][i][0], e1 = xacc[ct][i][1], e2 = xacc[ct][i][2], e3 = xacc[ct][i][3], t0, t1, t2, t3;
        asm volatile("v_accvgpr_read_b32 %4, %0\n\tv_accvgpr_read_b32 %5, %1\n\tv_accvgpr_read_b32 %6, %2\n\tv_accvgpr_read_b32 %7, %3\n\t"
                     "v_ldexp_f32 %4, %4, %8\n\tv_ldexp_f32 %5, %5, %8\n\tv_ldexp_f32 %6, %6, %8\n\tv_ldexp_f32 %7, %7, %8\n\t"
                     "v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
                     : "+a"(e0), "+a"(e1), "+a"(e2), "+a"(e3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(sh));
        xacc[ct][i] = f32x4{e0, e1, e2, e3};
      }
    asm volatile("s_nop 7" ::: "memory");
  };
  auto zero_state = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ct = 0; ct < PT; ++ct)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // (read-write operands: the tile stays in its registers — a fresh output would meet the untouched tile in a PHI)
        float e0 = xacc[ct][i][0], e1 = xacc[ct][i][1], e2 = xacc[ct][i][2], e3 = xacc[ct][i][3];
        asm volatile("v_accvgpr_write_b32 %0, 0\n\tv_accvgpr_write_b32 %1, 0\n\tv_accvgpr_write_b32 %2, 0\n\tv_accvgpr_write_b32 %3, 0"
                     : "+a"(e0), "+a"(e1), "+a"(e2), "+a"(e3));
        xacc[ct][i] = f32x4{e0, e1, e2, e3};
      }
    asm volatile("s_nop 7" ::: "memory");
  };
  const int c_lo = lc * 256;
  const int c_z = (kq ^ lc) << 4;
  const int bsw = q4 << 6;
  const int b_lo = (8 * kq + q4) * 256 + p4 * 16;
  const bf16_t* cbg = (const bf16_t*)uniform_ptr(a.cb + (((int64_t)b * a.G + g) * a.nchunks + c_first) * CBE);

  // ------------------------------------------------------------------ prologue
  issue_bc(0, 0);
  issue_bc(0, 1);
  issue_x(0);
  unsigned dt_next = 0;
  if (PREP) {
    prep(0, load_dt(0));
    dt_next = load_dt(min(1, nchunks - 1));
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    const PairVec& vec = sm.v[hi4][c & 1];
    const int shift = __builtin_amdgcn_readfirstlane(vec.shift);
    const bool reset_step = __builtin_amdgcn_readfirstlane(vec.reset) != 0;
    const bool ustd_step = __builtin_amdgcn_readfirstlane(vec.stdstep) != 0;
    if (shift != 0) rebase_state(shift);
    const unsigned char* Bt = reinterpret_cast<const unsigned char*>(sm.bt[c & 1]);
    const unsigned char* Ct = reinterpret_cast<const unsigned char*>(sm.ct[c & 1]);
    const unsigned xt = lds_xr + (c & 1) * XSLOT;
    bf16x8 cbv[NFR];                // causal C.B^T of this chunk (global, L2: requested first, used after pass 1)
#pragma unroll
    for (int f = 0; f < NFR; ++f) cbv[f] = *(const bf16x8*)(cbg + (int64_t)c * CBE + f * 512 + lane * 8);
    // The step runs as THREE passes so that few fragments are live at a time (112 vector registers a wave):
    //   1. Yoff^T = X'^T C^T for the four quarters of 32 state rows (C fragments + ONE bf16 copy of 32 state rows of one
    //      column tile live), 2. x~ = w_s x, then X' += B^T x~ quarter by quarter, 3. Ydiag with the causal C.B^T.
    f32x4 yo[PT][4];              // (first written by quarter 0's MFMAs: C = 0)
    unsigned dt_raw = dt_next;
    {
      bf16x8 cf[2][4];
      {
        const unsigned char* cp = Ct + xad(c_z, 0, c_lo);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) cf[0][ti] = ld8(cp + ti * 4096);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < 3) {        // the next quarter's C fragments are in flight under this quarter's MFMAs
          const unsigned char* cp = Ct + xad(c_z, 64 * (q + 1), c_lo);
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) cf[(q + 1) & 1][ti] = ld8(cp + ti * 4096);
        }
#pragma unroll
        for (int ct = 0; ct < PT; ++ct) {
          const bf16x8 sb = snap_tile(q, ct);
#pragma unroll
          for (int j = 0; j < 4; ++j) yo[ct][j] = mfma16(sb, cf[q & 1][j], q == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : yo[ct][j]);
        }
        // copies of the next chunk between the quarters: B, C behind quarter 0 / 1, x behind quarter 2
        if (more && q == 0) issue_bc(c + 1, 0);
        if (more && q == 1) issue_bc(c + 1, 1);
        if (more && q == 2) issue_x(c + 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- order of the rest of the step: Ydiag and the y stores FIRST (they need x~ and C.B^T, not the new state; the stores
    // then have the whole state update to drain, and the y accumulators are free during it), the state update last
    float ev[4];                    // 2^(cs_t + E) of this lane's token 16 ti + lc
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) ev[ti] = vec.ecs[16 * ti + lc];
    // ---- x~ = w_s x on the fragments (element j of fragment ks is token 32 ks + 8 kq + j)
    bf16x8 xw[PT][2];
    auto make_xw = [&](const float* wsrc) __attribute__((always_inline)) {
      f32x4 wq[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        wq[ks][0] = *(const f32x4*)(wsrc + 32 * ks + 8 * kq);
        wq[ks][1] = *(const f32x4*)(wsrc + 32 * ks + 8 * kq + 4);
      }
#pragma unroll
      for (int ct = 0; ct < PT; ++ct)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const u32x4v u = __builtin_bit_cast(u32x4v, read_xf(xt, ct, ks));
          u32x4v o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            const f32x2 pr = f32x2{bf16_lo(u[e]), bf16_hi(u[e])} * f32x2{wq[ks][e >> 1][(2 * e) & 3], wq[ks][e >> 1][(2 * e + 1) & 3]};
            const bf16x2 pk = {(bf16_t)pr[0], (bf16_t)pr[1]};
            o[e] = __builtin_bit_cast(unsigned, pk);
          }
          xw[ct][ks] = __builtin_bit_cast(bf16x8, o);
        }
    };
    // x~ for Ydiag: the frame the accumulators are in (a reset step's state update runs in the NEW frame: x~ is formed again
    // below; a standard step takes raw x here)
    // Ydiag + the y stores are written out once per kind of step (as a lambda inlined into both branches): joined behind
    // the branch, the scaled accumulators, the masked fragments and the raw x fragments of a standard step would meet the
    // floating step's registers in PHIs — copies of 48 accumulation registers and spills of state tiles
    const bool full = (c + 1) * HQ <= L;
    auto ydiag_and_store = [&](const float (&ev)[4]) __attribute__((always_inline)) {
      // ---- Ydiag on top of Yoff, same frame: the A operand is x~, the B operand the causal C.B^T fragment
  #pragma unroll
      for (int ti = 0; ti < 4; ++ti)
  #pragma unroll
        for (int ct = 0; ct < PT; ++ct) yo[ct][ti] = mfma16(xw[ct][0], cbv[ti == 0 ? 0 : ti == 1 ? 1 : ti == 2 ? 2 : 4], yo[ct][ti]);
  #pragma unroll
      for (int ti = 2; ti < 4; ++ti)
  #pragma unroll
        for (int ct = 0; ct < PT; ++ct) yo[ct][ti] = mfma16(xw[ct][1], cbv[ti == 2 ? 3 : 5], yo[ct][ti]);
      // ---- y = row factor * accumulators + D x, rounded to bf16 and stored (16 bytes a lane: two t-tiles joined)
      const f32x2 dh2 = {Dh, Dh};
      auto finish = [&](int ct, int ti) __attribute__((always_inline)) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) const u32x2 lds_u32x2;
        const u32x2 xr = *(lds_u32x2*)(size_t)(xt + xv_lo + 32 * (CT0 + ct) + ti * (16 * XROW));
        const f32x2 e2 = {ev[ti], ev[ti]};
        const f32x2 y0 = __builtin_elementwise_fma(f32x2{yo[ct][ti][0], yo[ct][ti][1]}, e2, dh2 * f32x2{bf16_lo(xr[0]), bf16_hi(xr[0])});
        const f32x2 y1 = __builtin_elementwise_fma(f32x2{yo[ct][ti][2], yo[ct][ti][3]}, e2, dh2 * f32x2{bf16_lo(xr[1]), bf16_hi(xr[1])});
        const bf16x2 p01 = {(bf16_t)y0[0], (bf16_t)y0[1]}, p23 = {(bf16_t)y1[0], (bf16_t)y1[1]};
        return u32x2{__builtin_bit_cast(unsigned, p01), __builtin_bit_cast(unsigned, p23)};
      };
  #pragma unroll
      for (int tp = 0; tp < 4; tp += 2) {
        const void* yrow = uniform_ptr(ygs + (int64_t)(c * HQ + 16 * tp) * a.ysl);
        const bool ok = full || c * HQ + 16 * tp + 16 * (kq & 1) + lc < L;
  #pragma unroll
        for (int ct = 0; ct < PT; ++ct) {
          const u32x2 ya = finish(ct, tp), yb = finish(ct, tp + 1);
          const auto s0 = __builtin_amdgcn_permlane16_swap(ya[0], yb[0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(ya[1], yb[1], false, false);
          const u32x4v w = {s0[0], s1[0], s0[1], s1[1]};
          if (ok) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3" :: "v"(yoff16), "v"(w), "s"(yrow), "n"(32 * (CT0 + ct)) : "memory");
        }
      }
    };
    if (!ustd_step) {
      make_xw(reset_step ? vec.wtd : vec.wts);
      ydiag_and_store(ev);
    } else {
      // ---- standard step: (1) the accumulators get their row factor now, (2) the per-head mask replaces C.B^T in its
      // registers, (3) the A operand becomes the raw x fragments (ssd_head.hip)
#pragma unroll
      for (int ct = 0; ct < PT; ++ct)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
          float e0 = yo[ct][ti][0], e1 = yo[ct][ti][1], e2 = yo[ct][ti][2], e3 = yo[ct][ti][3], t0, t1, t2, t3;
          asm volatile("v_accvgpr_read_b32 %4, %0\n\tv_accvgpr_read_b32 %5, %1\n\tv_accvgpr_read_b32 %6, %2\n\tv_accvgpr_read_b32 %7, %3\n\t"
                       "v_mul_f32 %4, %8, %4\n\tv_mul_f32 %5, %8, %5\n\tv_mul_f32 %6, %8, %6\n\tv_mul_f32 %7, %8, %7\n\t"
                       "v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
                       : "+a"(e0), "+a"(e1), "+a"(e2), "+a"(e3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(ev[ti]));
          yo[ct][ti] = f32x4{e0, e1, e2, e3};
        }
      asm volatile("s_nop 7" ::: "memory");
      {
        const int hi = kq >> 1;
        auto diag = [&](int t, int s0, float (&e)[8]) __attribute__((always_inline)) {
          const float cst = vec.cs[t];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const f32x4 cv = *(const f32x4*)(&vec.cs[s0 + 4 * hh]), dv = *(const f32x4*)(&vec.dtv[s0 + 4 * hh]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              e[4 * hh + j] = __builtin_amdgcn_exp2f(s0 + 4 * hh + j <= t ? cst - cv[j] : -__builtin_inff()) * dv[j];
          }
        };
        auto sepf = [&](int t, int wofs, float (&e)[8]) __attribute__((always_inline)) {
          const float u = vec.ut[t];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const f32x4 wv = *(const f32x4*)(&vec.ws[wofs + 4 * hh]);
#pragma unroll
            for (int j = 0; j < 4; ++j) e[4 * hh + j] = u * wv[j];
          }
        };
        auto apply = [&](int f, const float (&fac)[8]) __attribute__((always_inline)) {
          const u32x4v cw = __builtin_bit_cast(u32x4v, cbv[f]);
          u32x4v o;
#pragma unroll
          for (int jp = 0; jp < 4; ++jp) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            const bf16x2 pk = {(bf16_t)(bf16_lo(cw[jp]) * fac[2 * jp]), (bf16_t)(bf16_hi(cw[jp]) * fac[2 * jp + 1])};
            o[jp] = __builtin_bit_cast(unsigned, pk);
          }
          cbv[f] = __builtin_bit_cast(bf16x8, o);
        };
        // one fragment at a time, every factor formed where it is used (few live registers: the diagonal factors of a lane
        // half are formed twice rather than kept)
        const int sA = 8 * kq;
        {
          float fac[8];
          diag(16 * hi + lc, sA, fac);                          // fragment (0,0): diagonal block for hi = 0, nothing for hi = 1
#pragma unroll
          for (int j = 0; j < 8; ++j) fac[j] = hi ? 0.f : fac[j];
          apply(0, fac);
        }
        {
          float fac[8], eS[8];
          diag(16 * hi + lc, sA, fac);                          // fragment (1,0): diagonal block for hi = 1, separable for hi = 0
          sepf(16 + lc, sA & 15, eS);
#pragma unroll
          for (int j = 0; j < 8; ++j) fac[j] = hi ? fac[j] : eS[j];
          apply(1, fac);
        }
        {
          float eS[8];
          sepf(32 + lc, 16 + sA, eS);                            // blocks (2,0), (2,1)
          apply(2, eS);
        }
        {
          float fac[8];
          diag(32 + 16 * hi + lc, 32 + sA, fac);                // fragment (2,1) diagonal for hi = 0
#pragma unroll
          for (int j = 0; j < 8; ++j) fac[j] = hi ? 0.f : fac[j];
          apply(3, fac);
        }
        {
          float eS[8];
          sepf(48 + lc, 48 + sA, eS);                            // blocks (3,0), (3,1)
          apply(4, eS);
        }
        {
          float fac[8], eS[8];
          diag(32 + 16 * hi + lc, 32 + sA, fac);                // fragment (3,1): diagonal for hi = 1, separable (3,2) for hi = 0
          sepf(48 + lc, 48 + 32 + (sA & 15), eS);
#pragma unroll
          for (int j = 0; j < 8; ++j) fac[j] = hi ? fac[j] : eS[j];
          apply(5, fac);
        }
      }
#pragma unroll
      for (int ct = 0; ct < PT; ++ct)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xw[ct][ks] = read_xf(xt, ct, ks);
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) ev[ti] = 1.f;
      ydiag_and_store(ev);
    }
    // (the copies of the next chunk were issued long ago; this step's y stores stay in flight where the chunk is whole)
    __builtin_amdgcn_sched_barrier(0);
    if (reset_step) make_xw(vec.wts);        // (standard steps are reset steps: their weights onto zero)
    // ---- X' += B^T x~; a reset step drops the old state first (every bf16 copy of it has been taken in pass 1)
    if (reset_step) zero_state();
    {
      bf16x4 bt[2][8];            // B^T of a quarter: state tiles 2 q (0..3) and 2 q + 1 (4..7), k-steps 0, 1
      auto read_bt = [&](int q, bf16x4 (&d)[8]) __attribute__((always_inline)) {
        const unsigned char* bp = Bt + xad(bsw, 64 * q, b_lo);
        d[0] = tr4(bp);
        d[1] = tr4(bp + 1024);
        d[2] = tr4(bp + 8192);
        d[3] = tr4(bp + 8192 + 1024);
        d[4] = tr4(bp + 8);
        d[5] = tr4(bp + 8 + 1024);
        d[6] = tr4(bp + 8 + 8192);
        d[7] = tr4(bp + 8 + 8192 + 1024);
      };
      read_bt(0, bt[0]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < 3) read_bt(q + 1, bt[(q + 1) & 1]);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 bfrag = cat4(bt[q & 1][4 * ii + 2 * ks], bt[q & 1][4 * ii + 2 * ks + 1]);
#pragma unroll
            for (int ct = 0; ct < PT; ++ct) xacc[ct][2 * q + ii] = mfma16(bfrag, xw[ct][ks], xacc[ct][2 * q + ii]);
          }
        if (PREP && q == 2) dt_next = load_dt(min(c + 2, nchunks - 1));
      }
    }
    if (PREP && more) prep(c + 1, dt_raw);
    if (full) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PT + (PREP ? 1 : 0)) : "memory");      // (+ the dt load behind the stores)
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // final state of this segment, X = 2^E X'
  {
    const float e_fin = nchunks > 0 ? sm.v[hi4][(nchunks - 1) & 1].e_after : 0.f;
    const float sc = __builtin_amdgcn_exp2f(e_fin);
    float* fin = a.nseg > 1 ? a.seg_state + (int64_t)seg * gridDim.y * a.H * HP * HN : a.final_state;
    if (fin) {
#pragma unroll
      for (int ct = 0; ct < PT; ++ct)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 v = xacc[ct][i];
          *(f32x4*)(fin + (((int64_t)b * a.H + h) * HP + 16 * (CT0 + ct) + lc) * HN + 32 * (i >> 1) + 8 * kq + 4 * (i & 1)) =
              f32x4{v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc};
        }
    }
    if (PREP) {
      float* td = a.nseg > 1 ? a.seg_decay + (int64_t)seg * gridDim.y * a.H : a.total_decay;
      if (td && lane == 0) td[(int64_t)b * a.H + h] = decay_total;
    }
  }
}

__global__ __launch_bounds__(512) void ssd_pair_kernel(PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  PairSmem& sm = *reinterpret_cast<PairSmem*>(smem_raw);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = lds_addr_of(smem_raw);
  if (wave < 4) pair_march<3, 0, false>(a, sm, lds0, wave, lane);
  else pair_march<2, 3, true>(a, sm, lds0, wave, lane);
}

struct PairLayout {
  size_t seg_state, seg_decay, ctot, corr, dtt, total;
  int nseg, seg_chunks;
};
int pair_segments(int batch, int nheads, int nchunks) {
  if (const char* e = getenv("TV_SSD_NSEG")) return atoi(e) > 0 ? atoi(e) : 1;     // dev tool
  const int waves = batch * nheads;                  // (whole heads: two waves each)
  int nseg = waves >= 768 ? 1 : 1024 / waves;
  if (nseg > 16) nseg = 16;
  while (nseg > 1 && nchunks / nseg < 16) --nseg;
  return nseg < 1 ? 1 : nseg;
}
PairLayout pair_layout(int batch, int seqlen, int nheads, int ngroups) {
  PairLayout l;
  const size_t nchunks = (size_t)(seqlen + HQ - 1) / HQ;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  l.nseg = pair_segments(batch, nheads, (int)nchunks);
  l.seg_chunks = (int)((nchunks + l.nseg - 1) / l.nseg);
  l.seg_state = up((size_t)batch * ngroups * nchunks * CBE * sizeof(bf16_t));      // (the C.B^T pre-pass output comes first)
  const size_t st = (size_t)batch * nheads * HP * HN * sizeof(float);
  l.seg_decay = l.seg_state + (l.nseg > 1 ? up(l.nseg * st) : 0);
  l.ctot = l.seg_decay + (l.nseg > 1 ? up((size_t)l.nseg * batch * nheads * sizeof(float)) : 0);
  l.corr = l.ctot + (l.nseg > 1 ? up((size_t)batch * nheads * nchunks * sizeof(float)) : 0);
  l.dtt = l.corr + (l.nseg > 1 ? up(tv_ssd_correct_all_workspace_bytes(batch, nheads, (int)nchunks, l.nseg, HP)) : 0);
  l.total = l.dtt + up((size_t)batch * nheads * nchunks * HQ * sizeof(bf16_t));
  return l;
}

}  // namespace

bool tv_ssd_pair_supported(int seqlen, int nheads, int headdim, int ngroups, int dstate, int dtype,
                           int64_t xsl, int64_t bsl, int64_t bsg, int64_t csl, int64_t csg, int64_t ysl,
                           const void* x, const void* Bm, const void* Cm, const void* y) {
  if (dtype != TV_BF16 || dstate != HN || headdim != HP || seqlen < 1) return false;
  if (nheads % ngroups || (nheads / ngroups) % 4) return false;
  if (xsl % 8 || bsl % 8 || csl % 8 || bsg % 8 || csg % 8 || ysl % 8) return false;
  if (((uintptr_t)x & 15) || ((uintptr_t)Bm & 15) || ((uintptr_t)Cm & 15) || ((uintptr_t)y & 15)) return false;
  if (64 * xsl * 2 >= (1ll << 31) || 64 * bsl * 2 >= (1ll << 31) || 64 * csl * 2 >= (1ll << 31) ||
      64 * ysl * 2 >= (1ll << 31))
    return false;
  return true;
}

size_t tv_ssd_pair_workspace_bytes(int batch, int seqlen, int nheads, int ngroups) {
  return pair_layout(batch, seqlen, nheads, ngroups).total;
}

int tv_ssd_pair_launch(const void* x, const void* dt, const void* A, const void* Bm, const void* Cm,
                       const void* D, const void* dt_bias, const void* init_state, void* y,
                       void* final_state, void* total_decay, int batch, int seqlen, int nheads,
                       int ngroups, int64_t xsb, int64_t xsl, int64_t dsb, int64_t dsl, int64_t bsb,
                       int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg, int64_t ysb,
                       int64_t ysl, int dt_softplus, float dt_min, float dt_max, int group_map,
                       void* workspace, size_t workspace_bytes, const void* cb_pre, hipStream_t st) {
  const PairLayout lay = pair_layout(batch, seqlen, nheads, ngroups);
  TV_CHECK_ARG(workspace && workspace_bytes >= lay.total && (((uintptr_t)workspace) & 15) == 0,
               "ssd_pair: workspace of %zu bytes (16-byte aligned) required, got %zu", lay.total, workspace_bytes);
  unsigned char* wsb = (unsigned char*)workspace;
  PairArgs a;
  a.x = (const bf16_t*)x; a.Bm = (const bf16_t*)Bm; a.Cm = (const bf16_t*)Cm;
  a.cb = cb_pre ? (const bf16_t*)cb_pre : (const bf16_t*)workspace;
  a.A = (const float*)A; a.D = (const float*)D; a.dt_bias = (const float*)dt_bias;
  a.init = (const float*)init_state; a.y = (bf16_t*)y;
  a.final_state = (float*)final_state; a.total_decay = (float*)total_decay;
  a.L = seqlen; a.H = nheads; a.G = ngroups;
  a.nchunks = (seqlen + HQ - 1) / HQ;
  a.nseg = lay.nseg; a.seg_chunks = lay.seg_chunks;
  a.seg_state = lay.nseg > 1 ? (float*)(wsb + lay.seg_state) : nullptr;
  a.seg_decay = lay.nseg > 1 ? (float*)(wsb + lay.seg_decay) : nullptr;
  a.chunk_tot = lay.nseg > 1 ? (float*)(wsb + lay.ctot) : nullptr;
  // dt head-major (ssd_head.hip): both the march and the correction read it
  const int64_t lp = (int64_t)a.nchunks * HQ;
  bf16_t* dtt = (bf16_t*)(wsb + lay.dtt);
  tv_ssd_dt_transpose_launch(dt, dtt, batch, seqlen, nheads, dsb, dsl, st);
  a.dt = dtt;
  a.dsb = (int64_t)nheads * lp; a.dsl = 1; a.dsh = lp;
  a.xsb = xsb; a.xsl = xsl; a.bsb = bsb; a.bsl = bsl; a.bsg = bsg;
  a.csb = csb; a.csl = csl; a.csg = csg; a.ysb = ysb; a.ysl = ysl;
  a.softplus = dt_softplus; a.group_map = group_map; a.dt_min = dt_min; a.dt_max = dt_max;
  if (!cb_pre) {
    const int rc = tv_ssd_cb_prepass_launch(Bm, Cm, workspace, batch, seqlen, ngroups, bsb, bsl, bsg, csb, csl, csg, st);
    if (rc != TV_OK) return rc;
  }
  const int hpg = nheads / ngroups;
  const dim3 grid(ngroups * (hpg / 4), batch, a.nseg);
  hipError_t e = hipFuncSetAttribute((const void*)ssd_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PairSmem));
  if (e != hipSuccess) {
    tv_set_error("ssd_pair: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return TV_ERR_LAUNCH;
  }
  ssd_pair_kernel<<<grid, 512, sizeof(PairSmem), st>>>(a);
  if (a.nseg > 1) {
    const int rc = tv_ssd_correct_all_launch(y, dtt, A, Cm, dt_bias, a.seg_state, a.seg_decay, (float*)final_state,
                                             (float*)total_decay, a.chunk_tot, batch, seqlen, nheads, HP,
                                             ngroups, a.nseg, a.seg_chunks, ysb, ysl, a.dsb, a.dsl, a.dsh, csb, csl, csg,
                                             dt_softplus, dt_min, dt_max, group_map, wsb + lay.corr, st);
    if (rc != TV_OK) return rc;
  }
  TV_LAUNCH_CHECK();
}
