// S3 (fast path, v3, "impl 6"): Mamba-2 SSD selective scan as a HEAD-PER-WAVE march on CDNA4.
//
// Same single pass over HBM as ssd_slice.hip (x, dt, B, C read once, y written once, the running state
// never leaves the registers), but the unit of work is turned around.  The slice march gave one head to
// a whole work-group: three slice-waves did the MFMAs on three of the four SIMDs while five helper waves
// prepared operands, every role met every other at one barrier per chunk, and a 64-token step cost
// ~5 300 cycles for 190 MFMAs — 43 % of the wave-cycles parked (profiles/r02_ssd_scan_pmc.md).  Here
//
//   * a WAVE owns a whole head: X[128][P] lives in 8 x PT accumulator tiles (160 registers at head_dim
//     80) of a 512-register wave, one wave per SIMD, and that wave does everything for its head — dt ->
//     softplus -> DPP prefix sum, the decay mask, x~ = w_t x, Yoff = C.X, the state update, Ydiag, the
//     y stores — as ONE instruction stream the compiler can interleave (190 MFMAs beside ~900 vector
//     instructions per step); nothing is handed between waves except the B / C tiles;
//   * a work-group = the 4 (2, 1) heads of ONE B/C group that follow each other: B and C are copied to
//     LDS once per FOUR heads (LDS-DMA, each wave a quarter of the pieces) instead of once per head
//     (4x less L2 -> LDS traffic), every C / B^T fragment a wave reads serves all PT column tiles;
//   * 128 heads x 8 sequence segments = 1 024 waves = 4 per CU; segments > 0 start from a zero state
//     and are completed by the carried-in state correction (ssd_correct.hip), as in impl 4;
//   * the state is kept in a FLOATING FRAME: X = 2^E X', |E| <= RMAX.  A chunk's decay goes into the weights
//     (w_s = 2^(-cs_s - E) dt_s) and into ONE row factor 2^(cs_t + E) applied to the finished accumulators:
//         y_t = 2^(cs_t + E) ( C_t . X' + sum_{s <= t} CB[t][s] w_s x_s ) + D x_t,    X' += sum_s w_s B_s x_s
//     so no vector instruction touches the 160 state registers or the y accumulators between MFMAs, the
//     intra-chunk mask is the causal C.B^T itself — the same fragments for every head of the group, used
//     from global memory as they are (no per-head mask with its 48 exponentials a lane) — and x~ = w_s x_s
//     serves the state update and Ydiag alike.  Every factor formed lies within 2^(+-2 RMAX); when E + cs_Q
//     would leave [-RMAX, RMAX] the frame is re-based (X' *= 2^(E - RMAX), E = RMAX: one pass over the state,
//     every ~120 log2 units of decay), and a single chunk that decays by more than 2^(2 RMAX) (dt |A| > 1.3 per
//     token) runs round 2's standard step: true mask 2^(cs_t - cs_s) dt_s, X' *= 2^(E + cs_Q) mid-step, E = 0 —
//     only factors 2^(a - b), a >= b, like the reference's segment_sum form (modeling_nano.py:159-186);
//   * D x is added in fp32 from x in the accumulator layout.
// x reaches the MFMA operand layout through a wave-private LDS tile (register-staged: loaded one chunk
// ahead, written at the start of the step, read back transposed with ds_read_b64_tr_b16; rows padded to
// 192 bytes with the 32-byte pieces of rows 8..15 mod 16 swapped pairwise: conflict-free).
// Reference semantics: mamba_chunk_scan_combined call modeling_nano.py:639-653; arithmetic :775-851.
#include <stdlib.h>
#include <atomic>
#include <type_traits>
#include "ssd_common.hpp"

// ssd_slice.hip
int tv_ssd_cb_prepass_launch(const void* Bm, const void* Cm, void* cb, int batch, int seqlen, int ngroups,
                             int64_t bsb, int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg,
                             hipStream_t st);
// ssd_correct.hip
size_t tv_ssd_correct_all_workspace_bytes(int batch, int nheads, int nchunks, int nseg, int headdim);
unsigned* tv_ssd_correct_all_counters(void* workspace, int batch, int nheads, int nchunks, int nseg, int headdim);
int tv_ssd_correct_all_launch(void* y, const void* dt, const void* A, const void* Cm, const void* dt_bias,
                              const float* seg_state, const float* seg_decay, float* final_state,
                              float* total_decay, const float* chunk_tot, int batch, int seqlen, int nheads,
                              int headdim, int ngroups, int nseg, int seg_chunks, int64_t ysb, int64_t ysl,
                              int64_t dsb, int64_t dsl, int64_t dsh, int64_t csb, int64_t csl, int64_t csg, int dt_softplus,
                              float dt_min, float dt_max, int group_map, void* workspace, hipStream_t st);

#ifndef TV_HEAD_PIN
#define TV_HEAD_PIN 0
#endif
// TV_HEAD_Y16: 1 = y leaves in 16-byte stores (two t-tiles joined by v_permlane16_swap), 0 = 8-byte stores
#ifndef TV_HEAD_Y16
#define TV_HEAD_Y16 1
#endif
// TV_HEAD_RESET: 1 = a chunk that decays by more than 2^-RESET_THR starts the state anew (its carry-over, < 2^-RESET_THR of
// the old state, is dropped): the first k-step of the state update accumulates onto zero, the new state is built
// directly in the frame E = RMAX, and the re-basing pass such a head would need on every step never runs
#ifndef TV_HEAD_RESET
#define TV_HEAD_RESET 1
#endif
// Standard steps (a chunk that decays by more than 2^(2 RMAX)) are taken by the one kernel itself, as three side blocks of
// the one step body (scaled accumulators, per-head mask built into the C.B^T registers, raw x fragments).  Round 3's first
// scheme — a flag per work-group + a second, complete kernel, or a device-side check that sent such calls to the slice
// march — was measured (7.5 ms / 2.92 ms against 2.50 ms at 164 k tokens) and removed in round 4.
// TV_HEAD_UNTRACKED: 1 = the C.B^T / dt loads of the 4-wave fast kernels are inline asm with counted waits (needs a
// kernel without scratch), 0 = ordinary loads
#ifndef TV_HEAD_UNTRACKED
#define TV_HEAD_UNTRACKED 1
#endif
// TV_HEAD_XW_IN_Q0: 1 = x~ is formed tile by tile beside the Yoff MFMAs of quarter 0, 0 = all of it before the quarters
#ifndef TV_HEAD_XW_IN_Q0
#define TV_HEAD_XW_IN_Q0 1
#endif
// TV_HEAD_CBJ: the MFMA group of quarter 2 behind which the C.B^T loads are issued (3: measured best by ~1 %; 5 = behind the last x copies)
#ifndef TV_HEAD_CBJ
#define TV_HEAD_CBJ 3
#endif
// TV_HEAD_FENCE: 1 = a scheduling fence behind every group of PT MFMAs (pins the memory operations between them),
// 0 = one fence per quarter
#ifndef TV_HEAD_FENCE
#define TV_HEAD_FENCE 1
#endif
namespace {
using namespace ssdk;

constexpr int HQ = 64;            // tokens per chunk
constexpr int HN = 128;           // d_state
constexpr int NFR = 6;            // causal (t-tile, s-pair) fragments of a 64x64 chunk: (0,0) (1,0) (2,0) (2,1) (3,0) (3,1)
constexpr int CBE = NFR * 512;    // bf16 elements of C.B^T per (chunk, group)
constexpr float RMAX = 100.f;     // the floating frame stays within 2^(+-RMAX)
constexpr float RESET_THR = 64.f; // log2 decay of a chunk from which on the carried state is dropped (TV_HEAD_RESET)

struct HeadArgs {
  const bf16_t *x, *dt, *Bm, *Cm, *cb;
  const float *A, *D, *dt_bias, *init;
  bf16_t* y;
  float *final_state, *total_decay;
  float *seg_state, *seg_decay, *chunk_tot;     // nseg > 1: per-segment results for the combine / correction pass
  int nseg, seg_chunks;
  int L, H, P, G, nchunks;
  int64_t xsb, xsl, dsb, dsl, dsh, bsb, bsl, bsg, csb, csl, csg, ysb, ysl;      // dsh: elements between the heads of dt
  int softplus, group_map;
  float dt_min, dt_max;
  int dbg;
};

// -DTV_HEAD_ABLATE: timing ablations selected by env TV_HEAD_DBG (results are wrong): 1 no quarters, 2 no x~,
// 4 no y stores, 8 no C.B^T loads / Ydiag, 16 no prep, 32 no x copies after the first chunk, 64 no B / C copies after the
// first chunk, 128 no dt loads
#ifdef TV_HEAD_ABLATE
#define HDBG(a, bit) ((a).dbg & (bit))
#else
#define HDBG(a, bit) 0
#endif

// -DTV_HEAD_STAMP: wave 0 of work-group 0 sums the cycles (s_memtime) of eight phases of its steps;
// tv_ssd_head_debug_stamps() returns them
#ifdef TV_HEAD_STAMP
__device__ unsigned long long g_head_phases[64];
#define HSTAMP(slot) do { unsigned long long n__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n__) :: "memory"); ph_acc[slot] += n__ - ph_last; ph_last = n__; } while (0)
#else
#define HSTAMP(slot) do {} while (0)
#endif

struct __attribute__((aligned(16))) HeadVec {   // per-chunk vectors of one head (lane = token when written)
  float cs[HQ];       // inclusive cumsum of dt A inside the chunk, times log2(e)
  float dtv[HQ];      // discretised dt
  float ut[HQ];       // 2^(cs_t - cs_{16 (t/16)}): row factor of the separable off-diagonal mask blocks
  float wts[HQ];      // weight of token s in the state update (frame-dependent)
  float wtd[2][HQ];   // (by chunk parity: prep of the next chunk runs before this chunk's Ydiag) reset steps: weight of token s in Ydiag (the floating frame's; wts then builds the new state at E = RMAX)
  float ecs[HQ];      // 2^(cs_t + E): row factor of Yoff
  float ws[128];      // column factors of the separable blocks: t-tile 1 at [0,16), 2 at [16,48), 3 at [48,96)
};

template <int PT, int NW, int NB>
struct __attribute__((aligned(16))) HeadSmem {
  bf16_t bt[NB][HQ * HN];     // B tiles [t][n], 16-byte chunk index ^ 4(t & 3) (ds_read_b64_tr)
  bf16_t ct[NB][HQ * HN];     // C tiles [t][n], chunks XOR-swizzled for row reads
  bf16_t xr[NW][2][HQ * PT * 16 + 256];   // x tiles [t][P] of each wave's head, ring of 2 (wave-private) + the copy's overhang
  HeadVec v[NW];
};

__device__ __forceinline__ unsigned lds_lane_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ int xad(int a, int k, int b) {     // (a ^ k) + b, k wave-uniform
  int d;
  asm("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(b));
  return d;
}
__device__ __forceinline__ float rdlane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

#define HEAD_BARRIER(N)                                   \
  do {                                                    \
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); \
    __builtin_amdgcn_s_barrier();                         \
  } while (0)

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int PT, int NW, int NB>
__global__ __launch_bounds__(NW * 64) void ssd_head_kernel(HeadArgs a) {
  typedef HeadSmem<PT, NW, NB> Smem;
  // Loads the compiler does not track (inline asm + counted waits) are only safe in a kernel that spills NOTHING: a
  // register with such a load in flight may otherwise be saved before its data has arrived (seen: a spilling variant
  // read stale C.B^T fragments that way).  Only the 4-wave kernels are held to zero scratch
  // (tests/test_build_cpu.py reads it off the code object); the 2- and 1-wave variants use ordinary loads and full waits.
  constexpr bool UNTRACKED = TV_HEAD_UNTRACKED && NW == 4;
  constexpr int BD = NB - 1;            // B/C prefetch distance (chunks)
  static_assert(BD == 1, "the waits below are counted for a B/C ring of 2");
  constexpr int P = PT * 16;
  constexpr int KP = 16 / NW;           // 1 KiB pieces (4 token rows) of B per wave and chunk; as many of C
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, kq = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int b = blockIdx.y;
  const int hpg = a.H / a.G;
  const int g = blockIdx.x % a.G;
  const int hig = (blockIdx.x / a.G) * NW + wave;
  const int h = a.group_map ? (hig * a.G + g) : (g * hpg + hig);
  const int seg = blockIdx.z;
  const int c_first = seg * a.seg_chunks;
  const int t_first = c_first * HQ;
  const int nchunks = min(a.seg_chunks, a.nchunks - c_first);
  const int L = min(a.L - t_first, nchunks * HQ);
  HeadVec& vec = sm.v[wave];
  // LDS byte addresses are formed from ONE cast of the array base (every generic -> LDS cast carries a null check; a
  // dozen of them under scalar-register pressure made the backend emit an illegal VALU compare with src_shared_base)
  const unsigned lds0 = lds_addr_of(smem_raw);
  const unsigned lds_bt = lds0 + (unsigned)offsetof(Smem, bt), lds_ct = lds0 + (unsigned)offsetof(Smem, ct);
  const unsigned lds_xr = lds0 + (unsigned)offsetof(Smem, xr) + wave * (unsigned)sizeof(sm.xr[0]);       // this wave's two tiles
  constexpr unsigned XSLOT = sizeof(sm.xr[0][0]);

  {   // the vectors' never-written tails must hold finite values
    float* vz = reinterpret_cast<float*>(&vec);
    for (int i = lane; i < (int)(sizeof(HeadVec) / 4); i += 64) vz[i] = 0.f;
  }

  // ------------------------------------------------------------------ B / C copies (LDS-DMA)
  const bf16_t* Bg = a.Bm + (int64_t)b * a.bsb + (int64_t)g * a.bsg + (int64_t)t_first * a.bsl;
  const bf16_t* Cg = a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg + (int64_t)t_first * a.csl;
  // piece k of this wave = token rows 4 KP wave + 4 k + (lane >> 4), 16-byte chunk lane & 15 of the row, stored
  // swizzled: B chunk ^ 4 (row & 3) (the same for every k), C chunk ^ (row & 15) = (chunk ^ (lane >> 4)) ^ 4 (k & 3)
  const int bc_row0 = 4 * KP * wave + (lane >> 4);
  const unsigned off_b0 = (unsigned)((bc_row0 * a.bsl + ((lane & 15) ^ (4 * ((lane >> 4) & 3))) * 8) * 2);
  const unsigned off_c0 = (unsigned)(bc_row0 * a.csl * 2);
  const unsigned cgc0 = (unsigned)(((lane & 15) ^ (lane >> 4)) << 4);           // byte offset of the C chunk for k = 0 (4 KP wave is a multiple of 16)
  static_assert(KP % 4 == 0, "B/C pieces");
  // which = 0: the B pieces of chunk c, 1: the C pieces (KP / 4 groups of four pieces with one M0 set-up each)
  auto issue_bc = [&](int c, int which) {
    const int slot = c % NB;
    const int t0 = c * HQ;
    const bf16_t* Tc = which ? Cg + (int64_t)t0 * a.csl : Bg + (int64_t)t0 * a.bsl;
    const int64_t rl = which ? a.csl : a.bsl;
    const unsigned dst = (which ? lds_ct : lds_bt) + slot * (HQ * HN * 2);     // LDS byte address of the tile
    if (t0 + HQ <= L) {       // the scalar base moves by 16 rows per group of four
#pragma unroll
      for (int k = 0; k < KP; k += 4) {
        const void* sp = uniform_ptr(Tc + (int64_t)4 * k * rl);
        const unsigned r4 = (unsigned)(4 * rl * 2);      // bytes per 4 rows
        if (which == 0)
          glds16x4(sp, off_b0, off_b0 + r4 - 1024u, off_b0 + 2 * r4 - 2048u, off_b0 + 3 * r4 - 3072u,
                   dst + (KP * wave + k) * 1024);
        else
          glds16x4(sp, off_c0 + cgc0, off_c0 + r4 + (cgc0 ^ 64u) - 1024u, off_c0 + 2 * r4 + (cgc0 ^ 128u) - 2048u,
                   off_c0 + 3 * r4 + (cgc0 ^ 192u) - 3072u, dst + (KP * wave + k) * 1024);
      }
      return;
    }
    const void* sp = uniform_ptr(Tc);
#pragma unroll
    for (int k = 0; k < KP; ++k) {      // last, partial chunk: rows past the end repeat the last row (finite)
      const int row = 4 * KP * wave + 4 * k + (lane >> 4);
      const int rr = min(row, L - 1 - t0);
      const int cg = which ? (lane & 15) ^ (row & 15) : (lane & 15) ^ (4 * (row & 3));
      glds16(sp, (unsigned)((rr * rl + cg * 8) * 2), dst + (KP * wave + k) * 1024);
    }
  };

  // ------------------------------------------------------------------ x: LDS-DMA into the wave's own ring of 2 tiles
  constexpr int XROW = 2 * P;                               // bytes per row
  constexpr int NPC = P / 8;                                // 16-byte pieces per row
  const bf16_t* xg = a.x + (int64_t)b * a.xsb + (int64_t)t_first * a.xsl + (int64_t)h * P;
  // A copy instruction moves 64 consecutive 16-byte pieces of the row-major tile = RPI whole rows + the first pieces
  // of the next one (which the following instruction writes again with the same bytes): lane -> (row lane / NPC, piece
  // lane % NPC) for every instruction, the scalar base moves by RPI rows.  The last instruction's lanes past row 63 re-read
  // row 63 and land in the overhang behind the tile.
  constexpr int RPI = 64 / NPC;                             // whole rows per instruction (6 at head_dim 80)
  constexpr int NXI = (HQ + RPI - 1) / RPI;                 // instructions per tile (11)
  const int x_lrow = lane / NPC;
  const unsigned x_off = (unsigned)((x_lrow * a.xsl + (lane % NPC) * 8) * 2);
  const unsigned x_off_last = (unsigned)((min(x_lrow, HQ - 1 - RPI * (NXI - 1)) * a.xsl + (lane % NPC) * 8) * 2);
  // group g of chunk c: the copy instructions 4 g .. 4 g + 3 (< NXI) with ONE M0 set-up and one scalar base; the
  // instruction offset moves the global and the LDS address alike, so instruction j's lane offset is corrected by
  // j (bytes of RPI rows in memory - bytes of RPI rows in the tile)
  constexpr int NXG = (NXI + 3) / 4;
  auto issue_x = [&](int c, int g) {
    const int t0 = c * HQ;
    const bf16_t* xc = xg + (int64_t)t0 * a.xsl;
    const unsigned dst = lds_xr + (c & 1) * XSLOT + RPI * 4 * g * XROW;
    if (t0 + HQ > L) {        // last, partial chunk: rows clamped per lane
#pragma unroll
      for (int k = 4 * g; k < 4 * g + 4 && k < NXI; ++k)
        glds16(uniform_ptr(xc), (unsigned)((min(RPI * k + x_lrow, L - 1 - t0) * a.xsl + (lane % NPC) * 8) * 2),
               lds_xr + (c & 1) * XSLOT + RPI * k * XROW);
      return;
    }
    const void* sx = uniform_ptr(xc + (int64_t)RPI * 4 * g * a.xsl);
    const unsigned dj = (unsigned)(RPI * a.xsl * 2) - (unsigned)(RPI * XROW);
    const int n = NXI - 4 * g < 4 ? NXI - 4 * g : 4;
    const unsigned vl = (4 * g + n == NXI) ? x_off_last : x_off;      // the tile's last instruction clamps its rows
    unsigned keep;
    if (n == 4)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %5\n\t"
                   "global_load_lds_dwordx4 %2, %5 offset:%7\n\t"
                   "global_load_lds_dwordx4 %3, %5 offset:%8\n\t"
                   "global_load_lds_dwordx4 %4, %5 offset:%9\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(x_off), "v"(x_off + dj), "v"(x_off + 2 * dj), "v"(vl + 3 * dj), "s"(sx), "s"(dst),
                     "n"(RPI * XROW), "n"(2 * RPI * XROW), "n"(3 * RPI * XROW) : "memory");
    else if (n == 3)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %4\n\t"
                   "global_load_lds_dwordx4 %2, %4 offset:%6\n\t"
                   "global_load_lds_dwordx4 %3, %4 offset:%7\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(x_off), "v"(x_off + dj), "v"(vl + 2 * dj), "s"(sx), "s"(dst),
                     "n"(RPI * XROW), "n"(2 * RPI * XROW) : "memory");
    else if (n == 2)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %3\n\t"
                   "global_load_lds_dwordx4 %2, %3 offset:%5\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(x_off), "v"(vl + dj), "s"(sx), "s"(dst), "n"(RPI * XROW) : "memory");
    else
      glds16(sx, vl, dst);
  };
  // transposing reads (MFMA operand with k = token): lane (lc = column, kq) gets tokens 32 ks + 8 kq + 0..7 of column
  // 16 ct + lc as two ds_read_b64_tr_b16 (rows 8 kq + q4 and + 4, 8 bytes at column piece 4 ct + p4)
  const int xr_lo = (8 * kq + q4) * XROW + 8 * p4;
  auto read_xf = [&](unsigned xt, int ct, int ks) {          // xt: LDS byte address of the tile
    const unsigned p = xt + xr_lo + 32 * ct + ks * (32 * XROW);
    return cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(size_t)p), __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(size_t)(p + 4 * XROW)));
  };
  // y: a lane of the (transposed) accumulator tile (ct, ti) holds the four columns 16 ct + 4 kq + r of token 16 ti + lc:
  // 8 bytes, stored straight from the registers at the end of the step (nobody waits for these stores: the step's last
  // wait leaves them in flight).  x is read back in the same layout for the D x term.
  const int xv_lo = lc * XROW + 8 * kq;
  bf16_t* const ygs = a.y + (int64_t)b * a.ysb + (int64_t)t_first * a.ysl + (int64_t)h * P;
  const unsigned yoff0 = (unsigned)((lc * a.ysl + 4 * kq) * 2);
  unsigned yoff16 = (unsigned)(((16 * (kq & 1) + lc) * a.ysl + 8 * (kq >> 1)) * 2);     // 16-byte form (TV_HEAD_Y16)
  auto store_y_tile = [&](const void* yrow, bool ok, int ct, u32x2 v) {      // yrow: row 16 ti of the chunk, columns of this head
    if (ok) asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" :: "v"(yoff0), "v"(v), "s"(yrow), "n"(32 * ct) : "memory");
  };

  // ------------------------------------------------------------------ per-chunk vectors (lane = token)
  const float Ah = a.A[h];
  const float bias = a.dt_bias ? a.dt_bias[h] : 0.f;
  const float Dh = a.D ? a.D[h] : 0.f;
  const bf16_t* dtg = a.dt + (int64_t)b * a.dsb + (int64_t)t_first * a.dsl + (int64_t)h * a.dsh;
  float decay_total = 0.f;
  float E = 0.f;                  // X = 2^E X'
  // raw dt of chunk c, lane = token (rows clamped): a load the compiler does not track (the step's counted waits cover it)
  auto load_dt = [&](int c) {
    const int t = min(c * HQ + lane, L - 1);
    const bf16_t* p = dtg + (int64_t)t * a.dsl;
    unsigned r;
    if (HDBG(a, 128)) return 0u;
    if (!UNTRACKED) r = *(const unsigned short*)p;
    else asm volatile("global_load_ushort %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
  };
  // Decides how chunk c is marched and leaves its vectors in `vec`.  Returns the mode; f_out = the factor of the state:
  //   0  floating step in the current frame (f = 1);
  //   1  floating step after the frame has been re-based to E ~ +RMAX: the caller multiplies the state by 2^f (f = -m, an
  //      integer) first;
  //   2  standard step (one chunk decays by more than 2^(2 RMAX)): the old state is dropped (always a reset step), true mask;
  //      reported to the caller as 0 (nothing to do to the state), `std_next` carries it.
  bool reset_next = false, reset_cur = false, std_next = false, std_cur = false;     // (wave-uniform)
  auto prep = [&](int c, unsigned raw_bits, float& f_out) __attribute__((always_inline)) {
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
    // mode 1: X' *= 2^-m with the integer m = floor(RMAX - E) (v_ldexp_f32: exact, no underflow on the way), E += m
    const float mshift = mode == 1 ? __builtin_floorf(RMAX - E) : 0.f;
    const float Euse = E + mshift;                           // frame in which this chunk reads the state
    f_out = mode == 1 ? -mshift : 1.f;         // (mode 2 drops the old state: nothing to scale)
    vec.cs[lane] = cs2;
    vec.dtv[lane] = d;
    vec.ecs[lane] = __builtin_amdgcn_exp2f(cs2 + Euse);
    const bool rst = TV_HEAD_RESET && mode != 2 && cl2 <= -RESET_THR;
    const bool ustd = mode == 2;                                // standard step: always a reset step
    reset_next = rst || ustd;
    std_next = ustd;
    vec.wts[lane] = __builtin_amdgcn_exp2f(mode == 2 ? cl2 - cs2 : rst ? cl2 - cs2 - RMAX : -cs2 - Euse) * d;
    if (rst) vec.wtd[c & 1][lane] = __builtin_amdgcn_exp2f(-cs2 - Euse) * d;
    if (a.chunk_tot && lane == 0) a.chunk_tot[((int64_t)b * a.H + h) * a.nchunks + c_first + c] = cl2;
    if (__builtin_expect(mode == 2, 0)) {
      // separable factors of the off-diagonal mask blocks (pivot = first token of a t-tile): standard steps only
      const float p0 = rdlane(cs2, 0), p1 = rdlane(cs2, 16), p2 = rdlane(cs2, 32), p3 = rdlane(cs2, 48);
      const float pv = lane < 16 ? p0 : lane < 32 ? p1 : lane < 48 ? p2 : p3;
      vec.ut[lane] = __builtin_amdgcn_exp2f(fminf(cs2 - pv, 0.f));
      if (lane < 16) vec.ws[lane] = __builtin_amdgcn_exp2f(fminf(p1 - cs2, 0.f)) * d;
      if (lane < 32) vec.ws[16 + lane] = __builtin_amdgcn_exp2f(fminf(p2 - cs2, 0.f)) * d;
      if (lane < 48) vec.ws[48 + lane] = __builtin_amdgcn_exp2f(fminf(p3 - cs2, 0.f)) * d;
    }
    decay_total += cl;
    E = mode == 2 ? 0.f : rst ? RMAX : Euse + cl2;
    return __builtin_amdgcn_readfirstlane(ustd ? 0 : mode);
  };

  // ------------------------------------------------------------------ state, fragment addresses
  // The state lives in accumulation registers for the whole march.  Apart from the MFMAs every access goes through the
  // two helpers below with "a" constraints, so the register allocator never sees a vector-ALU use of it (it then keeps
  // vector-register copies of 150 state values alive across the step, or shuttles the state between the two files at
  // the loop boundaries: 300 - 700 moves and 200 - 450 spilled registers per step in the first builds).
  f32x4 xacc[PT][8];
#pragma unroll
  for (int ct = 0; ct < PT; ++ct) {
#pragma unroll
    for (int i = 0; i < 8; ++i) xacc[ct][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.init && seg == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        xacc[ct][i] = *(const f32x4*)(a.init + (((int64_t)b * a.H + h) * P + 16 * ct + lc) * HN + 32 * (i >> 1) + 8 * kq + 4 * (i & 1));
    }
  }
  // bf16 copy of the state rows n = 32 q + 8 kq + 0..7 of column tile ct (state tiles 2 q, 2 q + 1): the operand of Yoff.
  // The copies out of the accumulation registers are made HERE, eight at a time.  The tiles read were last written by
  // MFMAs a quarter-step or more ago (no MFMA -> v_accvgpr_read hazard in reach).
  auto snap_tile = [&](int q, int ct) {
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
  // X' *= 2^sh, sh a (negative) integer: re-basing of the frame.  v_ldexp_f32 is exact and cannot underflow on the way;
  // out of and back into the accumulation registers by hand, like every other non-MFMA access of the state.
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
  // State tiles 2m / 2m+1 hold the state rows n = 32m + 8kq + r / + 4 + r (r = accumulator register), so the pair is,
  // as an MFMA operand, the k slots n = 32m + 8kq + 0..7 — the order of a plain 16-byte row read of C.
  const int c_lo = lc * 256;
  const int c_z = (kq ^ lc) << 4;
  const int bsw = q4 << 6;
  const int b_lo = (8 * kq + q4) * 256 + p4 * 16;
  auto read_cq = [&](const unsigned char* Ct, int q, bf16x8 (&cf)[4]) {     // C[t = 16 ti + lc][n = 32 q + 8 kq + 0..7]
    const unsigned char* cp = Ct + xad(c_z, 64 * q, c_lo);
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) cf[ti] = ld8(cp + ti * 4096);
  };
  auto read_b2 = [&](const unsigned char* Bt, int i0, bf16x4 (&dst)[2][4]) {   // B^T for state tiles i0 = 2m, i0 + 1
    const unsigned char* bp = Bt + xad(bsw, 32 * i0, b_lo);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        dst[ii][2 * ks] = tr4(bp + ii * 8 + ks * 8192);
        dst[ii][2 * ks + 1] = tr4(bp + ii * 8 + ks * 8192 + 1024);
      }
  };
  const bf16_t* cbg = a.cb + (((int64_t)b * a.G + g) * a.nchunks + c_first) * CBE + lane * 8;

#ifdef TV_HEAD_STAMP
  unsigned long long ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_last = clock64();
#endif
  // ------------------------------------------------------------------ prologue
  issue_bc(0, 0);
  issue_bc(0, 1);
#pragma unroll
  for (int g = 0; g < NXG; ++g) issue_x(0, g);
  unsigned dt_next = load_dt(0);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(dt_next) :: "memory");
  float f_step;
  int mode = prep(0, dt_next, f_step);
  reset_cur = __builtin_amdgcn_readfirstlane((int)reset_next) != 0;
  std_cur = __builtin_amdgcn_readfirstlane((int)std_next) != 0;
  if (mode == 1) rebase_state((int)f_step);
  dt_next = load_dt(min(1, nchunks - 1));
  HEAD_BARRIER(0);

  // One 64-token step.  Floating frame, the common case — the chunk's decay sits in the weights
  // (w_s = 2^(-cs_s - E) dt_s) and in ONE row factor 2^(cs_t + E) applied at the very end, so Yoff and Ydiag accumulate
  // into the same tiles with no vector instruction in between, the mask is the causal C.B^T itself (the same for every
  // head of the group, taken from global memory as it is) and x~ serves the state update and Ydiag alike:
  //     y_t = 2^(cs_t + E) ( C_t . X' + sum_{s <= t} CB[t][s] x~_s ) + D x_t
  // A chunk that decays by more than 2^(2 RMAX) takes the true-mask form through three side blocks (`ustd_step`).
  // The step's memory operations are issued in small groups BETWEEN the MFMAs of the first three quarters (issued in a
  // burst they stalled the wave for ~5 000 cycles per step at the full queue of the copy path, the y stores for
  // another ~4 900): quarter 0: the y rows of the previous chunk, from the tile the x copies of the next chunk then
  // reuse; quarter 1: C.B^T of this chunk, x of the next; quarter 2: B / C of the next chunk, dt of the one after.
  auto step = [&](int c, bool more) __attribute__((always_inline)) {
    const bool reset_step = reset_cur;       // this chunk builds its state anew (its own value: prep below sets the next chunk's)
    const bool ustd_step = std_cur;          // ... and is a standard step (side blocks below)
    const unsigned char* Bt = reinterpret_cast<const unsigned char*>(sm.bt[c % NB]);
    const unsigned char* Ct = reinterpret_cast<const unsigned char*>(sm.ct[c % NB]);
    const unsigned xt = lds_xr + (c & 1) * XSLOT;
    // ---- x~ = w_s x on the fragments (element j of fragment ks is token 32 ks + 8 kq + j)
    bf16x8 xw[PT][2];
    f32x4 wq[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      wq[ks][0] = *(const f32x4*)(&vec.wts[32 * ks + 8 * kq]);
      wq[ks][1] = *(const f32x4*)(&vec.wts[32 * ks + 8 * kq + 4]);
    }
    auto make_xw = [&](int ct) {
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
#if !TV_HEAD_XW_IN_Q0
#pragma unroll
    for (int ct = 0; ct < PT; ++ct) make_xw(ct);
#endif
    HSTAMP(1);
    // ---- Yoff^T = X'^T C^T and X' (= f X') += B^T x~, in quarters of 32 state rows; each quarter in 8 groups of PT MFMAs
    // with the bf16 copy of the next quarter's state rows and a few memory operations behind each group
    f32x4 yo[PT][4];
#pragma unroll
    for (int ct = 0; ct < PT; ++ct)
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) yo[ct][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 cq[2][4];
    bf16x4 bq[2][2][4];
    bf16x8 sbq[2][PT];
    bf16x8 cbv[NFR];                // causal C.B^T of this chunk (L2 / L1: every head of the group reads the same 6 KiB)
    auto quarter = [&](int q, const bf16x8 (&cf)[4], const bf16x4 (&bt2)[2][4], const bf16x8 (&sb)[PT], bf16x8 (&sbn)[PT],
                       auto filler, bool fence = true) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < 4) {
#pragma unroll
          for (int ct = 0; ct < PT; ++ct) yo[ct][j] = mfma16(sb[ct], cf[j], yo[ct][j]);
        } else {
          const int ks = (j - 4) >> 1, ii = (j - 4) & 1;
          const bf16x8 bfrag = cat4(bt2[ii][2 * ks], bt2[ii][2 * ks + 1]);
          if (TV_HEAD_RESET && ks == 0 && reset_step) {
#pragma unroll
            for (int ct = 0; ct < PT; ++ct) xacc[ct][2 * q + ii] = mfma16(bfrag, xw[ct][0], f32x4{0.f, 0.f, 0.f, 0.f});
          } else {
#pragma unroll
            for (int ct = 0; ct < PT; ++ct) xacc[ct][2 * q + ii] = mfma16(bfrag, xw[ct][ks], xacc[ct][2 * q + ii]);
          }
        }
        if (q < 3 && j < PT) sbn[j] = snap_tile(q + 1, j);
#if TV_HEAD_XW_IN_Q0
        // x~ of column tile j beside the Yoff MFMAs of the first quarter (they do not need it; the state update, groups 4 - 7, does)
        if (q == 0 && j < 4) {
          if (j < PT) make_xw(j);
          if (j == 3) {
#pragma unroll
            for (int ct = 4; ct < PT; ++ct) make_xw(ct);
          }
        }
#endif
        filler(j);
        if (fence && TV_HEAD_FENCE) __builtin_amdgcn_sched_barrier(0);
      }
      if (fence && !TV_HEAD_FENCE) __builtin_amdgcn_sched_barrier(0);
    };
    read_cq(Ct, 0, cq[0]);
    read_b2(Bt, 0, bq[0]);
#pragma unroll
    for (int ct = 0; ct < PT; ++ct) sbq[0][ct] = snap_tile(0, ct);
    __builtin_amdgcn_sched_barrier(0);
    read_cq(Ct, 1, cq[1]);
    read_b2(Bt, 2, bq[1]);
    quarter(0, cq[0], bq[0], sbq[0], sbq[1], [&](int j) {           // B of the next chunk (late: the y stores of the last step drain first)
      if (more && j == 4 && !HDBG(a, 64)) issue_bc(c + 1, 0);
    });
    HSTAMP(2);
    read_cq(Ct, 2, cq[0]);
    read_b2(Bt, 4, bq[0]);
    unsigned dt_raw = dt_next;                // dt of chunk c + 1 (loaded a step ago), for prep below
    quarter(1, cq[1], bq[1], sbq[1], sbq[0], [&](int j) {           // C of the next chunk, the first x rows
      if (more && j == 0 && !HDBG(a, 64)) issue_bc(c + 1, 1);
      if (more && j == 4 && NXG > 0 && !HDBG(a, 32)) issue_x(c + 1, 0);
    });
    HSTAMP(3);
    read_cq(Ct, 3, cq[1]);
    read_b2(Bt, 6, bq[1]);
    quarter(2, cq[0], bq[0], sbq[0], sbq[1], [&](int j) {           // the rest of x, dt of the chunk after the next
      if (more && j == 0 && NXG > 1 && !HDBG(a, 32)) issue_x(c + 1, 1);
      if (j == TV_HEAD_CBJ && !HDBG(a, 8)) {       // C.B^T of this chunk (behind the last x copies: only dt is issued after it)
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
          u32x4v r;
          if (!UNTRACKED) r = *(const u32x4v*)(cbg + (int64_t)c * CBE + f * 512);
          else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(cbg + (int64_t)c * CBE + f * 512) : "memory");
          cbv[f] = __builtin_bit_cast(bf16x8, r);
        }
      }
      if (more && j == 4 && NXG > 2 && !HDBG(a, 32)) issue_x(c + 1, 2);
      if (j == 6) dt_next = load_dt(min(c + 2, nchunks - 1));
    });
    HSTAMP(4);
    float ev[4];                    // 2^(cs_t + E) of this lane's token 16 ti + lc
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) ev[ti] = vec.ecs[16 * ti + lc];
    // floating steps: the vectors of the NEXT chunk are prepared here, in one scheduling region with the MFMAs of the last
    // quarter (everything that reads this chunk's vectors has been read; the re-basing of the frame waits for the quarter)
    int mode_next = 0;
    if (!ustd_step && c + 1 < nchunks) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      mode_next = prep(c + 1, dt_raw, f_step);
    }
    quarter(3, cq[1], bq[1], sbq[1], sbq[0], [&](int) {}, false);
    HSTAMP(5);
    if (TV_HEAD_RESET && reset_step && !ustd_step) {
      // the state update ran on x~ in the NEW frame; Ydiag shares the accumulators with Yoff and needs the old frame's
      asm volatile("; reset step: x~ for Ydiag" ::);
      f32x4 wq[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        wq[ks][0] = *(const f32x4*)(&vec.wtd[c & 1][32 * ks + 8 * kq]);
        wq[ks][1] = *(const f32x4*)(&vec.wtd[c & 1][32 * ks + 8 * kq + 4]);
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
    }
    // C.B^T has landed; what was issued behind it (the last x copies, dt) may stay in flight
    if (UNTRACKED && more && (c + 2) * HQ <= L) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TV_HEAD_CBJ >= 4 ? 1 : (NXG > 2 ? NXI - 8 : 0) + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int f = 0; f < NFR; ++f) asm volatile("" : "+v"(cbv[f]));
    HSTAMP(6);
    if (ustd_step) {
      // ---- standard step inside the one step body (a chunk that decays by more than 2^-199: no single frame holds its
      // weights).  Three side blocks turn the common Ydiag below into the true-mask form:
      //   (1) the accumulators get their row factor NOW (2^(cs_t + E); tiles that the old state no longer reaches
      //       become zeros), by hand like every vector access of accumulation registers in this kernel;
      //   (2) the per-head mask M = CB .* 2^(cs_t - cs_s) dt_s [s <= t] replaces C.B^T in its registers: diagonal
      //       16x16 blocks one exponential per element (by lane half: fragments (0,0) / (1,0) and (2,1) / (3,1)), the
      //       other blocks separable around the first token of their t-tile, ut[t] ws[s];
      //   (3) the A operand becomes the raw x fragments.
      // The state update of such a chunk ran with the standard weights 2^(cs_Q - cs_s) dt_s onto zero (reset: E = 0).
      asm volatile("; standard step" ::);
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
        auto diag = [&](int t, int s0, float (&e)[8]) {       // e[j] = 2^(cs_t - cs_(s0 + j)) dt_(s0 + j) for s0 + j <= t, else 0
          const float cst = vec.cs[t];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const f32x4 cv = *(const f32x4*)(&vec.cs[s0 + 4 * hh]), dv = *(const f32x4*)(&vec.dtv[s0 + 4 * hh]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              e[4 * hh + j] = __builtin_amdgcn_exp2f(s0 + 4 * hh + j <= t ? cst - cv[j] : -__builtin_inff()) * dv[j];
          }
        };
        auto sepf = [&](int t, int wofs, float (&e)[8]) {      // e[j] = ut[t] ws[wofs + j]
          const float u = vec.ut[t];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const f32x4 wv = *(const f32x4*)(&vec.ws[wofs + 4 * hh]);
#pragma unroll
            for (int j = 0; j < 4; ++j) e[4 * hh + j] = u * wv[j];
          }
        };
        auto apply = [&](int f, const float (&fac)[8]) {       // cbv[f] <- bf16(CB_f .* fac)
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
        float eD[8], eS[8], fac[8];
        const int sA = 8 * kq;
        diag(16 * hi + lc, sA, eD);                           // diagonal blocks of fragments (0,0) [hi = 0] and (1,0) [hi = 1]
#pragma unroll
        for (int j = 0; j < 8; ++j) fac[j] = hi ? 0.f : eD[j];
        apply(0, fac);
        sepf(16 + lc, sA & 15, eS);                           // block (1,0)
#pragma unroll
        for (int j = 0; j < 8; ++j) fac[j] = hi ? eD[j] : eS[j];
        apply(1, fac);
        sepf(32 + lc, 16 + sA, eS);                           // blocks (2,0), (2,1)
        apply(2, eS);
        diag(32 + 16 * hi + lc, 32 + sA, eD);                 // diagonal blocks of fragments (2,1) [hi = 0] and (3,1) [hi = 1]
#pragma unroll
        for (int j = 0; j < 8; ++j) fac[j] = hi ? 0.f : eD[j];
        apply(3, fac);
        sepf(48 + lc, 48 + sA, eS);                           // blocks (3,0), (3,1)
        apply(4, eS);
        sepf(48 + lc, 48 + 32 + (sA & 15), eS);               // block (3,2)
#pragma unroll
        for (int j = 0; j < 8; ++j) fac[j] = hi ? eD[j] : eS[j];
        apply(5, fac);
      }
#pragma unroll
      for (int ct = 0; ct < PT; ++ct)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xw[ct][ks] = read_xf(xt, ct, ks);
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) ev[ti] = 1.f;
      // the vectors of the next chunk (floating steps prepared them inside quarter 3; this step still needed its own)
      if (c + 1 < nchunks) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mode_next = prep(c + 1, dt_raw, f_step);
      }
    }
    {
      // ---- Ydiag on top of Yoff, same frame: the A operand is x~, the B operand the causal C.B^T fragment
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int ct = 0; ct < PT; ++ct) yo[ct][ti] = mfma16(xw[ct][0], cbv[ti == 0 ? 0 : ti == 1 ? 1 : ti == 2 ? 2 : 4], yo[ct][ti]);
#pragma unroll
      for (int ti = 2; ti < 4; ++ti)
#pragma unroll
        for (int ct = 0; ct < PT; ++ct) yo[ct][ti] = mfma16(xw[ct][1], cbv[ti == 2 ? 3 : 5], yo[ct][ti]);
    }
    HSTAMP(7);
    // ---- vectors of the next chunk (every read of this chunk's is done; ev is in registers) and the re-basing of the
    // frame where it is due
    if (c + 1 < nchunks) {
      mode = mode_next;
      reset_cur = __builtin_amdgcn_readfirstlane((int)reset_next) != 0;
      std_cur = __builtin_amdgcn_readfirstlane((int)std_next) != 0;
      if (mode == 1) rebase_state((int)f_step);
    }
    HSTAMP(8);
    // ---- y = row factor * accumulators + D x, rounded to bf16 and stored
    const f32x2 dh2 = {Dh, Dh};
    const bool full = (c + 1) * HQ <= L;
    u32x2 xrv[4][PT];               // all reads first: the stores below are memory barriers for the compiler
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int ct = 0; ct < PT; ++ct) {
        typedef __attribute__((address_space(3))) const u32x2 lds_u32x2;
        xrv[ti][ct] = *(lds_u32x2*)(size_t)(xt + xv_lo + 32 * ct + ti * (16 * XROW));
      }
    auto finish = [&](int ct, int ti) {       // the lane's 8 bytes of y: columns 16 ct + 4 kq + 0..3 of token 16 ti + lc
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      const u32x2 xr = xrv[ti][ct];
      const f32x2 e2 = {ev[ti], ev[ti]};
      const f32x2 y0 = __builtin_elementwise_fma(f32x2{yo[ct][ti][0], yo[ct][ti][1]}, e2, dh2 * f32x2{bf16_lo(xr[0]), bf16_hi(xr[0])});
      const f32x2 y1 = __builtin_elementwise_fma(f32x2{yo[ct][ti][2], yo[ct][ti][3]}, e2, dh2 * f32x2{bf16_lo(xr[1]), bf16_hi(xr[1])});
      const bf16x2 p01 = {(bf16_t)y0[0], (bf16_t)y0[1]}, p23 = {(bf16_t)y1[0], (bf16_t)y1[1]};
      return u32x2{__builtin_bit_cast(unsigned, p01), __builtin_bit_cast(unsigned, p23)};
    };
#if TV_HEAD_Y16
    // Two t-tiles at a time: v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of the
    // other, after which a lane holds 16 contiguous bytes — columns 16 ct + 8 (kq >> 1) + 0..7 of token
    // 16 (tp + (kq & 1)) + lc — and the chunk leaves in 2 PT stores of 16 bytes instead of 4 PT of 8 (the epilogue is
    // bound by the issue of its stores).
    const unsigned yo16 = yoff16;      // (an asm operand alone does not capture the variable in a generic lambda)
#pragma unroll
    for (int tp = 0; tp < 4; tp += 2) {
      const void* yrow = uniform_ptr(ygs + (int64_t)(c * HQ + 16 * tp) * a.ysl);
      const bool ok = (full || c * HQ + 16 * tp + 16 * (kq & 1) + lc < L) && !HDBG(a, 4);
#pragma unroll
      for (int ct = 0; ct < PT; ++ct) {
        const u32x2 ya = finish(ct, tp), yb = finish(ct, tp + 1);
        const auto s0 = __builtin_amdgcn_permlane16_swap(ya[0], yb[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(ya[1], yb[1], false, false);
        const u32x4v w = {s0[0], s1[0], s0[1], s1[1]};
        if (ok) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3" :: "v"(yo16), "v"(w), "s"(yrow), "n"(32 * ct) : "memory");
      }
    }
#else
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const void* yrow = uniform_ptr(ygs + (int64_t)(c * HQ + 16 * ti) * a.ysl);
      const bool ok = (full || c * HQ + 16 * ti + lc < L) && !HDBG(a, 4);
#pragma unroll
      for (int ct = 0; ct < PT; ++ct) store_y_tile(yrow, ok, ct, finish(ct, ti));
    }
#endif
  };

  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    HSTAMP(0);
    step(c, more);
    HSTAMP(9);
    // (the copies of the next chunk landed before the C.B^T wait; this step's y stores stay in flight)
    if (UNTRACKED && (c + 1) * HQ <= L) HEAD_BARRIER(TV_HEAD_Y16 ? 2 * PT : 4 * PT);
    else HEAD_BARRIER(0);
  }
#ifdef TV_HEAD_STAMP
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0)
    for (int i = 0; i < 12; ++i) g_head_phases[16 * (wave & 1) + i] = ph_acc[i];
#endif
  // final state of this segment, X = 2^E X'
  {
    const float sc = __builtin_amdgcn_exp2f(E);
    float* fin = a.nseg > 1 ? a.seg_state + (int64_t)seg * gridDim.y * a.H * P * HN : a.final_state;
    // (the lane's indices are formed again behind the loop: held across it they cost the fast kernel its only spill)
    int lane_e = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_e));
    const int lc_e = lane_e & 15, kq_e = lane_e >> 4;
    if (fin) {
#pragma unroll
      for (int ct = 0; ct < PT; ++ct)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 v = xacc[ct][i];
          *(f32x4*)(fin + (((int64_t)b * a.H + h) * P + 16 * ct + lc_e) * HN + 32 * (i >> 1) + 8 * kq_e + 4 * (i & 1)) =
              f32x4{v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc};
        }
    }
    float* td = a.nseg > 1 ? a.seg_decay + (int64_t)seg * gridDim.y * a.H : a.total_decay;
    if (td && lane_e == 0) td[(int64_t)b * a.H + h] = decay_total;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// head_dim 80, four heads per work-group: the step as ONE hand-scheduled instruction stream (ssd_head_step.inc,
// generated by devtools/gen_head_step.py).  The C++ below keeps what runs once per chunk outside the matrix work —
// the vectors of the next chunk (prep), the mode decision, the scalar addresses — and hands everything else to the
// asm body: state in a0..a159 and the y accumulators in a160..a239 for the whole march, v86..v255 are the body's,
// nothing lives in those registers across a statement except the state.  Same arithmetic in the same order per
// accumulator as ssd_head_kernel<5,4,2> (bit-identical y and final state).
#ifdef TV_HEAD_STAMP
#include "ssd_head_step_stamp.inc"       // dev: python devtools/gen_head_step.py --stamps > csrc/ssd_head_step_stamp.inc
#define TV_STEP_STAMP_OPS , [stl] "+s"(st_last), [st0] "+s"(st0), [st1] "+s"(st1), [st2] "+s"(st2), [st3] "+s"(st3), [st4] "+s"(st4), [st5] "+s"(st5), \
    [st6] "+s"(st6), [st7] "+s"(st7), [st8] "+s"(st8), [st9] "+s"(st9), [st10] "+s"(st10)
#else
#include "ssd_head_step.inc"
#define TV_STEP_STAMP_OPS
#endif

struct __attribute__((aligned(16))) HeadVecA {
  HeadVec v;
  float one[HQ];      // weights / row factors of a standard step's Ydiag and epilogue
};
struct __attribute__((aligned(16))) HeadSmemA {
  bf16_t bt[2][HQ * HN];
  bf16_t ct[2][HQ * HN];
  bf16_t xr[4][2][HQ * 80 + 256];
  HeadVecA v[4];
};

__global__ __launch_bounds__(256) void ssd_head_asm_kernel(HeadArgs a) {
  typedef HeadSmemA Smem;
  constexpr int PT = 5, NW = 4, NB = 2, P = 80, KP = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, kq = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int b = blockIdx.y;
  const int hpg = a.H / a.G;
  const int g = blockIdx.x % a.G;
  const int hig = (blockIdx.x / a.G) * NW + wave;
  const int h = a.group_map ? (hig * a.G + g) : (g * hpg + hig);
  const int seg = blockIdx.z;
  const int c_first = seg * a.seg_chunks;
  const int t_first = c_first * HQ;
  const int nchunks = min(a.seg_chunks, a.nchunks - c_first);
  const int L = min(a.L - t_first, nchunks * HQ);
  HeadVec& vec = sm.v[wave].v;
  const unsigned lds0 = lds_addr_of(smem_raw);
  const unsigned lds_bt = lds0 + (unsigned)offsetof(Smem, bt), lds_ct = lds0 + (unsigned)offsetof(Smem, ct);
  const unsigned lds_xr = lds0 + (unsigned)offsetof(Smem, xr) + wave * (unsigned)sizeof(sm.xr[0]);
  const unsigned lds_vec = lds0 + (unsigned)offsetof(Smem, v) + wave * (unsigned)sizeof(HeadVecA);
  constexpr unsigned XSLOT = sizeof(sm.xr[0][0]);
  {
    float* vz = reinterpret_cast<float*>(&sm.v[wave]);
    for (int i = lane; i < (int)(sizeof(HeadVecA) / 4); i += 64) vz[i] = 0.f;
    sm.v[wave].one[lane] = 1.f;
  }

  // ---- B / C copies (as in ssd_head_kernel): piece k of this wave = token rows 16 wave + 4 k + (lane >> 4)
  const bf16_t* Bg = a.Bm + (int64_t)b * a.bsb + (int64_t)g * a.bsg + (int64_t)t_first * a.bsl;
  const bf16_t* Cg = a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg + (int64_t)t_first * a.csl;
  const int bc_row0 = 4 * KP * wave + (lane >> 4);
  const unsigned off_b0 = (unsigned)((bc_row0 * a.bsl + ((lane & 15) ^ (4 * ((lane >> 4) & 3))) * 8) * 2);
  const unsigned off_c0 = (unsigned)(bc_row0 * a.csl * 2);
  const unsigned cgc0 = (unsigned)(((lane & 15) ^ (lane >> 4)) << 4);
  const unsigned r4b = (unsigned)(4 * a.bsl * 2), r4c = (unsigned)(4 * a.csl * 2);
  const unsigned ob0 = off_b0, ob1 = off_b0 + r4b - 1024u, ob2 = off_b0 + 2 * r4b - 2048u, ob3 = off_b0 + 3 * r4b - 3072u;
  const unsigned oc0 = off_c0 + cgc0, oc1 = off_c0 + r4c + (cgc0 ^ 64u) - 1024u, oc2 = off_c0 + 2 * r4c + (cgc0 ^ 128u) - 2048u,
                 oc3 = off_c0 + 3 * r4c + (cgc0 ^ 192u) - 3072u;
  auto issue_bc_tail = [&](int c, int which) {       // a chunk with fewer than 64 rows: rows past the end repeat the last row
    const int slot = c % NB;
    const int t0 = c * HQ;
    const bf16_t* Tc = which ? Cg + (int64_t)t0 * a.csl : Bg + (int64_t)t0 * a.bsl;
    const int64_t rl = which ? a.csl : a.bsl;
    const unsigned dst = (which ? lds_ct : lds_bt) + slot * (HQ * HN * 2);
    const void* sp = uniform_ptr(Tc);
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int row = 4 * KP * wave + 4 * k + (lane >> 4);
      const int rr = min(row, L - 1 - t0);
      const int cg = which ? (lane & 15) ^ (row & 15) : (lane & 15) ^ (4 * (row & 3));
      glds16(sp, (unsigned)((rr * rl + cg * 8) * 2), dst + (KP * wave + k) * 1024);
    }
  };
  // ---- x copies
  constexpr int XROW = 2 * P, NPC = P / 8, RPI = 64 / NPC, NXI = (HQ + RPI - 1) / RPI;
  const bf16_t* xg = a.x + (int64_t)b * a.xsb + (int64_t)t_first * a.xsl + (int64_t)h * P;
  const int x_lrow = lane / NPC;
  const unsigned x_off = (unsigned)((x_lrow * a.xsl + (lane % NPC) * 8) * 2);
  const unsigned x_off_last = (unsigned)((min(x_lrow, HQ - 1 - RPI * (NXI - 1)) * a.xsl + (lane % NPC) * 8) * 2);
  const unsigned dj = (unsigned)(RPI * a.xsl * 2) - (unsigned)(RPI * XROW);
  const unsigned ox0 = x_off, ox1 = x_off + dj, ox2 = x_off + 2 * dj, ox3 = x_off + 3 * dj, oxl = x_off_last + 2 * dj;
  const unsigned xg4 = (unsigned)(RPI * 4 * a.xsl * 2);
  auto issue_x_tail = [&](int c) {
    const int t0 = c * HQ;
    const bf16_t* xc = xg + (int64_t)t0 * a.xsl;
#pragma unroll
    for (int k = 0; k < NXI; ++k)
      glds16(uniform_ptr(xc), (unsigned)((min(RPI * k + x_lrow, L - 1 - t0) * a.xsl + (lane % NPC) * 8) * 2),
             lds_xr + (c & 1) * XSLOT + RPI * k * XROW);
  };
  auto issue_full = [&](int c) {        // a whole chunk from C++ (prologue only)
    const void* sb = uniform_ptr(Bg + (int64_t)c * HQ * a.bsl);
    glds16x4(sb, ob0, ob1, ob2, ob3, lds_bt + (c % NB) * (HQ * HN * 2) + KP * wave * 1024);
    const void* sc = uniform_ptr(Cg + (int64_t)c * HQ * a.csl);
    glds16x4(sc, oc0, oc1, oc2, oc3, lds_ct + (c % NB) * (HQ * HN * 2) + KP * wave * 1024);
#pragma unroll
    for (int k = 0; k < NXI; ++k) {
      const void* sx = uniform_ptr(xg + (int64_t)(c * HQ + RPI * k) * a.xsl);
      glds16(sx, k == NXI - 1 ? x_off_last : x_off, lds_xr + (c & 1) * XSLOT + RPI * k * XROW);
    }
  };
  auto issue_chunk = [&](int c) {
    if ((c + 1) * HQ <= L) issue_full(c);
    else { issue_bc_tail(c, 0); issue_bc_tail(c, 1); issue_x_tail(c); }
  };

  // ---- lane parts of the LDS addresses the step reads (slot 0; the step adds the slot offsets)
  const int xr_lo = (8 * kq + q4) * XROW + 8 * p4, xv_lo = lc * XROW + 8 * kq;
  const int c_lo = lc * 256, c_z = (kq ^ lc) << 4, bsw = q4 << 6, b_lo = (8 * kq + q4) * 256 + p4 * 16;
  const unsigned ca0 = lds_ct + ((c_z ^ 0) + c_lo), ca1 = lds_ct + ((c_z ^ 64) + c_lo), ca2 = lds_ct + ((c_z ^ 128) + c_lo),
                 ca3 = lds_ct + ((c_z ^ 192) + c_lo);
  const unsigned ba0 = lds_bt + ((bsw ^ 0) + b_lo), ba1 = lds_bt + ((bsw ^ 64) + b_lo), ba2 = lds_bt + ((bsw ^ 128) + b_lo),
                 ba3 = lds_bt + ((bsw ^ 192) + b_lo);
  const unsigned xtr = lds_xr + xr_lo, xvr = lds_xr + xv_lo;
  const unsigned vw_wts = lds_vec + (unsigned)offsetof(HeadVec, wts) + 32 * kq;
  const unsigned vev_ecs = lds_vec + (unsigned)offsetof(HeadVec, ecs) + 4 * lc;
  const unsigned vev_one = lds_vec + (unsigned)offsetof(HeadVecA, one) + 4 * lc;
  // standard steps: the mask factors' addresses (token t = 16 (kq >> 1) + lc, columns s = 8 kq + j) and the diagonal test
  const unsigned sa_t = lds_vec + 4 * (16 * (kq >> 1) + lc), sa_s = lds_vec + 32 * kq, sa_lc = lds_vec + 4 * lc, sa_s15 = lds_vec + 32 * (kq & 1);
  const int sd0 = 16 * (kq >> 1) + lc - 8 * kq;
  const unsigned cbo = lane * 16, dto = lane * 2;
  const unsigned yst = (unsigned)(((lane / 10) * a.ysl + (lane % 10) * 8) * 2);      // row stores: lane = 16-byte piece of 6 rows of 160 bytes
  const int lrow = lane / 10;
  const unsigned xsr = lds_xr + 16 * lane;
  bf16_t* const ygs = a.y + (int64_t)b * a.ysb + (int64_t)t_first * a.ysl + (int64_t)h * P;
  const bf16_t* cbg = a.cb + (((int64_t)b * a.G + g) * a.nchunks + c_first) * CBE;
  const unsigned y6 = (unsigned)(12 * a.ysl);      // bytes of 6 rows of y

  // ---- per-chunk vectors (the same decisions and arithmetic as ssd_head_kernel's prep)
  const float Ah = a.A[h];
  const float bias = a.dt_bias ? a.dt_bias[h] : 0.f;
  const float Dh = a.D ? a.D[h] : 0.f;
  const bf16_t* dtg = a.dt + (int64_t)b * a.dsb + (int64_t)t_first * a.dsl + (int64_t)h * a.dsh;      // head-major: dsl == 1
  float decay_total = 0.f;
  float E = 0.f;
  bool reset_next = false, std_next = false;
  unsigned dead_next = 0;        // standard steps: bit ti = every row factor of t-tile ti has underflowed to zero
  auto prep = [&](int c, unsigned raw_bits, float& f_out) __attribute__((always_inline)) {
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
    f_out = mode == 1 ? -mshift : 1.f;
    vec.cs[lane] = cs2;
    vec.dtv[lane] = d;
    const float rowf = __builtin_amdgcn_exp2f(cs2 + Euse);
    vec.ecs[lane] = rowf;
    {
      const unsigned long long nz = __builtin_amdgcn_ballot_w64(rowf != 0.f);
      dead_next = ((nz & 0xffffull) == 0 ? 1u : 0u) | (((nz >> 16) & 0xffffull) == 0 ? 2u : 0u) | (((nz >> 32) & 0xffffull) == 0 ? 4u : 0u) |
                  ((nz >> 48) == 0 ? 8u : 0u);
    }
    const bool rst = TV_HEAD_RESET && mode != 2 && cl2 <= -RESET_THR;
    const bool ustd = mode == 2;
    reset_next = rst || ustd;
    std_next = ustd;
    vec.wts[lane] = __builtin_amdgcn_exp2f(mode == 2 ? cl2 - cs2 : rst ? cl2 - cs2 - RMAX : -cs2 - Euse) * d;
    if (rst) vec.wtd[c & 1][lane] = __builtin_amdgcn_exp2f(-cs2 - Euse) * d;
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
    return __builtin_amdgcn_readfirstlane(ustd ? 0 : mode);
  };

  // ---- state: zero, or the caller's initial state (first segment)
  asm volatile(TV_HEAD_STATE_ZERO ::: TV_HEAD_STATE_CLOBBERS);
  if (a.init && seg == 0) {
#pragma unroll
    for (int ct = 0; ct < PT; ++ct)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 v = *(const f32x4*)(a.init + (((int64_t)b * a.H + h) * P + 16 * ct + lc) * HN + 32 * (i >> 1) + 8 * kq + 4 * (i & 1));
        asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\tv_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
                     :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "n"(32 * ct + 4 * i), "n"(32 * ct + 4 * i + 1), "n"(32 * ct + 4 * i + 2), "n"(32 * ct + 4 * i + 3)
                     : TV_HEAD_STATE_CLOBBERS);
      }
  }

  // ---- prologue: chunk 0's copies, its vectors
  issue_chunk(0);
  unsigned dt_next = *(const unsigned short*)(dtg + lane);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(dt_next) :: "memory");
  float f_step;
  int mode = prep(0, dt_next, f_step);
  bool reset_cur = __builtin_amdgcn_readfirstlane((int)reset_next) != 0;
  bool std_cur = __builtin_amdgcn_readfirstlane((int)std_next) != 0;
  unsigned dead_cur = dead_next;
  dt_next = *(const unsigned short*)(dtg + (int64_t)min(1, nchunks - 1) * HQ + lane);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(dt_next) :: "memory");
  HEAD_BARRIER(0);

#ifdef TV_HEAD_STAMP
  unsigned st_last = (unsigned)clock64(), st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, st6 = 0, st7 = 0, st8 = 0, st9 = 0, st10 = 0, st11 = 0, st12 = 0;
#endif
  // The vectors of chunk c + 1 are made inside step c (beside the MFMAs of its state update); what the next step's set-up needs
  // comes back in scalar registers: its flags (re-base / reset / standard / dead t-tiles), the re-basing shift, E, the decay total.
  unsigned nflags = (mode == 1 ? 1u : 0u) | (reset_cur ? 2u : 0u) | (std_cur ? 4u : 0u) | (dead_cur << 4);
  int nsh = mode == 1 ? (int)f_step : 0;
  const unsigned vvec = lds_vec + 4 * lane;
  const unsigned vw2_0 = lds_vec + (unsigned)offsetof(HeadVec, wtd) + 32 * kq;
  const unsigned lb0 = lds_bt + KP * wave * 1024, lc0 = lds_ct + KP * wave * 1024;
  const unsigned fl_const = a.softplus ? (1u << 8) : 0u;
  // running scalar pointers (bytes): C.B^T and y of chunk c, dt of chunk min(c + 2, last), B / C / x of chunk c + 1
  const char* pcb_r = (const char*)uniform_ptr(cbg + 1024);              // (the step's offsets are -2048 .. 3072)
  const char* py_r = (const char*)uniform_ptr(ygs);
  const char* pdt_r = (const char*)uniform_ptr(dtg + (int64_t)min(2, nchunks - 1) * HQ);
  const char* pb_r = (const char*)uniform_ptr(Bg + (int64_t)HQ * a.bsl);
  const char* pc_r = (const char*)uniform_ptr(Cg + (int64_t)HQ * a.csl);
  const char* px_r = (const char*)uniform_ptr(xg + (int64_t)HQ * a.xsl);
  const int64_t y_step = (int64_t)HQ * a.ysl * 2, b_step = (int64_t)HQ * a.bsl * 2, c_step = (int64_t)HQ * a.csl * 2, x_step = (int64_t)HQ * a.xsl * 2;
  unsigned sbc = 0, sxs = 0;
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    const bool copy = more && (c + 2) * HQ <= L;
    const int rem = min(L - c * HQ, HQ);
    const unsigned flags = __builtin_amdgcn_readfirstlane(nflags | (copy ? 8u : 0u) | fl_const);
    const int sh = __builtin_amdgcn_readfirstlane(nsh);
    const unsigned vw2 = vw2_0 + (c & 1) * 256;       // (reset steps: Ydiag's weights, the old frame's)
    const unsigned vev = (flags & 4u) ? vev_one : vev_ecs;
    const void* pcb = pcb_r; const void* py = py_r; const void* pdt = pdt_r;
    const void* pb = pb_r; const void* pc = pc_r; const void* px = px_r;
    pcb_r += CBE * 2; py_r += y_step; pb_r += b_step; pc_r += c_step; px_r += x_step;
    if (c + 3 < nchunks) pdt_r += HQ * 2;
    const unsigned lb = lb0 + (sbc ^ (HQ * HN * 2)), lcc = lc0 + (sbc ^ (HQ * HN * 2)), lx = lds_xr + (sxs ^ XSLOT);
    const int lrem1 = L - (c + 1) * HQ;
    const unsigned par1 = (unsigned)(c + 1) & 1u;
    unsigned dt_new, o_flags, o_e, o_dtot, o_cl2;
    int o_sh;
    asm volatile(TV_HEAD_STEP_ASM
                 : [dtout] "=&v"(dt_new), [oflags] "=&s"(o_flags), [osh] "=&s"(o_sh), [oe] "=&s"(o_e), [odtot] "=&s"(o_dtot), [ocl2] "=&s"(o_cl2) TV_STEP_STAMP_OPS
                 : [ca0] "v"(ca0), [ca1] "v"(ca1), [ca2] "v"(ca2), [ca3] "v"(ca3), [ba0] "v"(ba0), [ba1] "v"(ba1), [ba2] "v"(ba2), [ba3] "v"(ba3),
                   [xtr] "v"(xtr), [xvr] "v"(xvr), [vw] "v"(vw_wts), [vw2] "v"(vw2), [vev] "v"(vev), [cbo] "v"(cbo), [yst] "v"(yst), [lrow] "v"(lrow), [xsr] "v"(xsr), [dto] "v"(dto),
                   [ob0] "v"(ob0), [ob1] "v"(ob1), [ob2] "v"(ob2), [ob3] "v"(ob3), [oc0] "v"(oc0), [oc1] "v"(oc1), [oc2] "v"(oc2), [oc3] "v"(oc3),
                   [ox0] "v"(ox0), [ox1] "v"(ox1), [ox2] "v"(ox2), [ox3] "v"(ox3), [oxl] "v"(oxl),
                   [at] "v"(sa_t), [as] "v"(sa_s), [alc] "v"(sa_lc), [as15] "v"(sa_s15), [d0] "v"(sd0),
                   [dtin] "v"(dt_next), [lane] "v"(lane), [vvec] "v"(vvec),
                   [sbc] "s"(sbc), [sxs] "s"(sxs), [flags] "s"(flags), [sh] "s"(sh), [dh] "s"(Dh), [pcb] "s"(pcb), [py] "s"(py), [y6] "s"(y6), [rem] "s"(rem),
                   [pdt] "s"(pdt), [pb] "s"(pb), [pc] "s"(pc), [px] "s"(px), [xg4] "s"(xg4), [lb] "s"(lb), [lc] "s"(lcc), [lx] "s"(lx),
                   [lrem1] "s"(lrem1), [par1] "s"(par1), [bias] "s"(bias), [ah] "s"(Ah), [dtmin] "s"(a.dt_min), [dtmax] "s"(a.dt_max), [ein] "s"(E), [dtot] "s"(decay_total)
                 : TV_HEAD_STEP_CLOBBERS);
    if (more && !copy) {       // the next chunk is the sequence's last, partial one
      issue_bc_tail(c + 1, 0);
      issue_bc_tail(c + 1, 1);
      issue_x_tail(c + 1);
    }
    if (more) {       // the step has prepared chunk c + 1
      nflags = o_flags; nsh = o_sh;
      E = __uint_as_float(o_e); decay_total = __uint_as_float(o_dtot);
      if (a.chunk_tot && lane == 0) a.chunk_tot[((int64_t)b * a.H + h) * a.nchunks + c_first + c + 1] = __uint_as_float(o_cl2);
    }
    dt_next = dt_new;
    sbc ^= HQ * HN * 2; sxs ^= XSLOT;
#ifdef TV_HEAD_STAMP
    { unsigned long long n__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n__) :: "memory"); st11 += (unsigned)n__ - st_last; st_last = (unsigned)n__; }
#endif
    if (copy) HEAD_BARRIER(11);       // (the step's eleven row stores may stay in flight)
    else HEAD_BARRIER(0);
#ifdef TV_HEAD_STAMP
    { unsigned long long n__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n__) :: "memory"); st12 += (unsigned)n__ - st_last; st_last = (unsigned)n__; }
#endif
  }
#ifdef TV_HEAD_STAMP
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
    const unsigned sv[13] = {st0, st1, st2, st3, st4, st5, st6, st7, st8, st9, st10, st11, st12};
    for (int i = 0; i < 13; ++i) g_head_phases[16 * wave + i] = sv[i];
  }
#endif
  // ---- final state of this segment, X = 2^E X'
  {
    const float sc = __builtin_amdgcn_exp2f(E);
    float* fin = a.nseg > 1 ? a.seg_state + (int64_t)seg * gridDim.y * a.H * P * HN : a.final_state;
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    if (fin) {
#pragma unroll
      for (int ct = 0; ct < PT; ++ct)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float v0, v1, v2, v3;
          asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                       : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3)
                       : "n"(32 * ct + 4 * i), "n"(32 * ct + 4 * i + 1), "n"(32 * ct + 4 * i + 2), "n"(32 * ct + 4 * i + 3));
          *(f32x4*)(fin + (((int64_t)b * a.H + h) * P + 16 * ct + lc) * HN + 32 * (i >> 1) + 8 * kq + 4 * (i & 1)) =
              f32x4{v0 * sc, v1 * sc, v2 * sc, v3 * sc};
        }
    }
    float* td = a.nseg > 1 ? a.seg_decay + (int64_t)seg * gridDim.y * a.H : a.total_decay;
    if (td && lane == 0) td[(int64_t)b * a.H + h] = decay_total;
  }
}

// dt (B, L, H) -> (B, H, Lp) with Lp = 64 nchunks: a wave (= a head) then reads the 64 tokens of a chunk as ONE 128-byte line.
// Token-major, the same 64 values are 2 bytes each out of 64 lines that all 128 heads share; the lines are evicted between
// the visits of the work-groups that want them and came from HBM 11 times over (profiles/r04_ssd_scan_read_attribution.json:
// 0.48 GB of reads for 0.04 GB of dt at 164 k tokens).  grid (nchunks, B), 256 threads; tokens past L are written as zeros.
__global__ __launch_bounds__(256) void ssd_dt_transpose_kernel(const bf16_t* __restrict__ dt, bf16_t* __restrict__ out, int L, int H,
                                                               int64_t dsb, int64_t dsl, int64_t lp, int vec, unsigned* __restrict__ zero2) {
  // (the first launch of a scan call also clears the two counters of the correction pass's walker list)
  if (zero2 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) zero2[threadIdx.x] = 0;
  __shared__ __attribute__((aligned(16))) unsigned short tile[HQ][136];       // [token][head], rows of 272 bytes (16-byte multiple)
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const unsigned short* src = (const unsigned short*)dt + (int64_t)b * dsb;
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  for (int h0 = 0; h0 < H; h0 += 128) {
    const int nh = min(128, H - h0);
    if (vec) {          // 16-byte loads: 8 heads of a token (rows and base 16-byte aligned, H a multiple of 8)
      for (int i = tid; i < HQ * 16; i += 256) {
        const int t = i >> 4, hc = i & 15, tok = c * HQ + t;
        u16x8 v = {};
        if (tok < L && 8 * hc < nh) v = *(const u16x8*)(src + (int64_t)tok * dsl + h0 + 8 * hc);
        *(u16x8*)(&tile[t][8 * hc]) = v;
      }
    } else {
      for (int i = tid; i < HQ * 128; i += 256) {
        const int t = i >> 7, h = i & 127, tok = c * HQ + t;
        tile[t][h] = (tok < L && h < nh) ? src[(int64_t)tok * dsl + h0 + h] : (unsigned short)0;
      }
    }
    __syncthreads();
    unsigned short* dst = (unsigned short*)out + ((int64_t)b * H + h0) * lp + (int64_t)c * HQ;
    for (int i = tid; i < 128 * 8; i += 256) {       // 16-byte stores: 8 tokens of a head
      const int h = i >> 3, tc = i & 7;
      if (h < nh) {
        u16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[8 * tc + j][h];
        *(u16x8*)(dst + (int64_t)h * lp + 8 * tc) = v;
      }
    }
    __syncthreads();
  }
}

// tv_ssd_head_set_asm(0) / TV_HEAD_ASM=0 sends head_dim 80 x 4 heads back to the C++ step (A/B runs, the bit-identity test)
std::atomic<int> g_head_asm{-1};
bool head_asm_enabled() {
  const int f = g_head_asm.load(std::memory_order_relaxed);
  if (f >= 0) return f != 0;
  static const bool on = [] { const char* e = getenv("TV_HEAD_ASM"); return !e || atoi(e) != 0; }();
  return on;
}
static_assert(sizeof(HeadSmemA) <= 160 * 1024, "LDS budget");

// heads of one group per work-group: 4, 2 or 1
int pick_nw(int hpg) { return hpg % 4 == 0 ? 4 : hpg % 2 == 0 ? 2 : 1; }

// 1 024 waves fill the chip (one per SIMD); a segment is at least 16 chunks long
int pick_segments(int batch, int nheads, int nchunks) {
  if (const char* e = getenv("TV_SSD_NSEG")) return atoi(e) > 0 ? atoi(e) : 1;     // dev tool
  const int waves = batch * nheads;
  int nseg = waves >= 768 ? 1 : 1024 / waves;
  if (nseg > 16) nseg = 16;
  while (nseg > 1 && nchunks / nseg < 16) --nseg;
  return nseg < 1 ? 1 : nseg;
}

struct HeadLayout {
  size_t cb, seg_state, seg_decay, ctot, corr, dtt, total;
  int nseg, seg_chunks;
};
HeadLayout head_layout(int batch, int seqlen, int nheads, int headdim, int ngroups) {
  HeadLayout l;
  const size_t nchunks = (size_t)(seqlen + HQ - 1) / HQ;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  l.nseg = pick_segments(batch, nheads, (int)nchunks);
  l.seg_chunks = (int)((nchunks + l.nseg - 1) / l.nseg);
  l.cb = 0;
  l.seg_state = up((size_t)batch * ngroups * nchunks * CBE * sizeof(bf16_t));
  const size_t st = (size_t)batch * nheads * headdim * HN * sizeof(float);
  l.seg_decay = l.seg_state + (l.nseg > 1 ? up(l.nseg * st) : 0);
  l.ctot = l.seg_decay + (l.nseg > 1 ? up((size_t)l.nseg * batch * nheads * sizeof(float)) : 0);
  l.corr = l.ctot + (l.nseg > 1 ? up((size_t)batch * nheads * nchunks * sizeof(float)) : 0);
  l.dtt = l.corr + (l.nseg > 1 ? up(tv_ssd_correct_all_workspace_bytes(batch, nheads, (int)nchunks, l.nseg, headdim)) : 0);
  l.total = l.dtt + up((size_t)batch * nheads * nchunks * HQ * sizeof(bf16_t));       // dt, head-major
  return l;
}

template <int PT, int NW>
hipError_t launch_head(const HeadArgs& a, dim3 grid, hipStream_t st) {
  constexpr int NB = 2;
  typedef HeadSmem<PT, NW, NB> Smem;
  static_assert(sizeof(Smem) <= 160 * 1024, "LDS budget");
  hipError_t e = hipFuncSetAttribute((const void*)ssd_head_kernel<PT, NW, NB>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem));
  if (e != hipSuccess) return e;
  ssd_head_kernel<PT, NW, NB><<<grid, NW * 64, sizeof(Smem), st>>>(a);
  return hipSuccess;
}

}  // namespace

// dt (B, L, H) -> head-major (B, H, 64 nchunks) for the march kernels of this file and of ssd_pair.hip
void tv_ssd_dt_transpose_launch(const void* dt, void* out, int batch, int seqlen, int nheads, int64_t dsb, int64_t dsl,
                                unsigned* zero2, hipStream_t st) {
  const int nchunks = (seqlen + HQ - 1) / HQ;
  const int vec = (((uintptr_t)dt) & 15) == 0 && dsl % 8 == 0 && dsb % 8 == 0 && nheads % 8 == 0;
  ssd_dt_transpose_kernel<<<dim3(nchunks, batch), 256, 0, st>>>((const bf16_t*)dt, (bf16_t*)out, seqlen, nheads, dsb, dsl,
                                                               (int64_t)nchunks * HQ, vec, zero2);
}

#ifdef TV_HEAD_STAMP
extern "C" int tv_ssd_head_debug_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_head_phases), sizeof(g_head_phases));
}
#endif

extern "C" void tv_ssd_head_set_asm(int on) { g_head_asm.store(on, std::memory_order_relaxed); }

bool tv_ssd_head_supported(int seqlen, int nheads, int headdim, int ngroups, int dstate, int dtype,
                           int64_t xsl, int64_t bsl, int64_t bsg, int64_t csl, int64_t csg, int64_t ysl,
                           const void* x, const void* Bm, const void* Cm, const void* y) {
  if (dtype != TV_BF16 || dstate != HN || seqlen < 1) return false;
  if (headdim != 32 && headdim != 64 && headdim != 80) return false;
  if (xsl % 8 || bsl % 8 || csl % 8 || bsg % 8 || csg % 8 || ysl % 8) return false;
  if (((uintptr_t)x & 15) || ((uintptr_t)Bm & 15) || ((uintptr_t)Cm & 15) || ((uintptr_t)y & 15)) return false;
  if (64 * xsl * 2 >= (1ll << 31) || 64 * bsl * 2 >= (1ll << 31) || 64 * csl * 2 >= (1ll << 31) ||
      64 * ysl * 2 >= (1ll << 31))
    return false;
  (void)nheads; (void)ngroups;
  return true;
}

size_t tv_ssd_head_workspace_bytes(int batch, int seqlen, int nheads, int headdim, int ngroups) {
  return head_layout(batch, seqlen, nheads, headdim, ngroups).total;
}

int tv_ssd_head_launch(const void* x, const void* dt, const void* A, const void* Bm, const void* Cm,
                       const void* D, const void* dt_bias, const void* init_state, void* y,
                       void* final_state, void* total_decay, int batch, int seqlen, int nheads, int headdim,
                       int ngroups, int64_t xsb, int64_t xsl, int64_t dsb, int64_t dsl, int64_t bsb,
                       int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg, int64_t ysb,
                       int64_t ysl, int dt_softplus, float dt_min, float dt_max, int group_map,
                       void* workspace, size_t workspace_bytes, const void* cb_pre, hipStream_t st) {
  const HeadLayout lay = head_layout(batch, seqlen, nheads, headdim, ngroups);
  TV_CHECK_ARG(workspace && workspace_bytes >= lay.total && (((uintptr_t)workspace) & 15) == 0,
               "ssd_head: workspace of %zu bytes (16-byte aligned) required, got %zu", lay.total, workspace_bytes);
  unsigned char* wsb = (unsigned char*)workspace;
  HeadArgs a;
  a.x = (const bf16_t*)x; a.dt = (const bf16_t*)dt; a.Bm = (const bf16_t*)Bm; a.Cm = (const bf16_t*)Cm;
  a.cb = cb_pre ? (const bf16_t*)cb_pre : (const bf16_t*)workspace;
  a.A = (const float*)A; a.D = (const float*)D; a.dt_bias = (const float*)dt_bias;
  a.init = (const float*)init_state; a.y = (bf16_t*)y;
  a.final_state = (float*)final_state; a.total_decay = (float*)total_decay;
  a.L = seqlen; a.H = nheads; a.P = headdim; a.G = ngroups;
  a.nchunks = (seqlen + HQ - 1) / HQ;
  a.nseg = lay.nseg; a.seg_chunks = lay.seg_chunks;
  a.seg_state = lay.nseg > 1 ? (float*)(wsb + lay.seg_state) : nullptr;
  a.seg_decay = lay.nseg > 1 ? (float*)(wsb + lay.seg_decay) : nullptr;
  a.chunk_tot = lay.nseg > 1 ? (float*)(wsb + lay.ctot) : nullptr;
  // dt head-major into the workspace (one small launch: 2 x 42 MB at 164 k tokens); both the march and the correction read it
  {
    const int64_t lp = (int64_t)a.nchunks * HQ;
    bf16_t* dtt = (bf16_t*)(wsb + lay.dtt);
    unsigned* counters = lay.nseg > 1 ? tv_ssd_correct_all_counters(wsb + lay.corr, batch, nheads, a.nchunks, lay.nseg, headdim) : nullptr;
    tv_ssd_dt_transpose_launch(dt, dtt, batch, seqlen, nheads, dsb, dsl, counters, st);
    a.dt = dtt; dt = dtt;
    dsb = (int64_t)nheads * lp; dsl = 1; a.dsh = lp;
  }
  a.xsb = xsb; a.xsl = xsl; a.dsb = dsb; a.dsl = dsl; a.bsb = bsb; a.bsl = bsl; a.bsg = bsg;
  a.csb = csb; a.csl = csl; a.csg = csg; a.ysb = ysb; a.ysl = ysl;
  a.softplus = dt_softplus; a.group_map = group_map; a.dt_min = dt_min; a.dt_max = dt_max;
  { const char* e = getenv("TV_HEAD_DBG"); a.dbg = e ? atoi(e) : 0; }
  if (!cb_pre) {
    const int rc = tv_ssd_cb_prepass_launch(Bm, Cm, workspace, batch, seqlen, ngroups, bsb, bsl, bsg, csb, csl, csg, st);
    if (rc != TV_OK) return rc;
  }
  const int hpg = nheads / ngroups;
  const int nw = pick_nw(hpg);
  const dim3 grid(ngroups * (hpg / nw), batch, a.nseg);
  hipError_t e = hipSuccess;
  const int key = headdim / 16 * 10 + nw;
  switch (key) {
    case 54:
      if (head_asm_enabled()) {
        e = hipFuncSetAttribute((const void*)ssd_head_asm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(HeadSmemA));
        if (e == hipSuccess) ssd_head_asm_kernel<<<grid, 256, sizeof(HeadSmemA), st>>>(a);
      } else {
        e = launch_head<5, 4>(a, grid, st);
      }
      break;
    case 52: e = launch_head<5, 2>(a, grid, st); break;
    case 51: e = launch_head<5, 1>(a, grid, st); break;
    case 44: e = launch_head<4, 4>(a, grid, st); break;
    case 42: e = launch_head<4, 2>(a, grid, st); break;
    case 41: e = launch_head<4, 1>(a, grid, st); break;
    case 24: e = launch_head<2, 4>(a, grid, st); break;
    case 22: e = launch_head<2, 2>(a, grid, st); break;
    case 21: e = launch_head<2, 1>(a, grid, st); break;
    default: TV_UNSUPPORTED("ssd_head: head_dim %d", headdim);
  }
  if (e != hipSuccess) {
    tv_set_error("ssd_head: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return TV_ERR_LAUNCH;
  }
  if (a.nseg > 1) {
    const int rc = tv_ssd_correct_all_launch(y, dt, A, Cm, dt_bias, a.seg_state, a.seg_decay, (float*)final_state,
                                             (float*)total_decay, a.chunk_tot, batch, seqlen, nheads, headdim,
                                             ngroups, a.nseg, a.seg_chunks, ysb, ysl, dsb, dsl, a.dsh, csb, csl, csg,
                                             dt_softplus, dt_min, dt_max, group_map, wsb + lay.corr, st);
    if (rc != TV_OK) return rc;
  }
  TV_LAUNCH_CHECK();
}
