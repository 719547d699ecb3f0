// Persistent bf16 GEMM on 256 x 192 tiles whose finished tile LEAVES SLOWLY ("drips"): the ViT linears of SURVEY 8f-4 (timm
// block via base_vision.py:146-170,274-278; InternVideo2 Attention / Mlp vit_scale_clean.py:188-320),
//     C[M][N] = epilogue( A[M][K] . W[N][K]^T ),   epilogues of gemm.hip (bias / bias + exact GELU / C += ...).
//
// Why a third kernel.  (1) The ViT's widths are multiples of 192, not of 256: 1 152 = 6 x 192 (4.5 x 256: the output
// projections burn 10 % of their MFMAs on a half-empty tile column, here and in the library's 256 x 256 macro-tile alike),
// 3 456 = 18 x 192, 4 352 = 22.7 x 192.  (2) gemm_persist.hip measured where the rest goes:
// the K loop runs at 1.5 PFLOP/s, but a tile's 128 KB of C cost 2.8 us in the CU's memory pipeline in front of the
// copies, whoever issues them and whoever waits, and the GELU is vector-pipe work in a burst — 20 % of the qkv launch,
// 44 % of fc1.  Both can only be hidden by SPREADING them over the next tile's K loop, which needs the packed tile to
// wait somewhere.  A 256 x 192 tile leaves room: the copy ring is 112 KiB, the remaining 48 KiB of LDS hold HALF a packed
// tile (128 rows x 384 B), the other half waits in 24 registers.
//
//   * K-tile = two phases (m-half 0, m-half 1) of 24 MFMAs per wave (v_mfma_f32_16x16x32_bf16, operands swapped as in
//     gemm.hip).  A wave (wr = wave / 2, wc = wave % 2) owns rows {32 wr .. +31} of both 128-row halves and columns
//     {96 wc .. +95}: 12 W fragments read once per K-tile, 4 A fragments per phase.  The two wave groups (waves 0-3 / 4-7 =
//     the two waves of every SIMD) run one barrier apart: one group's MFMAs beside the other's reads, copies and drip.
//   * copies (LDS-DMA, source-side XOR swizzle): per K-tile and wave A0 2 + A1 2 + W 3 instructions of 1 KiB.  Phase 1 of
//     K-tile t issues A1(t + 1), phase 2 issues W(t + 2) and A0(t + 2) into the buffers phase 1 just read; every phase waits
//     vmcnt(7) = everything but the two youngest phases' copies, and what a wait retires is read one phase later.
//   * epilogue: m-half 0 becomes final behind phase 1 of the last K-tile, m-half 1 behind phase 2.  Half 0 is packed
//     (bias added / old C added, rounded to bf16 — for GELU the PRE-activation) into the staging rows, half 1 into 24
//     registers.  During the NEXT tile: phases 0-5 read one 1 KiB piece of staging each and store it a phase later (whole
//     384-byte rows; the GELU is applied here, eight values per lane and phase), phase 6 writes the register half into
//     staging, phases 7-13 drip it the same way.  Accumulating epilogue: the old C of half 0 arrives by LDS-DMA IN the
//     staging rows (phases 14 / 15, same lane <-> address map as the drip) and is replaced in place; the old C of half 1
//     is loaded into the 24 registers at the start of the last K-tile (untracked loads, retired by the phase waits).
//   * bias: 96 columns per wave in two registers (lane = column), fetched by ds_bpermute in the epilogue — LDS is full.
//   * tiles: banded order and XCD eighths as gemm_persist.hip; edge tiles shifted back inside the matrix, stores masked.
// Takes K >= 1 152 (the drip needs 16 K-tiles), K % 128 == 0, N % 8 == 0, M >= 256, N >= 192, fp32 bias.  Results: the same
// rounding points as gemm.hip (bit-equal for the bias epilogues: same K order per accumulator).
#include <stdio.h>
#include "gemm_common.hpp"

namespace tvgemm {
namespace {

constexpr int DBN = 192;
constexpr int A_HALF = 128 * BK * 2;              // 16 KiB: 128 rows x 128 B
constexpr int W_TILE = DBN * BK * 2;              // 24 KiB
constexpr int KT_BYTES = 2 * A_HALF + W_TILE;     // one K-tile: A0 A1 W
constexpr int RING = 2 * KT_BYTES;                // 112 KiB: [A0 x 2 parities][A1 x 2][W x 2] (every fragment offset < 64 KiB)
constexpr int A1_OFF = 2 * A_HALF, W_OFF = 4 * A_HALF;
constexpr int STAGE_ROW = DBN * 2;                // 384 B
constexpr int STAGE_BYTES = 128 * STAGE_ROW;      // 48 KiB
constexpr int LDS_TOTAL = RING + STAGE_BYTES;     // 160 KiB exactly
static_assert(LDS_TOTAL == 163840, "LDS budget");
enum { KT_FIRST = 0, KT_MID = 1, KT_PRELAST = 2, KT_LAST = 3 };
constexpr int MIN_KT = 18;

struct DArgs {
  GemmArgs g;
  int ntiles, tiles_n;
  int dbg;       // dev switches (TV_GEMM_DBG): 1 = no epilogue, 2 = no stores, 4 = no drip / dump, 8 = no conversion of the accumulators
#ifdef TV_DRIP_STAMP
  unsigned long long* stamps;      // dev build: [work-group < 8][wave][2 phases][5 sums + count]
#endif
};

#ifdef TV_DRIP_STAMP
__device__ __forceinline__ unsigned long long drip_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define DRIP_STAMP(i) do { const unsigned long long n_ = drip_now(); st_sum[st_ph][i] += n_ - st_t; st_t = n_; } while (0)
#else
#define DRIP_STAMP(i) do { } while (0)
#endif

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4e __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) u32x4e lds_u32x4;

#include "gemm_drip_regs.inc"

template <int EPI>
__global__ __launch_bounds__(512) void gemm_drip_kernel(DArgs da) {
  const GemmArgs& a = da.g;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wr = wave >> 1, wc = wave & 1;
  const int lc = lane & 15, kq = lane >> 4;

  // ---- this work-group's tiles: positions slot, slot + wpx, ... of the XCD's eighth of the banded order
  const int wpx = (int)gridDim.x >> 3;
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int chunk = (da.ntiles + 7) >> 3;
  int idx = xcd * chunk + slot;
  const int idx_end = min(da.ntiles, (xcd + 1) * chunk);
  if (idx >= idx_end) return;

  const int lda = (int)a.lda, ldw = (int)a.ldw, ldc = (int)a.ldc;
  const int per_band = a.group_m * da.tiles_n;
  auto decode = [&](int i, int& m0, int& n0) __attribute__((always_inline)) {
    const int band = i / per_band, in_band = i - band * per_band;
    const int rows = min(a.group_m, a.tiles_m - band * a.group_m);
    const int tn = in_band / rows, tm = band * a.group_m + (in_band - tn * rows);
    m0 = __builtin_amdgcn_readfirstlane(min(tm * BM, a.M - BM));
    n0 = __builtin_amdgcn_readfirstlane(min(tn * DBN, a.N - DBN));
  };
  // rows / columns of a shifted edge tile that belong to its neighbour
  auto skip_rows = [&](int m0) __attribute__((always_inline)) { return (-m0) & (BM - 1); };
  auto skip_cols = [&](int n0) __attribute__((always_inline)) {
    const int r = n0 % DBN;
    return r ? DBN - r : 0;
  };

  const unsigned lds0 = lds_addr_of(smem_raw);
  const unsigned stage0 = lds0 + RING;

  // ---- copies: A half = 16 pieces of 1 KiB (8 rows x 128 B), wave w copies pieces 2w, 2w + 1; W = 24 pieces, wave w
  // copies 3w .. 3w + 2 (one M0 set-up per item, the later pieces through the instruction offset, which moves source and
  // destination alike: their source offsets are passed less 1 024 j).  LDS byte (row, physical chunk p) holds source chunk
  // p ^ f(row), f(row) = (row >> 1) & 7.
  unsigned voffA[2], voffW[3];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 16 * wave + 8 * j + (lane >> 3);
    const int ch = (lane & 7) ^ ((row >> 1) & 7);
    voffA[j] = (unsigned)((row * lda + ch * 8) * 2) - (unsigned)(1024 * j);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int row = 24 * wave + 8 * j + (lane >> 3);
    const int ch = (lane & 7) ^ ((row >> 1) & 7);
    voffW[j] = (unsigned)((row * ldw + ch * 8) * 2) - (unsigned)(1024 * j);
  }
  auto dma2 = [&](const void* sp, unsigned v0, unsigned v1, unsigned dst) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "s"(sp), "s"(dst) : "memory");
  };
  auto dma3 = [&](const void* sp, unsigned v0, unsigned v1, unsigned v2, unsigned dst) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %4\n\t"
                 "global_load_lds_dwordx4 %2, %4 offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, %4 offset:2048\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "s"(sp), "s"(dst) : "memory");
  };

  // ---- fragment reads: 16 rows x 64 bytes, lane (lc = row, kq = 16-byte chunk of the k-step)
  const unsigned frag_lo = (unsigned)(lc * 128 + ((kq ^ (lc >> 1)) << 4));
  const unsigned a_b0 = lds0 + (unsigned)(wr * 32 * 128) + frag_lo, a_b1 = lds0 + (unsigned)(wr * 32 * 128) + (frag_lo ^ 64u);
  const unsigned w_b0 = lds0 + (unsigned)(W_OFF + wc * 96 * 128) + frag_lo;
  const unsigned w_b1 = lds0 + (unsigned)(W_OFF + wc * 96 * 128) + (frag_lo ^ 64u);
#define DG_LD(base, off) (*(lds_bf16x8*)(size_t)((base) + (unsigned)(off)))

  // accumulators [m-half][m-tile][n-tile] = a[0:95], m-half 1 of the finished tile (packed; accumulating epilogue: its
  // old C first) = a[96:119]: hardware registers named in gemm_drip_regs.inc, never values the compiler allocates
  bf16x8 af[2][2];             // [m-tile][k-step]
  bf16x8 wfr[6][2];            // [n-tile][k-step]
  u32x4e dd = {0u, 0u, 0u, 0u};    // the piece of staging on its way out
  // fp32 bias of the NEXT tile in accumulator layout (n-tile n: columns 96 wc + 16 n + 4 kq + 0..3): the accumulators of a
  // tile start from it (v_accvgpr_write behind the conversion of the tile before), so the epilogue adds nothing
  u32x4e bias_acc[6];
#pragma unroll
  for (int n = 0; n < 6; ++n) bias_acc[n] = u32x4e{0u, 0u, 0u, 0u};
  const int nkt = a.K / BK;
#ifdef TV_DRIP_STAMP
  unsigned long long st_sum[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}, st_t = 0;
  int st_ph = 0, st_n = 0;
#endif

  auto opaque_lane = [&]() __attribute__((always_inline)) {
    int lx = lane;
    asm volatile("" : "+v"(lx));
    return lx;
  };

  // ---- tile state (work-group uniform)
  int cm0, cn0;
  decode(idx, cm0, cn0);
  int nm0 = cm0, nn0 = cn0, pm0 = cm0, pn0 = cn0;
  const bf16_t* Ais = a.A + (int64_t)cm0 * lda;
  const bf16_t* Wis = a.W + (int64_t)cn0 * ldw;
  int tsub = 0;
  int prev_ops = 0;
  bool have_prev = false;

  // K-tile tt - tsub of the tile the copy pointers stand on, into ring buffer PAR (= tt & 1): the descriptor of the copy,
  // issued between the MFMAs of the phase (drip_mma)
  auto copy_A = [&](int tt, auto HT, auto PART) __attribute__((always_inline)) {
    constexpr int h = decltype(HT)::value, par = decltype(PART)::value;
    const bf16_t* b = Ais + (int64_t)((tt - tsub) * BK) + (int64_t)(128 * h) * lda;
    return DripCopy{voffA[0], voffA[1], 0u, uniform_ptr(b), lds0 + (unsigned)(h * A1_OFF + par * A_HALF) + (unsigned)(2 * wave * 1024)};
  };
  auto copy_W = [&](int tt, auto PART) __attribute__((always_inline)) {
    constexpr int par = decltype(PART)::value;
    const bf16_t* b = Wis + (int64_t)((tt - tsub) * BK);
    return DripCopy{voffW[0], voffW[1], voffW[2], uniform_ptr(b), lds0 + (unsigned)(W_OFF + par * W_TILE) + (unsigned)(3 * wave * 1024)};
  };
  auto issue_now = [&](const DripCopy& c, bool three) __attribute__((always_inline)) {       // prologue only
    if (three) dma3(c.base, c.v0, c.v1, c.v2, c.dst);
    else dma2(c.base, c.v0, c.v1, c.dst);
  };

  // ---- the drip: piece j (0..5) of a wave = 8 rows x 128 bytes of its 16 staging rows x 384 bytes: rows 16 w + 8 (j / 3) ..
  // + 7, chunks 8 (j % 3) .. + 7 — lane l takes row l / 8, chunk l % 8, so every 8 lanes store one whole 128-byte line and
  // both the LDS address and the matrix offset are a LANE constant + a work-group-uniform term.  (Staging byte (row,
  // physical chunk p) holds chunk p ^ (row & 7): the XOR stays inside a group of 8 chunks.)
  const unsigned drip_lds = stage0 + (unsigned)((16 * wave + (lane >> 3)) * STAGE_ROW) + (unsigned)((((lane & 7) ^ (lane >> 3)) & 7) << 4);
  const unsigned drip_goff = (unsigned)(((lane >> 3) * ldc + 8 * (lane & 7)) * 2);
  auto drip_read = [&](int j) __attribute__((always_inline)) {
    const int j3 = j >= 3 ? 1 : 0;
    dd = *(lds_u32x4*)(size_t)(drip_lds + (unsigned)(j3 * 8 * STAGE_ROW + (j - 3 * j3) * 128));
  };
  auto gelu8 = [&](u32x4e o) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float x0 = gelu_erf_f(bf16_lo(o[r])), x1 = gelu_erf_f(bf16_hi(o[r]));
      const bf16x2 p2 = {(bf16_t)x0, (bf16_t)x1};
      o[r] = __builtin_bit_cast(unsigned, p2);
    }
    return o;
  };
  // piece j of m-half h of the tile at (m0, n0): one store with a scalar base and the lane's constant offset
  auto drip_store = [&](int h, int j, int m0, int n0) __attribute__((always_inline)) {
    const int j3 = j >= 3 ? 1 : 0;
    const int r0 = 128 * h + 16 * wave + 8 * j3, c0 = 64 * (j - 3 * j3);
    u32x4e o = dd;
    if (EPI == EPI_BIAS_GELU) o = gelu8(o);
    if (da.dbg & 16) { m0 = 0; n0 = 0; }          // dev: every store into the first tile (stays in L2)
    const void* base = uniform_ptr(a.C + (int64_t)(m0 + r0) * ldc + (n0 + c0));
    if (da.dbg & 2) { asm volatile("" ::"v"(o)); return; }
    const int skr = skip_rows(m0), skc = skip_cols(n0);
    // (s_nop: a vector instruction must not overwrite the data registers of a 16-byte store in the next cycle, and the
    // compiler does not know this statement is a store)
    if ((skr | skc) == 0) {
      const int fl = (da.dbg >> 6) & 3;       // dev: store flavours
      if (fl == 0) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(drip_goff), "v"(o), "s"(base) : "memory");
      else if (fl == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(drip_goff), "v"(o), "s"(base) : "memory");
      else if (fl == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(drip_goff), "v"(o), "s"(base) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(drip_goff), "v"(o), "s"(base) : "memory");
    } else {          // shifted edge tile: the rows / columns that belong to the neighbour stay
      const int lx = opaque_lane();
      if (r0 + (lx >> 3) >= skr && c0 + 8 * (lx & 7) >= skc)
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(drip_goff), "v"(o), "s"(base) : "memory");
    }
  };
  // old C of m-half 0 of the tile at (m0, n0) into the staging rows by LDS-DMA: a copy writes 1 KiB of LDS in lane order,
  // so its pieces are 64 consecutive 16-byte slots of the wave's rows (slot s = row s / 24, physical chunk s % 24)
  auto cold_dma3 = [&](int j0, int m0, int n0) __attribute__((always_inline)) {
    unsigned v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int sl = 64 * (j0 + i) + lane;
      const int r16 = sl / 24, row = 16 * wave + r16, ch = (sl - 24 * r16) ^ (row & 7);
      v[i] = (unsigned)((row * ldc + 8 * ch) * 2) - (unsigned)(1024 * i);
    }
    const bf16_t* b = a.C + (int64_t)m0 * ldc + n0;
    dma3(uniform_ptr(b), v[0], v[1], v[2], stage0 + (unsigned)(wave * 6144 + j0 * 1024));
  };

  // ---- lane layout of the accumulators after the v_permlane16_swap of the two m-tiles: 8 consecutive columns
  // 96 wc + 16 n + 8 (kq >> 1) + 0..7 of row 32 wr + 16 (kq & 1) + lc of the m-half
  auto lane_row = [&](int lx) __attribute__((always_inline)) { return 32 * wr + 16 * ((lx >> 4) & 1) + (lx & 15); };
  auto lane_ch0 = [&](int lx) __attribute__((always_inline)) { return 12 * wc + (lx >> 5); };       // + 2 n
  auto stage_lane = [&](int lx, int n) __attribute__((always_inline)) {
    const int row = lane_row(lx);
    return stage0 + (unsigned)(row * STAGE_ROW) + (unsigned)(((lane_ch0(lx) + 2 * n) ^ (row & 7)) << 4);
  };
  // m-half H of the finished accumulators: + bias / + old C, rounded to bf16 (GELU: the pre-activation), into the staging
  // rows (H = 0) or the 24 registers (H = 1)
  auto epi_piece = [&](auto HT, auto NT, int lx) __attribute__((always_inline)) {
    constexpr int H = decltype(HT)::value, n = decltype(NT)::value;
    float x0[4], x1[4], v[8];
    drip_acc_read<H, n>(x0, x1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0[r]), __float_as_uint(x1[r]), false, false);
      v[r] = __uint_as_float(sw[0]);
      v[4 + r] = __uint_as_float(sw[1]);
    }
    if (EPI == EPI_ACCUM) {
      u32x4e c;
      if (H == 0) c = *(lds_u32x4*)(size_t)stage_lane(lx, n);
      else c = drip_pk_read<n>();
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[2 * r] += bf16_lo(c[r]); v[2 * r + 1] += bf16_hi(c[r]); }
    }
    u32x4e o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bf16x2 p2 = {(bf16_t)v[2 * r], (bf16_t)v[2 * r + 1]};
      o[r] = __builtin_bit_cast(unsigned, p2);
    }
    if (H == 0) *(lds_u32x4*)(size_t)stage_lane(lx, n) = o;
    else drip_pk_write<n>(o);
  };
  auto epi_half = [&](auto HT) __attribute__((always_inline)) {
    using std::integral_constant;
    if (da.dbg & 9) return;
    // m-half 1's old C (cold_regs, start of the last K-tile) is older than the 7 copies issued since
    if (EPI == EPI_ACCUM && decltype(HT)::value == 1) GEMM_WAIT_VM(7);
    const int lx = opaque_lane();
    epi_piece(HT, integral_constant<int, 0>{}, lx); epi_piece(HT, integral_constant<int, 1>{}, lx);
    epi_piece(HT, integral_constant<int, 2>{}, lx); epi_piece(HT, integral_constant<int, 3>{}, lx);
    epi_piece(HT, integral_constant<int, 4>{}, lx); epi_piece(HT, integral_constant<int, 5>{}, lx);
  };
  // the accumulators of m-half H <- the bias of the tile that starts next (both m-tiles the same columns)
  auto acc_init = [&](auto HT) __attribute__((always_inline)) {
    constexpr int H = decltype(HT)::value;
    if (EPI == EPI_ACCUM) return;            // (that epilogue's first MFMAs start from zero instead)
    drip_acc_init<H, 0>(bias_acc[0]); drip_acc_init<H, 1>(bias_acc[1]); drip_acc_init<H, 2>(bias_acc[2]);
    drip_acc_init<H, 3>(bias_acc[3]); drip_acc_init<H, 4>(bias_acc[4]); drip_acc_init<H, 5>(bias_acc[5]);
  };
  // the register half into the staging rows
  auto dump = [&]() __attribute__((always_inline)) {
    const int lx = opaque_lane();
    drip_pk_to_lds<0>(stage_lane(lx, 0)); drip_pk_to_lds<1>(stage_lane(lx, 1)); drip_pk_to_lds<2>(stage_lane(lx, 2));
    drip_pk_to_lds<3>(stage_lane(lx, 3)); drip_pk_to_lds<4>(stage_lane(lx, 4)); drip_pk_to_lds<5>(stage_lane(lx, 5));
  };
  // old C of m-half 1 of the tile at (m0, n0) into the 24 registers: loads the compiler does not track, retired by the
  // phase waits (they are older than the copies those leave in flight)
  auto cold_regs = [&](int m0, int n0) __attribute__((always_inline)) {
    const int lx = opaque_lane();
    const bf16_t* p = a.C + (int64_t)(m0 + 128 + lane_row(lx)) * ldc + (n0 + 8 * lane_ch0(lx));
    drip_pk_load(p);
  };
  auto bias_regs = [&](int n0) __attribute__((always_inline)) {
    const int lx = opaque_lane();
    const float* p = (const float*)a.bias + (n0 + 96 * wc + 4 * (lx >> 4));
    asm volatile("global_load_dwordx4 %0, %6, off\n\t"
                 "global_load_dwordx4 %1, %6, off offset:64\n\t"
                 "global_load_dwordx4 %2, %6, off offset:128\n\t"
                 "global_load_dwordx4 %3, %6, off offset:192\n\t"
                 "global_load_dwordx4 %4, %6, off offset:256\n\t"
                 "global_load_dwordx4 %5, %6, off offset:320"
                 : "=&v"(bias_acc[0]), "=&v"(bias_acc[1]), "=&v"(bias_acc[2]), "=&v"(bias_acc[3]), "=&v"(bias_acc[4]), "=&v"(bias_acc[5])
                 : "v"(p) : "memory");
  };

  // ---- what phase 2 of K-tile t of the CURRENT tile does for the previous one (phase 2's load segment is the short one:
  // four fragment reads): K-tiles 0-5 read a piece of m-half 0 from the staging rows, 1-6 store the piece read before,
  // 6 writes the parked half into the rows, 7-12 / 8-13 the same for m-half 1; accumulating epilogue: K-tiles 14 / 15 copy
  // the current tile's old C (m-half 0) into the rows.  Returns the number of vector-memory operations it issued (they
  // sit in the queue behind the copies the phase's wait must leave in flight).
  auto extras = [&](int t) __attribute__((always_inline)) {
    int ops = 0;
    if (da.dbg & 5) return ops;
    if (have_prev) {
      if (!(da.dbg & 32)) {
        if (t >= 1 && t <= 6) { drip_store(0, t - 1, pm0, pn0); ops = 1; }
        else if (t >= 8 && t <= 13) { drip_store(1, t - 8, pm0, pn0); ops = 1; }
      }
      if (da.dbg & 32) {
        // dev: the piece read in THIS phase is stored behind this phase's MFMAs (late_store)
      }
      if (t < 6) drip_read(t);
      else if (t == 6) dump();
      else if (t < 13) drip_read(t - 7);
    }
    if (EPI == EPI_ACCUM) {
      if (t == 14) { cold_dma3(0, cm0, cn0); ops += 3; }
      else if (t == 15) { cold_dma3(3, cm0, cn0); ops += 3; }
    }
    return ops;
  };

  // ---- prologue: W(0) A0(0) A1(0) W(1) A0(1); W(0) and A0(0) have landed behind vmcnt(7)
  {
    using std::integral_constant;
    typedef integral_constant<int, 0> I0;
    typedef integral_constant<int, 1> I1;
    issue_now(copy_W(0, I0{}), true); issue_now(copy_A(0, I0{}, I0{}), false); issue_now(copy_A(0, I1{}, I0{}), false);
    issue_now(copy_W(1, I1{}), true); issue_now(copy_A(1, I0{}, I1{}), false);
  }
  if (EPI != EPI_ACCUM) {
    bias_regs(cn0);
    GEMM_WAIT_VM(0);
    acc_init(std::integral_constant<int, 0>{});
    acc_init(std::integral_constant<int, 1>{});
  }
  GEMM_WAIT_VM(7);
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();          // the second wave of every SIMD runs one barrier behind
#ifdef TV_DRIP_STAMP
  st_t = drip_now();
#endif

  // Copy schedule.  The MFMA segment of phase 1 of K-tile t issues A1(t + 1), that of phase 2 issues W(t + 2) and A0(t + 2)
  // (into the buffers phase 1 of t read: every wave is past those reads, the other group's included, one barrier after
  // phase 1's MFMAs).  The wait in phase 1's load segment retires A1(t) (issued one K-tile ago) and leaves W / A0(t + 1) in
  // flight: 5 operations; the wait in phase 2's retires W / A0(t + 1) and leaves A1(t + 1): 2 operations — plus whatever
  // the load segment itself put into the queue (`more`).  What a wait retires is read one phase later.
  auto wait = [&](int base, int more) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    DRIP_STAMP(0);                   // issue of the reads / extras + the LDS reads' return
    const int n = base + more;
    if (n <= 2) GEMM_WAIT_VM(2);
    else if (n == 3) GEMM_WAIT_VM(3);
    else if (n == 4) GEMM_WAIT_VM(4);
    else if (n == 5) GEMM_WAIT_VM(5);
    else if (n == 6) GEMM_WAIT_VM(6);
    else if (n <= 8) GEMM_WAIT_VM(7);
    else if (n <= 10) GEMM_WAIT_VM(9);
    else GEMM_WAIT_VM(11);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DRIP_STAMP(1);                   // waiting for the copies
    __builtin_amdgcn_s_barrier();
    DRIP_STAMP(2);                   // first barrier
    __builtin_amdgcn_sched_barrier(0);
  };
  // 24 MFMAs of one m-half as ONE statement on the named accumulators, the phase's copies between them (gemm_drip_regs.inc)
  auto mma_half = [&](auto MHT, auto ZT, auto NT, const DripCopy& c0, const DripCopy& c1) __attribute__((always_inline)) {
    drip_mma<decltype(MHT)::value, decltype(ZT)::value, decltype(NT)::value>(af, wfr, c0, c1);
    DRIP_STAMP(3);                   // issue of the 24 MFMAs and the copies
    __builtin_amdgcn_sched_barrier(0);
  };

  auto ktile = [&](int t, auto DBT, auto KINDT) __attribute__((always_inline)) {
    using std::integral_constant;
    typedef integral_constant<int, 0> I0;
    typedef integral_constant<int, 1> I1;
    constexpr int DB = decltype(DBT)::value;
    constexpr int KIND = decltype(KINDT)::value;
    constexpr int TA = DB * A_HALF, TW = DB * W_TILE;
    typedef integral_constant<bool, KIND == KT_FIRST && EPI == EPI_ACCUM> Z;      // (bias epilogues: acc_init)
    typedef integral_constant<bool, KIND == KT_LAST> NP;
    typedef integral_constant<int, DB> PAR;
    typedef integral_constant<int, DB ^ 1> PARX;
    // ---- phase 1: W(t), A0(t)
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      wfr[n][0] = DG_LD(w_b0, TW + n * 2048);
      wfr[n][1] = DG_LD(w_b1, TW + n * 2048);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      af[m][0] = DG_LD(a_b0, TA + m * 2048);
      af[m][1] = DG_LD(a_b1, TA + m * 2048);
    }
    int more = prev_ops;             // what phase 2 of the K-tile before put into the queue is younger than A1(t) too
    if (KIND == KT_LAST && EPI == EPI_ACCUM && !(da.dbg & 1)) { cold_regs(cm0, cn0); more += 6; }
    if (KIND == KT_FIRST && have_prev && grp == 0) { epi_half(I1{}); acc_init(I1{}); }      // the previous tile's m-half 1 (group 1: behind its MFMAs)
    const DripCopy cA1 = copy_A(t + 1, I1{}, PARX{});
    wait(5, more);
    mma_half(I0{}, Z{}, NP{}, cA1, cA1);
    if (KIND == KT_LAST && grp == 1) {
      epi_half(I0{});
      if (EPI != EPI_ACCUM) GEMM_WAIT_VM(7);          // the next tile's bias (phase 2 of the K-tile before) is older than the 7 copies since
      acc_init(I0{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    DRIP_STAMP(4);                   // second barrier (+ the epilogue behind the MFMAs)
#ifdef TV_DRIP_STAMP
    st_ph = 1;
#endif
    // ---- phase 2: A1(t)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      af[m][0] = DG_LD(a_b0, A1_OFF + TA + m * 2048);
      af[m][1] = DG_LD(a_b1, A1_OFF + TA + m * 2048);
    }
    more = 0;
    if (KIND == KT_PRELAST) {
      // every later copy belongs to the next tile (a work-group without one copies its last tile's first K-tiles again:
      // the counted waits stay the same, nothing reads them)
      if (idx + wpx < idx_end) decode(idx + wpx, nm0, nn0);
      Ais = a.A + (int64_t)nm0 * lda;
      Wis = a.W + (int64_t)nn0 * ldw;
      tsub = nkt;
      if (EPI != EPI_ACCUM) { bias_regs(nn0); more = 6; }
    }
    if (KIND == KT_LAST && grp == 0) {
      epi_half(I0{});
      if (EPI != EPI_ACCUM) GEMM_WAIT_VM(7);
      acc_init(I0{});
    }
    more += extras(t);
    prev_ops = more;
    const DripCopy cW = copy_W(t + 2, PAR{}), cA0 = copy_A(t + 2, I0{}, PAR{});
    wait(2, more);
    mma_half(I1{}, Z{}, NP{}, cW, cA0);
    if ((da.dbg & 32) && have_prev && !(da.dbg & 5)) {       // dev: the drip's store behind the MFMAs (and their copies)
      if (t < 6) drip_store(0, t, pm0, pn0);
      else if (t >= 7 && t < 13) drip_store(1, t - 7, pm0, pn0);
    }
    if (KIND == KT_LAST && grp == 1) { epi_half(I1{}); acc_init(I1{}); }
    __builtin_amdgcn_s_barrier();
    DRIP_STAMP(4);
#ifdef TV_DRIP_STAMP
    st_ph = 0;
    ++st_n;
#endif
  };

  for (;;) {
    using std::integral_constant;
    typedef integral_constant<int, 0> I0;
    typedef integral_constant<int, 1> I1;
    {
      int hp = __builtin_amdgcn_readfirstlane((int)have_prev);          // (opaque: the first tile is not peeled off into a second copy of the loop)
      asm volatile("" : "+s"(hp));
      have_prev = hp != 0;
    }
    ktile(0, I0{}, integral_constant<int, KT_FIRST>{});
    for (int t = 1;; t += 2) {
      ktile(t, I1{}, integral_constant<int, KT_MID>{});
      if (t + 1 == nkt - 2) break;
      ktile(t + 1, I0{}, integral_constant<int, KT_MID>{});
    }
    ktile(nkt - 2, I0{}, integral_constant<int, KT_PRELAST>{});
    ktile(nkt - 1, I1{}, integral_constant<int, KT_LAST>{});
    pm0 = cm0; pn0 = cn0;
    have_prev = true;
    idx += wpx;
    if (idx >= idx_end) break;
    cm0 = nm0; cn0 = nn0;
    tsub = 0;
  }
  // ---- the last tile leaves at once: m-half 0 from the staging rows, m-half 1 from the registers
  if (grp == 0) {
    __builtin_amdgcn_s_barrier();          // barrier counts match again
    epi_half(std::integral_constant<int, 1>{});
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (!(da.dbg & 1)) {
    for (int j = 0; j < 6; ++j) {
      drip_read(j);
      drip_store(0, j, pm0, pn0);
    }
    const int lx = opaque_lane();
    const int rt = 128 + lane_row(lx);
    auto last_piece = [&](auto NT) __attribute__((always_inline)) {
      constexpr int n = decltype(NT)::value;
      u32x4e o = drip_pk_read<n>();
      if (EPI == EPI_BIAS_GELU) o = gelu8(o);
      const int col = 8 * (lane_ch0(lx) + 2 * n);
      bf16_t* cp = a.C + (int64_t)(pm0 + rt) * ldc + (pn0 + col);
      if (da.dbg & 2) asm volatile("" ::"v"(o));
      else if (rt >= skip_rows(pm0) && col >= skip_cols(pn0)) *(u32x4e*)cp = o;
    };
    using std::integral_constant;
    last_piece(integral_constant<int, 0>{}); last_piece(integral_constant<int, 1>{}); last_piece(integral_constant<int, 2>{});
    last_piece(integral_constant<int, 3>{}); last_piece(integral_constant<int, 4>{}); last_piece(integral_constant<int, 5>{});
  }
  GEMM_WAIT_VM(0);        // the copies issued for a tile that does not exist must not land in another work-group's LDS
#ifdef TV_DRIP_STAMP
  if (da.stamps && blockIdx.x < 8 && lane == 0) {
    unsigned long long* o = da.stamps + ((size_t)blockIdx.x * 8 + wave) * 12;
    for (int ph = 0; ph < 2; ++ph) {
      for (int i = 0; i < 5; ++i) o[ph * 6 + i] = st_sum[ph][i];
      o[ph * 6 + 5] = (unsigned long long)st_n;
    }
  }
#endif
}

template <int EPI>
int launch(const DArgs& da, int grid, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute((const void*)gemm_drip_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
  if (e != hipSuccess) {
    tv_set_error("gemm (drip): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return TV_ERR_LAUNCH;
  }
  gemm_drip_kernel<EPI><<<dim3((unsigned)grid), 512, LDS_TOTAL, st>>>(da);
  TV_LAUNCH_CHECK();
}

int g_drip_mode = -1;      // -1 automatic, 0 never, 1 wherever the shape allows

}  // namespace

bool drip_takes(const GemmArgs& a, int epilogue, int grid) {
  static const int off = [] { const char* e = getenv("TV_GEMM_DRIP"); return e && atoi(e) == 0; }();
  if (g_drip_mode == 0 || (g_drip_mode < 0 && off)) return false;
  const int nkt = a.K / BK;
  if (a.K % (2 * BK) || nkt < MIN_KT) return false;
  if (a.M < BM || a.N < DBN || a.N % 8 || a.ldc % 8 || ((uintptr_t)a.C & 15)) return false;
  if (a.lda < 64 || a.ldw < 64) return false;
  if (epilogue != EPI_ACCUM && (a.bias == nullptr || !a.bias_f32)) return false;
  if (256 * a.ldc * 2 >= (1ll << 31)) return false;
  const int64_t tiles = (int64_t)a.tiles_m * ((a.N + DBN - 1) / DBN);
  if (g_drip_mode < 0 && tiles < 4ll * grid) return false;
  return true;
}

int launch_drip(const GemmArgs& a0, int epilogue, int grid, hipStream_t st) {
  DArgs da;
  da.g = a0;
  da.tiles_n = (a0.N + DBN - 1) / DBN;
  da.ntiles = a0.tiles_m * da.tiles_n;
  {
    static const int gm_env = [] { const char* e = getenv("TV_GEMM_DRIP_GROUP_M"); return e ? atoi(e) : 0; }();     // dev tool
    da.g.group_m = gm_env > 0 ? gm_env : 4;      // measured (2 / 4 / 8 / 16) at N 1 152 and 3 456: 4
    if (da.g.group_m > a0.tiles_m) da.g.group_m = a0.tiles_m;
  }
  static const int dbg_env = [] { const char* e = getenv("TV_GEMM_DBG"); return e ? atoi(e) : 0; }();
  da.dbg = dbg_env;
#ifdef TV_DRIP_STAMP
  static unsigned long long* stamp_buf = [] { void* p = nullptr; (void)hipMalloc(&p, 8 * 8 * 12 * 8); return (unsigned long long*)p; }();
  da.stamps = stamp_buf;
  (void)hipMemsetAsync(stamp_buf, 0, 8 * 8 * 12 * 8, st);
#endif
  int rc;
  switch (epilogue) {
    case EPI_BIAS: rc = launch<EPI_BIAS>(da, grid, st); break;
    case EPI_BIAS_GELU: rc = launch<EPI_BIAS_GELU>(da, grid, st); break;
    default: rc = launch<EPI_ACCUM>(da, grid, st); break;
  }
#ifdef TV_DRIP_STAMP
  {
    static int printed = 0;
    static const int want = [] { const char* e = getenv("TV_DRIP_STAMP_PRINT"); return e ? atoi(e) : 2; }();
    if (printed < want) {
      ++printed;
      unsigned long long h[8 * 8 * 12];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(h, stamp_buf, sizeof(h), hipMemcpyDeviceToHost);
      static const char* nm[5] = {"issue+lds", "copy wait", "barrier 1", "mfma", "barrier 2"};
      for (int wv = 0; wv < 8; wv += 4) {
        fprintf(stderr, "[drip stamps] epi %d N %d K %d, work-group 0 wave %d, cycles per phase over %llu K-tiles:", epilogue, a0.N, a0.K, wv,
                h[wv * 12 + 5]);
        for (int ph = 0; ph < 2; ++ph) {
          double tot = 0;
          fprintf(stderr, "\n    phase %d:", ph + 1);
          for (int i = 0; i < 5; ++i) {
            const double v = (double)h[wv * 12 + ph * 6 + i] / (double)(h[wv * 12 + 5] ? h[wv * 12 + 5] : 1);
            tot += v;
            fprintf(stderr, " %s %.0f", nm[i], v);
          }
          fprintf(stderr, " | total %.0f", tot);
        }
        fprintf(stderr, "\n");
      }
    }
  }
#endif
  return rc;
}

void set_drip(int mode) { g_drip_mode = mode; }

}  // namespace tvgemm

extern "C" void tv_gemm_set_drip(int mode) { tvgemm::set_drip(mode); }
