// Persistent bf16 GEMM for the ViT linears (timm block via base_vision.py:146-170,274-278, InternVideo2
// Attention / Mlp vit_scale_clean.py:188-320):  C[M][N] = epilogue( A[M][K] . W[N][K]^T ), epilogues as in gemm.hip.
//
// Why a second kernel.  Three of the four linears of a SigLIP block have K = 1 152: 18 K-tiles of 64.  With one work-group
// per output tile every tile pays, outside its K loop, the pipeline fill, the conversion of 256 x 256 accumulators and —
// the largest part — 128 KB of C stored with the matrix pipe idle (a CU writes at ~25 GB/s: 5 us against 26 us of K loop):
// 24 - 32 % of a launch (DESIGN.md section 5, K sweep).  Here a work-group stays on its CU and walks a list of tiles:
//   * the copy pipeline (one half-tile per phase, six phases ahead, gemm.hip's schedule) never stops: the last two
//     K-tiles of a tile already issue the first copies of the next one;
//   * the epilogue is taken apart by QUADRANT.  A wave's 128 x 64 outputs are four quadrants (mh, nh) of 32 accumulator
//     registers; in the last K-tile they become final one phase after the other (Q00, Q01, Q11, Q10), and each is
//     converted and stored in the LOAD segment of the following phase — the segment in which the wave's SIMD partner
//     (the other wave group, one barrier apart) owns the matrix pipe.  The next tile's first K-tile starts every
//     accumulator from a zero C operand, so no register is cleared and no second accumulator set is needed;
//   * the stores enter the same in-order vmcnt queue as the copies.  Every counted wait of the nine phases behind a
//     store burst allows for exactly the stores that are younger than the half-tile it needs (14 / 18 / 22 / 26 ...), so
//     a store has five phases (~2 500 cycles) to be acknowledged before it can hold up a copy;
//   * bias: 1 KiB per tile by LDS-DMA into one of two slots (every wave issues the same copy: the counts stay uniform),
//     read back in the epilogue;  accumulate (C += A W^T): the old C enters THROUGH THE MATRIX PIPE — 16 x 32 pieces of C
//     are loaded in MFMA operand layout six phases ahead and multiplied by a constant 0/1 selector fragment
//     (acc[n][m] += sum_k I[n][k] C[m][k]: exact, a bf16 times 1.0 added in fp32), eight half-quadrant steps spread over
//     K-tiles 1 - 16, so the epilogue of all three variants is the same convert-and-store.
//   * tiles: ids that share an XCD (blockIdx % 8, observed dispatch order; speed only) own one contiguous eighth of the
//     banded tile order (bands of group_m m-tiles, m fastest), and the 32 work-groups of an XCD walk it side by side: they
//     share every A / W k-slab through their L2 exactly like the per-tile kernel's dispatch order did.
//   * edge tiles are SHIFTED back inside the matrix (m0 = M - 256, n0 = N - 256) instead of clamped: every copy address
//     is tile-independent + a scalar base, and the stores of rows / columns that belong to the neighbour are masked.
#include "gemm_common.hpp"

namespace tvgemm {
namespace {

constexpr int BIAS_SLOT = 1024;
constexpr int RESID_BYTES = 8 * 2048;               // accumulate: two 1 KiB pieces of the old C per wave
constexpr int LDS_BYTES = RING_BYTES + 2 * BIAS_SLOT + RESID_BYTES;
enum { KT_FIRST = 0, KT_MID = 1, KT_LAST = 2 };

struct PArgs {
  GemmArgs g;
  int ntiles;
  int dbg;       // dev switches (TV_GEMM_DBG): 1 = no epilogue, 2 = epilogue arithmetic without stores, 4 = plain wait counts
};

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4e __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;

template <int EPI>
__global__ __launch_bounds__(512) void gemm_persist_kernel(PArgs pa) {
  const GemmArgs& a = pa.g;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int lc = lane & 15, kq = lane >> 4;

  // ---- this work-group's tiles: positions slot, slot + wpx, ... of the XCD's eighth of the banded order
  const int wpx = (int)gridDim.x >> 3;
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int chunk = (pa.ntiles + 7) >> 3;
  int idx = xcd * chunk + slot;
  const int idx_end = min(pa.ntiles, (xcd + 1) * chunk);
  if (idx >= idx_end) return;

  // Work-group uniform state is kept SMALL (two ints per tile generation, two running copy pointers): the kernel sits at
  // the 102-SGPR limit, and every uniform value the scalar file cannot hold becomes a vector register the main loop
  // cannot spare either.  Edge tiles are shifted back inside the matrix, so the rows / columns a tile must not store
  // follow from its origin alone: skip = (-origin) mod 256.
  const int lda = (int)a.lda, ldw = (int)a.ldw, ldc = (int)a.ldc;
  const int per_band = a.group_m * a.tiles_n;
  auto decode = [&](int i, int& m0, int& n0) __attribute__((always_inline)) {
    const int band = i / per_band, in_band = i - band * per_band;
    const int rows = min(a.group_m, a.tiles_m - band * a.group_m);
    const int tn = in_band / rows, tm = band * a.group_m + (in_band - tn * rows);
    m0 = __builtin_amdgcn_readfirstlane(min(tm * BM, a.M - BM));
    n0 = __builtin_amdgcn_readfirstlane(min(tn * BN, a.N - BN));
  };

  const unsigned lds0 = lds_addr_of(smem_raw);
  const unsigned bias_lds0 = lds0 + RING_BYTES;
  const unsigned resid_lds = bias_lds0 + 2 * BIAS_SLOT + (unsigned)(wave * 2048);

  // ---- staging: half-tile h of operand X = 16 pieces of 1 KiB (8 rows x 128 B); wave w copies pieces 2w, 2w + 1 with one
  // M0 set-up (the second piece through the instruction offset, which moves source and destination alike: its source
  // offset is passed less 1 024).  LDS byte (piece i, lane l) = row 8 i + l / 8, physical chunk l % 8, holding source
  // chunk (l % 8) ^ f(row), f(row) = (row >> 1) & 7.
  unsigned voffA[2], voffW[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 16 * wave + 8 * j + (lane >> 3);
    const int ch = (lane & 7) ^ ((row >> 1) & 7);
    voffA[j] = (unsigned)((row * lda + ch * 8) * 2) - (unsigned)(1024 * j);
    voffW[j] = (unsigned)((row * ldw + ch * 8) * 2) - (unsigned)(1024 * j);
  }
  const unsigned piece0 = (unsigned)(2 * wave * 1024);
  auto dma2 = [&](const void* sp, unsigned v0, unsigned v1, unsigned dst) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "s"(sp), "s"(dst) : "memory");
  };

  // ---- fragment reads (as gemm.hip)
  const unsigned frag_lo = (unsigned)(lc * 128 + ((kq ^ (lc >> 1)) << 4));
  const unsigned a_b0 = lds0 + (unsigned)(wr * 64 * 128) + frag_lo, a_b1 = lds0 + (unsigned)(wr * 64 * 128) + (frag_lo ^ 64u);
  const unsigned w_b0 = lds0 + (unsigned)(wc * 32 * 128) + frag_lo, w_b1 = lds0 + (unsigned)(wc * 32 * 128) + (frag_lo ^ 64u);
#define PG_LD(base, off) (*(lds_bf16x8*)(size_t)((base) + (unsigned)(off)))

  f32x4 acc[2][2][4][2];
  bf16x8 af[4][2];
  bf16x8 wf0[2][2][2], wf1[2][2];

  const int nkt = a.K / BK;

  // ---- epilogue geometry: after the v_permlane16_swap of two m-tiles a lane holds 8 consecutive columns
  // 16 n + 8 (kq >> 1) + 0..7 of row 16 (mp + (kq & 1)) + lc of the wave's quadrant.  The lane terms are rebuilt from
  // the lane id where they are used (once per quadrant and tile): hoisted out of the tile loop they would be registers
  // the main loop has to carry.
  auto opaque_lane = [&]() __attribute__((always_inline)) {
    int lx = lane;
    asm volatile("" : "+v"(lx));
    return lx;
  };
  const bool has_bias = EPI != EPI_ACCUM && a.bias != nullptr;

  // ---- tile state (work-group uniform): origins of the current / next / previous tile, the two pointers the copies are
  // issued from (they move on to the next tile in K-tile nkt - 2, two K-tiles before the MFMAs do) and the K-tile index
  // that tile's K-tile 0 has in the current tile's count (0, or nkt once the pointers have moved on)
  int cm0, cn0;
  decode(idx, cm0, cn0);
  int nm0 = cm0, nn0 = cn0, pm0 = cm0, pn0 = cn0;
  const bf16_t* Ais = a.A + (int64_t)cm0 * lda;
  const bf16_t* Wis = a.W + (int64_t)cn0 * ldw;
  int tsub = 0;
  int jpar = 0;            // parity of the tile counter: bias slot of the current tile
  bool have_prev = false;
  auto c_base = [&](int m0, int n0) __attribute__((always_inline)) { return a.C + (int64_t)m0 * ldc + n0; };
  auto is_edge = [&](int m0, int n0) __attribute__((always_inline)) { return ((m0 | n0) & (BM - 1)) != 0; };

  // item (tt, k): k 0: W0, 1: A0, 2: W1, 3: A1 of K-tile tt - tsub of the tile the copy pointers stand on
  auto issue_item = [&](int tt, auto KT) __attribute__((always_inline)) {
    constexpr int k = decltype(KT)::value;
    constexpr int h = k >> 1;
    const int tk = tt - tsub;
    const unsigned dst = lds0 + (unsigned)((tt & 1) * TILE_BYTES + (2 * ((k & 1) ? 0 : 1) + h) * HALF_BYTES) + piece0;
    if (k & 1) {
      const bf16_t* b = Ais + (int64_t)(tk * BK) + (int64_t)(128 * h) * lda;
      dma2(uniform_ptr(b), voffA[0], voffA[1], dst);
    } else {
      const bf16_t* b = Wis + (int64_t)(tk * BK) + (int64_t)(128 * h) * ldw;
      dma2(uniform_ptr(b), voffW[0], voffW[1], dst);
    }
  };

  // ---- epilogue of one quadrant of the tile at (m0, n0).  Two v_permlane16_swap levels: the first joins two m-tiles so
  // that a lane holds 8 consecutive columns (16 n + 8 (kq >> 1) + 0..7 of row 16 (mp + (kq & 1)) + lc), the second, on
  // the packed bf16 pairs, joins the two n-tiles: register set 0 then holds m-tile mp, set 1 m-tile mp + 1, lane (lc, kq)
  // columns 16 (kq & 1) + 8 (kq >> 1) + 0..7 of row lc — every store instruction writes 16 rows x 64 contiguous bytes
  // (the wave's whole share of those rows) instead of 32 rows x 32 bytes.
  auto epi_quadrant = [&](auto MHT, auto NHT, int m0, int n0, unsigned bias_lds) __attribute__((always_inline)) {
    constexpr int MH = decltype(MHT)::value, NH = decltype(NHT)::value;
    bf16_t* Cb = c_base(m0, n0);
    const int skr = (-m0) & (BM - 1), skc = (-n0) & (BN - 1);
    const int lx = opaque_lane();
    const int lcx = lx & 15, kqx = lx >> 4;
    u32x4e o[2][2];              // [mp / 2][n]
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col_t = 128 * NH + 16 * n + 32 * wc + 8 * (kqx >> 1);
      float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (EPI != EPI_ACCUM && has_bias) {
        if (a.bias_f32) {
          const f32x4 b0 = *(lds_f32x4*)(size_t)(bias_lds + (unsigned)(col_t * 4));
          const f32x4 b1 = *(lds_f32x4*)(size_t)(bias_lds + (unsigned)(col_t * 4 + 16));
#pragma unroll
          for (int r = 0; r < 4; ++r) { bv[r] = b0[r]; bv[4 + r] = b1[r]; }
        } else {
          const bf16x8 b8 = *(lds_bf16x8*)(size_t)(bias_lds + (unsigned)(col_t * 2));
#pragma unroll
          for (int r = 0; r < 8; ++r) bv[r] = (float)b8[r];
        }
      }
#pragma unroll
      for (int mp = 0; mp < 4; mp += 2) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[MH][NH][mp][n][r]),
                                                           __float_as_uint(acc[MH][NH][mp + 1][n][r]), false, false);
          v[r] = __uint_as_float(sw[0]);
          v[4 + r] = __uint_as_float(sw[1]);
        }
        if (EPI != EPI_ACCUM) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += bv[r];
        }
        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = gelu_erf_f((float)(bf16_t)v[r]);       // the GEMM's own bf16 rounding first
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bf16x2 pk = {(bf16_t)v[2 * r], (bf16_t)v[2 * r + 1]};
          o[mp >> 1][n][r] = __builtin_bit_cast(unsigned, pk);
        }
      }
    }
    if (pa.dbg & 1) return;      // dev: no stores (the K loop alone)
    if (pa.dbg & 8) {            // dev: the first store shape (32 rows x 32 bytes per instruction), for A/B runs
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int mq = 0; mq < 2; ++mq) {
          const int lane_r = 64 * wr + 16 * (kqx & 1) + lcx, lane_c = 32 * wc + 8 * (kqx >> 1);
          unsigned char* cp = (unsigned char*)(Cb + (int64_t)(128 * MH + 32 * mq) * ldc + (128 * NH + 16 * n)) +
                              (unsigned)((lane_r * ldc + lane_c) * 2);
          *(u32x4e*)cp = o[mq][n];
        }
      return;
    }
    const int col_t = 128 * NH + 32 * wc + 16 * (kqx & 1) + 8 * (kqx >> 1);
    const unsigned lane_coff = (unsigned)(((64 * wr + lcx) * ldc + 32 * wc + 16 * (kqx & 1) + 8 * (kqx >> 1)) * 2);
#pragma unroll
    for (int mq = 0; mq < 2; ++mq) {
      u32x4e s0, s1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const auto sw = __builtin_amdgcn_permlane16_swap(o[mq][0][r], o[mq][1][r], false, false);
        s0[r] = sw[0];
        s1[r] = sw[1];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row_t = 128 * MH + 64 * wr + 16 * (2 * mq + i) + lcx;
        unsigned char* cp = (unsigned char*)(Cb + (int64_t)(128 * MH + 16 * (2 * mq + i)) * ldc + 128 * NH) + lane_coff;
        if (pa.dbg & 2) {            // dev: the epilogue's arithmetic without its stores
          if (i) asm volatile("" ::"v"(s1)); else asm volatile("" ::"v"(s0));
        } else if (!is_edge(m0, n0) || (row_t >= skr && col_t >= skc)) *(u32x4e*)cp = i ? s1 : s0;
      }
    }
  };

  // ---- prologue: items 0 .. 6 of the first tile; W0(0) and A0(0) have landed behind vmcnt(10)
  {
    using std::integral_constant;
    issue_item(0, integral_constant<int, 0>{}); issue_item(0, integral_constant<int, 1>{});
    issue_item(0, integral_constant<int, 2>{}); issue_item(0, integral_constant<int, 3>{});
    issue_item(1, integral_constant<int, 0>{}); issue_item(1, integral_constant<int, 1>{});
    issue_item(1, integral_constant<int, 2>{});
  }
  GEMM_WAIT_VM(10);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    wf0[0][n][0] = PG_LD(w_b0, 2 * HALF_BYTES + n * 2048);
    wf0[0][n][1] = PG_LD(w_b1, 2 * HALF_BYTES + n * 2048);
  }
  if (wr == 1) __builtin_amdgcn_s_barrier();          // the second wave of every SIMD runs one barrier behind

  // One phase = [fragment reads + one copy + extras + counted wait] barrier [16 MFMAs] barrier
  auto mma_quadrant = [&](auto MHT, auto NHT, auto ZT, const bf16x8 (&wf)[2][2]) __attribute__((always_inline)) {
    constexpr int mh = decltype(MHT)::value, nh = decltype(NHT)::value;
    constexpr bool ZERO = decltype(ZT)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[mh][nh][m][n] = mfma16(wf[n][ks], af[m][ks], (ZERO && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mh][nh][m][n]);
  };
  auto mma_end = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // accumulate: the old C of half a quadrant — (MH, NH), m-tiles 2 H and 2 H + 1, 32 columns, in MFMA operand layout
  // (lane: lc = row, kq = 8-column chunk) — is copied into 2 KiB of the wave's own LDS six phases before resid_mma
  // reads it back and multiplies it into the accumulators (through LDS, not through registers: a register with a load
  // in flight that the compiler does not know about is copied or re-assigned at the joins of the K loop's branches).
  // Eight such steps per tile, each in a phase that works on its quadrant anyway (static accumulator registers behind
  // a uniform branch on the K-tile index), seven phases apart so that one buffer serves all of them:
  //   use  (K-tile, phase): (3,1) Q00 h0  (4,4) Q10 h0  (6,3) Q11 h0  (8,2) Q01 h0  (10,1) Q00 h1  (11,4) Q10 h1  (13,3) Q11 h1  (15,2) Q01 h1
  //   load six phases earlier: (1,3) (3,2) (5,1) (6,4) (8,3) (10,2) (12,1) (13,4)
  auto resid_load = [&](auto MHT, auto NHT, auto HT) __attribute__((always_inline)) {
    constexpr int MH = decltype(MHT)::value, NH = decltype(NHT)::value, H = decltype(HT)::value;
    const int lx = opaque_lane();
    const unsigned lane_roff = (unsigned)(((64 * wr + (lx & 15)) * ldc + 32 * wc + 8 * (lx >> 4)) * 2);
    const unsigned char* p0 = (const unsigned char*)(c_base(cm0, cn0) + (int64_t)(128 * MH + 32 * H) * ldc + 128 * NH);
    const unsigned char* p1 = p0 + (int64_t)16 * ldc * 2 - 1024;       // second piece: the instruction offset moves both sides
    unsigned keep;
    const void *q0 = uniform_ptr(p0), *q1 = uniform_ptr(p1);
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)resid_lds);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_roff), "s"(q0), "s"(q1), "s"(dst) : "memory");
  };
  auto resid_mma = [&](auto MHT, auto NHT, auto HT) __attribute__((always_inline)) {
    constexpr int MH = decltype(MHT)::value, NH = decltype(NHT)::value, H = decltype(HT)::value;
    // the two pieces, as the copies laid them down: 16 bytes per lane
    const int lx = opaque_lane();
    u32x4e r0 = *(__attribute__((address_space(3))) const u32x4e*)(size_t)(resid_lds + (unsigned)(lx * 16));
    u32x4e r1 = *(__attribute__((address_space(3))) const u32x4e*)(size_t)(resid_lds + (unsigned)(lx * 16 + 1024));
    // selector fragments: W-side row n (= lc), k-chunk kq: 1.0 at k = lc + 16 sel
    // (rebuilt from the lane id at every step: eight registers the main loop does not have to carry)
    u32x4e z0 = {0, 0, 0, 0}, z1 = {0, 0, 0, 0};
    {
      const int lc = lx & 15, kq = lx >> 4;
      const unsigned one = (lc & 1) ? 0x3f800000u : 0x00003f80u;
      const int d = (lc & 7) >> 1;
      const bool on0 = kq == (lc >> 3), on1 = kq == 2 + (lc >> 3);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        z0[i] = (on0 && d == i) ? one : 0u;
        z1[i] = (on1 && d == i) ? one : 0u;
      }
    }
    // inline asm with the accumulator tied in place (the builtin behind a branch made the allocator move whole accumulator
    // tiles between registers at the joins); s_nop: the operands were written by vector / LDS instructions just above,
    // and the hazard recogniser does not look inside an asm
    asm volatile("s_nop 4" : "+v"(r0), "+v"(r1), "+v"(z0), "+v"(z1));
    f32x4 &c00 = acc[MH][NH][2 * H][0], &c01 = acc[MH][NH][2 * H][1];
    f32x4 &c10 = acc[MH][NH][2 * H + 1][0], &c11 = acc[MH][NH][2 * H + 1][1];
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c00) : "v"(z0), "v"(r0));
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c01) : "v"(z1), "v"(r0));
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c10) : "v"(z0), "v"(r1));
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c11) : "v"(z1), "v"(r1));
    // the compiler may copy these tiles with vector moves at the branch's join; an MFMA result read by a vector
    // instruction needs wait states that only software provides, and for an asm nobody but us inserts them
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c00), "+v"(c01), "+v"(c10), "+v"(c11));
  };

  // DB: K-tile buffer (t & 1); KIND: first / middle / last K-tile of a tile
  auto ktile = [&](int t, auto DBT, auto KINDT) __attribute__((always_inline)) {
    using std::integral_constant;
    constexpr int DB = decltype(DBT)::value;
    constexpr int KIND = decltype(KINDT)::value;
    constexpr int TB = DB * TILE_BYTES;
    constexpr bool Z = KIND == KT_FIRST;
    // phase p issues item g + 6 = (t + 1 + (p + 2) / 4, (p + 2) % 4) and waits until item g + 1 has landed; the ops that
    // may stay in flight are the five younger half-tiles (10) plus the epilogue stores issued in phases g - 5 .. g
    auto copy = [&](auto PT_) __attribute__((always_inline)) {
      constexpr int p = decltype(PT_)::value;
      issue_item(t + 1 + (p + 2) / 4, integral_constant<int, (p + 2) % 4>{});
    };
    auto wait = [&](auto PT_) __attribute__((always_inline)) {
      constexpr int p = decltype(PT_)::value;
      __builtin_amdgcn_sched_barrier(0);
      // (an edge tile's masked stores may be skipped by whole waves — s_cbranch_execz — so their number is unknown: the
      // plain count, which is never too lenient, serves there)
      if (KIND == KT_LAST && p > 1) {
        if (is_edge(cm0, cn0) || (pa.dbg & 7)) GEMM_WAIT_VM(10);
        else if (p == 2) GEMM_WAIT_VM(14);
        else if (p == 3) GEMM_WAIT_VM(18);
        else GEMM_WAIT_VM(22);
      } else if (KIND == KT_FIRST) {
        if (have_prev && !is_edge(pm0, pn0) && !(pa.dbg & 7)) {
          if (p == 4) GEMM_WAIT_VM(22);
          else GEMM_WAIT_VM(26);
        } else GEMM_WAIT_VM(10);
      } else GEMM_WAIT_VM(10);
      __builtin_amdgcn_s_barrier();
    };
    // ---- phase 1: A0(t)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      af[m][0] = PG_LD(a_b0, TB + m * 2048);
      af[m][1] = PG_LD(a_b1, TB + m * 2048);
    }
    copy(integral_constant<int, 1>{});
    if (KIND == KT_MID && t == nkt - 2) {
      // every later copy belongs to the next tile (a work-group without one copies its last tile's first K-tiles again:
      // the counted waits stay the same, nothing reads them)
      if (idx + wpx < idx_end) decode(idx + wpx, nm0, nn0);
      Ais = a.A + (int64_t)nm0 * lda;
      Wis = a.W + (int64_t)nn0 * ldw;
      tsub = nkt;
    }
    if (KIND == KT_FIRST) {
      if (have_prev)
        epi_quadrant(integral_constant<int, 1>{}, integral_constant<int, 0>{}, pm0, pn0,
                     bias_lds0 + (unsigned)((jpar ^ 1) * BIAS_SLOT));
    }
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 5) resid_load(integral_constant<int, 1>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{});
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 12) resid_load(integral_constant<int, 1>{}, integral_constant<int, 1>{}, integral_constant<int, 1>{});
    wait(integral_constant<int, 1>{});
    mma_quadrant(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<bool, Z>{}, wf0[DB]);
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 3) resid_mma(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{});
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 10) resid_mma(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, 1>{});
    mma_end();
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: W1(t)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      wf1[n][0] = PG_LD(w_b0, TB + 3 * HALF_BYTES + n * 2048);
      wf1[n][1] = PG_LD(w_b1, TB + 3 * HALF_BYTES + n * 2048);
    }
    copy(integral_constant<int, 2>{});
    if (KIND == KT_FIRST && has_bias) {
      // the current tile's bias row: 1 KiB (fp32) / 512 B (bf16) into slot jpar, the same copy from every wave
      const unsigned char* bp = (const unsigned char*)a.bias + (int64_t)cn0 * (a.bias_f32 ? 4 : 2);
      const unsigned bl = a.bias_f32 ? (unsigned)(lane * 16) : (unsigned)((lane & 31) * 16);
      glds16(uniform_ptr(bp), bl, (unsigned)__builtin_amdgcn_readfirstlane((int)(bias_lds0 + (unsigned)(jpar * BIAS_SLOT))));
    }
    if (KIND == KT_LAST)
      epi_quadrant(integral_constant<int, 0>{}, integral_constant<int, 0>{}, cm0, cn0,
                   bias_lds0 + (unsigned)(jpar * BIAS_SLOT));
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 3) resid_load(integral_constant<int, 1>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{});
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 10) resid_load(integral_constant<int, 1>{}, integral_constant<int, 0>{}, integral_constant<int, 1>{});
    wait(integral_constant<int, 2>{});
    mma_quadrant(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<bool, Z>{}, wf1);
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 8) resid_mma(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{});
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 15) resid_mma(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 1>{});
    mma_end();
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: A1(t)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      af[m][0] = PG_LD(a_b0, TB + HALF_BYTES + m * 2048);
      af[m][1] = PG_LD(a_b1, TB + HALF_BYTES + m * 2048);
    }
    copy(integral_constant<int, 3>{});
    if (KIND == KT_LAST)
      epi_quadrant(integral_constant<int, 0>{}, integral_constant<int, 1>{}, cm0, cn0,
                   bias_lds0 + (unsigned)(jpar * BIAS_SLOT));
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 1) resid_load(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{});
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 8) resid_load(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, 1>{});
    wait(integral_constant<int, 3>{});
    mma_quadrant(integral_constant<int, 1>{}, integral_constant<int, 1>{}, integral_constant<bool, Z>{}, wf1);
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 6) resid_mma(integral_constant<int, 1>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{});
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 13) resid_mma(integral_constant<int, 1>{}, integral_constant<int, 1>{}, integral_constant<int, 1>{});
    mma_end();
    __builtin_amdgcn_s_barrier();
    // ---- phase 4: W0(t + 1) into the other register set
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      wf0[DB ^ 1][n][0] = PG_LD(w_b0, (TILE_BYTES - TB) + 2 * HALF_BYTES + n * 2048);
      wf0[DB ^ 1][n][1] = PG_LD(w_b1, (TILE_BYTES - TB) + 2 * HALF_BYTES + n * 2048);
    }
    copy(integral_constant<int, 4>{});
    if (KIND == KT_LAST)
      epi_quadrant(integral_constant<int, 1>{}, integral_constant<int, 1>{}, cm0, cn0,
                   bias_lds0 + (unsigned)(jpar * BIAS_SLOT));
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 6) resid_load(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{});
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 13) resid_load(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 1>{});
    wait(integral_constant<int, 4>{});
    mma_quadrant(integral_constant<int, 1>{}, integral_constant<int, 0>{}, integral_constant<bool, Z>{}, wf0[DB]);
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 4) resid_mma(integral_constant<int, 1>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{});
    if (EPI == EPI_ACCUM && KIND == KT_MID && t == 11) resid_mma(integral_constant<int, 1>{}, integral_constant<int, 0>{}, integral_constant<int, 1>{});
    mma_end();
    __builtin_amdgcn_s_barrier();
  };

  for (;;) {
    using std::integral_constant;
    ktile(0, integral_constant<int, 0>{}, integral_constant<int, KT_FIRST>{});
    for (int t = 1; t < nkt - 1; t += 2) {
      ktile(t, integral_constant<int, 1>{}, integral_constant<int, KT_MID>{});
      ktile(t + 1, integral_constant<int, 0>{}, integral_constant<int, KT_MID>{});
    }
    ktile(nkt - 1, integral_constant<int, 1>{}, integral_constant<int, KT_LAST>{});
    idx += wpx;
    if (idx >= idx_end) break;
    // next tile becomes current
    pm0 = cm0; pn0 = cn0;
    cm0 = nm0; cn0 = nn0;
    tsub = 0;
    jpar ^= 1;
    have_prev = true;
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();          // barrier counts match again
  epi_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, cm0, cn0,
               bias_lds0 + (unsigned)(jpar * BIAS_SLOT));
  GEMM_WAIT_VM(0);        // the copies issued for a tile that does not exist must not land in another work-group's LDS
}

template <int EPI>
int launch(const PArgs& pa, int grid, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute((const void*)gemm_persist_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) {
    tv_set_error("gemm (persistent): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return TV_ERR_LAUNCH;
  }
  gemm_persist_kernel<EPI><<<dim3((unsigned)grid), 512, LDS_BYTES, st>>>(pa);
  TV_LAUNCH_CHECK();
}

int cu_count() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
    return cus;
  }();
  return n;
}

// tv_gemm_set_persist: mode -1 automatic (default; TV_GEMM_PERSIST=0 in the environment switches it off), 0 never, 1
// wherever the shape allows it (tests: a few tiles per work-group on a small grid); grid 0 = one work-group per CU.
int g_mode = -1, g_grid = 0;

int grid_size() {
  static const int grid_env = [] { const char* e = getenv("TV_GEMM_PERSIST_GRID"); return e ? atoi(e) : 0; }();   // dev tool
  int grid = g_grid > 0 ? g_grid : (grid_env > 0 ? grid_env : cu_count());
  grid = (grid / 8) * 8;
  return grid < 8 ? 8 : grid;
}

}  // namespace

// Shapes the persistent kernel takes: whole 16-byte pieces everywhere, at least one full tile in both directions (edge
// tiles are shifted back, not clamped), 18 K-tiles for the accumulate variant (its eight C steps end in K-tile 15), and
// enough tiles that every work-group has a few (the per-tile kernel is as good below that).
bool persist_takes(const GemmArgs& a, int epilogue) {
  static const int off = [] { const char* e = getenv("TV_GEMM_PERSIST"); return e && atoi(e) == 0; }();
  if (g_mode == 0 || (g_mode < 0 && off)) return false;
  const int nkt = a.K / BK;
  if (a.K % (2 * BK) || nkt < (epilogue == EPI_ACCUM ? 18 : 4)) return false;
  if (a.M < BM || a.N < BN || a.N % 8 || a.ldc % 8 || ((uintptr_t)a.C & 15)) return false;
  if (a.lda < 64 || a.ldw < 64) return false;
  if (g_mode < 0 && (int64_t)a.tiles_m * a.tiles_n < 4ll * grid_size()) return false;
  if (256 * a.ldc * 2 >= (1ll << 31)) return false;
  return true;
}

int launch_persist(const GemmArgs& a, int epilogue, hipStream_t st) {
  PArgs pa;
  pa.g = a;
  pa.ntiles = a.tiles_m * a.tiles_n;
  static const int dbg_env = [] { const char* e = getenv("TV_GEMM_DBG"); return e ? atoi(e) : 0; }();
  pa.dbg = dbg_env;
  const int grid = grid_size();
  switch (epilogue) {
    case EPI_BIAS: return launch<EPI_BIAS>(pa, grid, st);
    case EPI_BIAS_GELU: return launch<EPI_BIAS_GELU>(pa, grid, st);
    default: return launch<EPI_ACCUM>(pa, grid, st);
  }
}

void set_persist(int mode, int grid) { g_mode = mode; g_grid = grid; }

}  // namespace tvgemm

extern "C" void tv_gemm_set_persist(int mode, int grid) { tvgemm::set_persist(mode, grid); }
