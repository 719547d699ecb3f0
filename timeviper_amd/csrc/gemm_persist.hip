// Persistent bf16 GEMM for the ViT linears (timm block via base_vision.py:146-170,274-278, InternVideo2
// Attention / Mlp vit_scale_clean.py:188-320):  C[M][N] = epilogue( A[M][K] . W[N][K]^T ) for the bias epilogues of gemm.hip
// (EPI_BIAS: the qkv projection; EPI_BIAS_GELU: fc1 + exact GELU).
//
// Why a second kernel.  The linears of a SigLIP block have K = 1 152: 18 K-tiles of 64.  With one work-group per output
// tile every tile pays, outside its K loop, the pipeline fill, the conversion of 256 x 256 accumulators (with the
// GELU: ~1 700 vector instructions a wave) and 128 KB of C stores: 24 - 32 % of a launch (DESIGN.md, K sweep).  Here a
// work-group stays on its CU and walks a list of tiles:
//   * the copy pipeline (one half-tile per phase, six phases ahead, gemm.hip's schedule) never stops: the last two
//     K-tiles of a tile already issue the first copies of the next one;
//   * the epilogue is taken apart by QUADRANT.  A wave's 128 x 64 outputs are four quadrants (mh, nh) of 32 accumulator
//     registers; in the last K-tile they become final one phase after the other (Q00, Q01, Q11, Q10).  The next tile's
//     first K-tile starts every accumulator from a zero C operand, so no register is cleared and no second accumulator
//     set is needed.  A quadrant's bias / GELU / conversion / stores run right behind its last MFMAs, and BOTH wave
//     groups do theirs in the same interval: group 0 (one barrier ahead) in the load segment of its next phase, group 1
//     at the tail of its MFMA segment — two waves of a SIMD issuing vector instructions together get twice the issue rate
//     of one (wave64 on a SIMD-32), where the first build (both groups in their load segments, i.e. one after the other
//     with the partner parked at the barrier) exposed the GELU twice per quadrant;
//   * the stores enter the copies' in-order vmcnt queue; the counted waits stay at the plain count (ten pieces), which
//     is never too lenient.  Measured: letting them allow for exactly the stores in front of them changes nothing, and
//     neither does moving all stores to waves that never wait (wave group 0 copies, group 1 stores, hand-over through
//     LDS: built, correct, slower): what a tile's 128 KB of stores cost (~2.8 us, 23 B/clk and CU) they cost in the CU's
//     memory pipeline, in front of the copies, whoever issues them and whoever waits.  Store shapes measured on qkv
//     (9.95 ms): 16 rows x 64 B per instruction 10.05, whole 128-byte lines (values misplaced, timing only) 9.70,
//     `nt` 12.5, `sc0 sc1` 11.1 ms;
//   * bias: 1 KiB per tile by LDS-DMA into one of two slots (every wave issues the same copy: the counts stay uniform),
//     read back in the epilogue;
//   * tiles: ids that share an XCD (blockIdx % 8, observed dispatch order; speed only) own one contiguous eighth of the
//     banded tile order (bands of group_m m-tiles, m fastest), and the 32 work-groups of an XCD walk it side by side: they
//     share every A / W k-slab through their L2 exactly like the per-tile kernel's dispatch order did.
//   * edge tiles are SHIFTED back inside the matrix (m0 = M - 256, n0 = N - 256) instead of clamped: every copy address
//     is tile-independent + a scalar base, and the stores of rows / columns that belong to the neighbour are masked.
// The accumulate epilogue (C += A W^T) stays on the per-tile kernel: a persistent variant was built (old C entering
// through the matrix pipe: 16 x 32 pieces of C copied to LDS five phases ahead and multiplied by a 0 / 1 selector
// fragment, eight steps a tile; parity-green, in the history of this file) and measured 18 % slower per K-tile than
// the bias variants — the sixteen uniform branches a K-tile that pick a step's static accumulator registers cost more
// than the hidden epilogue returns (proj 4.57 against 4.46 ms, fc2 15.0 against 12.4 ms per 2 048 frames).
#include "gemm_common.hpp"

namespace tvgemm {
namespace {

constexpr int BIAS_SLOT = 1024;
constexpr int LDS_BYTES = RING_BYTES + 2 * BIAS_SLOT;
enum { KT_FIRST = 0, KT_MID = 1, KT_LAST = 2 };

struct PArgs {
  GemmArgs g;
  int ntiles;
  int dbg;       // dev switches (TV_GEMM_DBG): 1 = no epilogue, 2 = epilogue arithmetic without stores
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
  const bool has_bias = a.bias != nullptr;

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

  // ---- epilogue of one quadrant of the tile at (m0, n0): v_permlane16_swap joins two m-tiles so that a lane holds 8
  // consecutive columns (16 n + 8 (kq >> 1) + 0..7) of row 16 (mp + (kq & 1)) + lc: 16-byte stores.  (A second swap level
  // on the packed pairs that makes every store instruction cover 16 rows x 64 bytes instead of 32 rows x 32 bytes was
  // built and measured 1 % slower: the store cost does not follow the number of lines an instruction touches.)
  auto epi_quadrant = [&](auto MHT, auto NHT, int m0, int n0, unsigned bias_lds) __attribute__((always_inline)) {
    constexpr int MH = decltype(MHT)::value, NH = decltype(NHT)::value;
    if (pa.dbg & 1) return;      // dev: the K loop alone
    bf16_t* Cb = c_base(m0, n0);
    const int skr = (-m0) & (BM - 1), skc = (-n0) & (BN - 1);
    const bool edge = ((m0 | n0) & (BM - 1)) != 0;
    const int lx = opaque_lane();
    const int lane_r = 64 * wr + 16 * ((lx >> 4) & 1) + (lx & 15), lane_c = 32 * wc + 8 * (lx >> 5);
    const unsigned lane_coff = (unsigned)((lane_r * ldc + lane_c) * 2);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col_t = 128 * NH + 16 * n + lane_c;
      float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (has_bias) {
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
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += bv[r];
        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = gelu_erf_f((float)(bf16_t)v[r]);       // the GEMM's own bf16 rounding first
        }
        u32x4e o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bf16x2 pk = {(bf16_t)v[2 * r], (bf16_t)v[2 * r + 1]};
          o[r] = __builtin_bit_cast(unsigned, pk);
        }
        const int row_t = 128 * MH + 16 * mp + lane_r;
        unsigned char* cp = (unsigned char*)(Cb + (int64_t)(128 * MH + 16 * mp) * ldc + (128 * NH + 16 * n)) + lane_coff;
        if (pa.dbg & 2) asm volatile("" ::"v"(o));        // dev: the epilogue's arithmetic without its stores
        else if (!edge || (row_t >= skr && col_t >= skc)) *(u32x4e*)cp = o;
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

  // DB: K-tile buffer (t & 1); KIND: first / middle / last K-tile of a tile
  auto ktile = [&](int t, auto DBT, auto KINDT) __attribute__((always_inline)) {
    using std::integral_constant;
    typedef integral_constant<int, 0> I0;
    typedef integral_constant<int, 1> I1;
    constexpr int DB = decltype(DBT)::value;
    constexpr int KIND = decltype(KINDT)::value;
    constexpr int TB = DB * TILE_BYTES;
    constexpr bool Z = KIND == KT_FIRST;
    const unsigned bias_cur = bias_lds0 + (unsigned)(jpar * BIAS_SLOT);
    // phase p issues item g + 6 = (t + 1 + (p + 2) / 4, (p + 2) % 4) and waits until item g + 1 has landed: at most the ten
    // youngest operations stay in flight — the five younger half-tiles, or fewer of them where stores or the bias copy sit
    // between them in the queue (stricter than needed there, never too lenient)
    auto copy = [&](auto PT_) __attribute__((always_inline)) {
      constexpr int p = decltype(PT_)::value;
      issue_item(t + 1 + (p + 2) / 4, integral_constant<int, (p + 2) % 4>{});
    };
    auto wait = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
      GEMM_WAIT_VM(10);
      __builtin_amdgcn_s_barrier();
    };
    // the quadrant that became final in this phase: group 1 converts and stores it at once, group 0 in its next load
    // segment — the same interval
    auto tail = [&](auto MHT, auto NHT) __attribute__((always_inline)) {
      mma_end();
      if (KIND == KT_LAST && wr == 1) epi_quadrant(MHT, NHT, cm0, cn0, bias_cur);
      __builtin_amdgcn_s_barrier();
    };
    // ---- phase 1: A0(t)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      af[m][0] = PG_LD(a_b0, TB + m * 2048);
      af[m][1] = PG_LD(a_b1, TB + m * 2048);
    }
    copy(I1{});
    if (KIND == KT_MID && t == nkt - 2) {
      // every later copy belongs to the next tile (a work-group without one copies its last tile's first K-tiles again:
      // the counted waits stay the same, nothing reads them)
      if (idx + wpx < idx_end) decode(idx + wpx, nm0, nn0);
      Ais = a.A + (int64_t)nm0 * lda;
      Wis = a.W + (int64_t)nn0 * ldw;
      tsub = nkt;
    }
    if (KIND == KT_FIRST && have_prev && wr == 0)
      epi_quadrant(I1{}, I0{}, pm0, pn0, bias_lds0 + (unsigned)((jpar ^ 1) * BIAS_SLOT));
    wait();
    mma_quadrant(I0{}, I0{}, integral_constant<bool, Z>{}, wf0[DB]);
    tail(I0{}, I0{});
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
      glds16(uniform_ptr(bp), bl, (unsigned)__builtin_amdgcn_readfirstlane((int)bias_cur));
    }
    if (KIND == KT_LAST && wr == 0) epi_quadrant(I0{}, I0{}, cm0, cn0, bias_cur);
    wait();
    mma_quadrant(I0{}, I1{}, integral_constant<bool, Z>{}, wf1);
    tail(I0{}, I1{});
    // ---- phase 3: A1(t)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      af[m][0] = PG_LD(a_b0, TB + HALF_BYTES + m * 2048);
      af[m][1] = PG_LD(a_b1, TB + HALF_BYTES + m * 2048);
    }
    copy(integral_constant<int, 3>{});
    if (KIND == KT_LAST && wr == 0) epi_quadrant(I0{}, I1{}, cm0, cn0, bias_cur);
    wait();
    mma_quadrant(I1{}, I1{}, integral_constant<bool, Z>{}, wf1);
    tail(I1{}, I1{});
    // ---- phase 4: W0(t + 1) into the other register set
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      wf0[DB ^ 1][n][0] = PG_LD(w_b0, (TILE_BYTES - TB) + 2 * HALF_BYTES + n * 2048);
      wf0[DB ^ 1][n][1] = PG_LD(w_b1, (TILE_BYTES - TB) + 2 * HALF_BYTES + n * 2048);
    }
    copy(integral_constant<int, 4>{});
    if (KIND == KT_LAST && wr == 0) epi_quadrant(I1{}, I1{}, cm0, cn0, bias_cur);
    wait();
    mma_quadrant(I1{}, I0{}, integral_constant<bool, Z>{}, wf0[DB]);
    tail(I1{}, I0{});
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
  if (wr == 0) {
    __builtin_amdgcn_s_barrier();          // barrier counts match again
    epi_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, cm0, cn0,
                 bias_lds0 + (unsigned)(jpar * BIAS_SLOT));
  }
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

// Shapes the persistent kernel takes: the bias epilogues, whole 16-byte pieces everywhere, at least one full tile in both
// directions (edge tiles are shifted back, not clamped) and enough tiles that every work-group has a few (the per-tile
// kernel is as good below that).
bool persist_takes(const GemmArgs& a, int epilogue) {
  static const int off = [] { const char* e = getenv("TV_GEMM_PERSIST"); return e && atoi(e) == 0; }();
  if (g_mode == 0 || (g_mode < 0 && off)) return false;
  const int nkt = a.K / BK;
  if (epilogue == EPI_ACCUM || a.K % (2 * BK) || nkt < 4) return false;
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
    default: return launch<EPI_BIAS_GELU>(pa, grid, st);
  }
}

void set_persist(int mode, int grid) { g_mode = mode; g_grid = grid; }
int persist_grid() { return grid_size(); }

}  // namespace tvgemm

extern "C" void tv_gemm_set_persist(int mode, int grid) { tvgemm::set_persist(mode, grid); }
