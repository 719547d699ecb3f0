// bf16 GEMM with fused epilogues for the ViT linears ("fused beyond patch-embed"):
//
//     C[M][N] = epilogue( A[M][K] . W[N][K]^T )          A, W, C bf16 row-major (W = nn.Linear.weight as stored)
//
//   EPI_BIAS       C = acc + bias                                        (qkv projection)
//   EPI_BIAS_GELU  C = gelu(bf16(acc + bias))   exact (erf) GELU         (timm Mlp fc1 + act: base_vision.py:274-278 via
//                                                                         timm; vit_scale_clean.py:296-320)
//   EPI_ACCUM      C = C + acc                  (x += h W^T)             (attention / MLP output projections accumulating
//                                                                         into the residual stream, model/vit/siglip.py)
// The unfused path writes fc1's 13 GB of pre-activations per 2 048 frames and reads / rewrites them in a GELU pass
// (4.5 ms at 5.7 TB/s, 6.9 % of the 10 k-frame step); here the activation is applied to the accumulators.  The
// rounding points are the unfused path's: acc + bias is rounded to bf16 FIRST (what the GEMM would have written), GELU
// is evaluated in fp32 on that value with the same erf approximation as tv_gelu_fwd, and rounded again.
//
// Main loop: 256 x 256 x 64 tiles, 8 waves, one work-group per CU, v_mfma_f32_16x16x32_bf16 with the operands SWAPPED
// (W rows on the A side) so that a lane's 4 accumulator registers are 4 consecutive columns of one C row (8-byte pieces of
// a row; two m-tiles joined by v_permlane16_swap give 16-byte stores).
//   * LDS: 2 K-tiles x {A, W} x 2 halves of 128 rows x 128 bytes = 128 KiB, filled by LDS-DMA (global_load_lds_dwordx4:
//     every wave 2 instructions per half-tile), XOR-swizzled on the SOURCE side (chunk ^ (row >> 1) & 7) so that the
//     16-row x 64-byte fragment reads (ds_read_b128) are conflict-free.
//   * A wave owns rows {64 wr .. +63} of BOTH 128-row halves and columns {32 wc .. +31} of both 128-column halves
//     (wr = wave / 4, wc = wave % 4): quadrant (mh, nh) of its 128 x 64 outputs touches only half-tiles A[mh] and W[nh],
//     so the four phases of a K-tile need — and release — the four half-tiles one after the other:
//         phase 1: read A0 | Q(0,0)   phase 2: read W1 | Q(0,1)   phase 3: read A1 | Q(1,1)   phase 4: read W0(t+1) | Q(1,0)
//     (W0 of a K-tile is read one phase early into a second register set and serves phases 1 and 4: 8 / 4 / 8 / 4
//     fragment reads a phase instead of 12 / 4 / 8 / 0), and each half-tile buffer is busy for two phases out of eight:
//     every phase issues ONE half-tile copy, six phases ahead of its read (phase 1 of t: A1(t+1), 2: W0(t+2), 3: A0(t+2),
//     4: W1(t+2)); a counted s_waitcnt vmcnt(10) per phase keeps five half-tiles (80 KiB) in flight across the
//     barriers — never a drain inside the loop.
//   * The two wave groups wr = 0 / 1 (the two waves of each SIMD) run ONE BARRIER APART: while one group issues its 16
//     MFMAs of a phase, the other issues the fragment reads, copies and waits of its next phase — the matrix pipe of a
//     SIMD always has one wave feeding it (MI355X guide, "two waves per SIMD": pair matrix with memory).
//     Rules the schedule obeys (derived in DESIGN.md section 3): data retired by the wait in phase j is read in phase
//     j + 1 or later; a half-tile is overwritten no earlier than 2 phases after the phase of its last fragment read.
#include "gemm_common.hpp"

namespace tvgemm {
namespace {

constexpr int LDS_BYTES = RING_BYTES;

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int lc = lane & 15, kq = lane >> 4;

  // ---- tile of this work-group: ids that share an XCD (id % 8) walk neighbouring tiles, N fastest
  const int ntiles = a.tiles_m * a.tiles_n;
  int id = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, x = id % 8;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + id / 8;
  }
  // ... in bands of `gm` m-tiles, m fastest: the ~32 work-groups an XCD runs at a time cover gm x (32 / gm) tiles and
  // share each A / W k-slab through its L2 while they march through K together
  const int gm = a.group_m;
  const int band = id / (gm * a.tiles_n), in_band = id % (gm * a.tiles_n);
  const int rows_in_band = min(gm, a.tiles_m - band * gm);
  const int tm = band * gm + in_band % rows_in_band, tn = in_band / rows_in_band;
  const int m0 = tm * BM, n0 = tn * BN;

  const unsigned lds0 = lds_addr_of(smem_raw);

  // ---- staging: half-tile h of operand X = 16 pieces of 1 KiB (8 rows x 128 B); wave w copies pieces 2w, 2w + 1.
  // LDS byte (piece i, lane l) = row 8 i + l / 8, physical chunk l % 8, which holds source chunk (l % 8) ^ f(row),
  // f(row) = (row >> 1) & 7.  Rows past the matrix end repeat the last row (finite values, never stored).
  unsigned voffA[2][2], voffW[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 16 * wave + 8 * j + (lane >> 3);               // row of the half-tile
      const int ch = (lane & 7) ^ ((row >> 1) & 7);
      const int ra = min(m0 + 128 * h + row, a.M - 1) - m0;
      const int rw = min(n0 + 128 * h + row, a.N - 1) - n0;
      voffA[h][j] = (unsigned)((ra * a.lda + ch * 8) * 2);
      voffW[h][j] = (unsigned)((rw * a.ldw + ch * 8) * 2);
    }
  const bf16_t* Ag = a.A + (int64_t)m0 * a.lda;
  const bf16_t* Wg = a.W + (int64_t)n0 * a.ldw;
  const unsigned piece0 = (unsigned)(2 * wave * 1024);
  // X: 0 = A, 1 = W;  h: half;  t: K-tile
  auto stage = [&](int X, int h, int t) __attribute__((always_inline)) {
    const void* sp = uniform_ptr((X ? Wg : Ag) + (int64_t)t * BK);
    const unsigned dst = lds0 + (unsigned)((t & 1) * TILE_BYTES + (2 * X + h) * HALF_BYTES) + piece0;
    const unsigned v0 = X ? voffW[h][0] : voffA[h][0], v1 = X ? voffW[h][1] : voffA[h][1];
    unsigned keep;
    // (two M0 set-ups: an instruction offset would move the source address too, and a clamped row's offset may be < 1024)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\t"
                 "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "s"(sp), "s"(dst) : "memory");
  };

  // ---- fragment reads: 16 rows x 64 bytes, lane (lc = row, kq = 16-byte chunk of the k-step); the LDS address of a
  // fragment = one of four lane bases (operand side x k-step) + a compile-time offset (K-tile buffer, half, 16-row tile)
  const unsigned frag_lo = (unsigned)(lc * 128 + ((kq ^ (lc >> 1)) << 4));
  const unsigned a_b0 = lds0 + (unsigned)(wr * 64 * 128) + frag_lo, a_b1 = lds0 + (unsigned)(wr * 64 * 128) + (frag_lo ^ 64u);
  const unsigned w_b0 = lds0 + (unsigned)(wc * 32 * 128) + frag_lo, w_b1 = lds0 + (unsigned)(wc * 32 * 128) + (frag_lo ^ 64u);
  typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
#define GEMM_LD(base, off) (*(lds_bf16x8*)(size_t)((base) + (unsigned)(off)))

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[i][j][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 af[4][2];                          // A rows of the current m-half: [m-tile][k-step]
  bf16x8 wf0[2][2][2], wf1[2][2];           // W0 of the current and of the next K-tile (by K-tile parity), W1

  const int nkt = a.K / BK;
  const int nitems = 4 * nkt;               // half-tiles in issue order: item 4 t + {0: W0, 1: A0, 2: W1, 3: A1} of K-tile t

  // Copy schedule (ONE half-tile per phase, each 6 phases ahead of its read): phase g = 4 t + p (p = 1..4) READS item g —
  // p 1: A0(t), 2: W1(t), 3: A1(t), 4: W0(t + 1), which then stays in registers for phases 1 and 4 of K-tile t + 1 — and
  // ISSUES item g + 6 (items 0 .. 6 come from the prologue).  Item g + 6 lands in the buffer of item g - 2, read two
  // phases ago (rule 2); the wait of phase g retires item g + 1, read one phase later (rule 1), and leaves the 5 younger
  // half-tiles (80 KiB) in flight.
  // item = 4 t + k, k a COMPILE-TIME constant at every call (0: W0, 1: A0, 2: W1, 3: A1) — the offset registers are
  // picked statically (a run-time index would put the arrays into scratch)
  auto issue_item = [&](int t, auto KT) __attribute__((always_inline)) {
    constexpr int k = decltype(KT)::value;
    stage((k & 1) ? 0 : 1, k >> 1, t);
  };
  auto wait_items = [&](int allow) __attribute__((always_inline)) {       // at most `allow` half-tiles stay in flight
    if (allow >= 5) GEMM_WAIT_VM(10);
    else if (allow == 4) GEMM_WAIT_VM(8);
    else if (allow == 3) GEMM_WAIT_VM(6);
    else if (allow == 2) GEMM_WAIT_VM(4);
    else if (allow == 1) GEMM_WAIT_VM(2);
    else GEMM_WAIT_VM(0);
  };

  // ---- prologue: items 0 .. 6; W0(0) and A0(0) have landed behind vmcnt(10)
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
    wf0[0][n][0] = GEMM_LD(w_b0, 2 * HALF_BYTES + n * 2048);
    wf0[0][n][1] = GEMM_LD(w_b1, 2 * HALF_BYTES + n * 2048);
  }
  if (wr == 1) __builtin_amdgcn_s_barrier();          // the second wave of every SIMD runs one barrier behind

  // One phase = [fragment reads + one copy + counted wait] barrier [16 MFMAs] barrier
  auto mma_quadrant = [&](int mh, int nh, const bf16x8 (&wf)[2][2]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[mh][nh][m][n] = mfma16(wf[n][ks], af[m][ks], acc[mh][nh][m][n]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // DB: K-tile buffer (t & 1); TAIL: the last two K-tiles, where the copies run out and the waits shrink
  auto ktile = [&](int t, auto DBT, auto TAILT) __attribute__((always_inline)) {
    constexpr int DB = decltype(DBT)::value;
    constexpr bool TAIL = decltype(TAILT)::value;
    constexpr int TB = DB * TILE_BYTES;
    const int g0 = 4 * t;
    // phase p of K-tile t (g = 4 t + p) issues item g + 6 = 4 (t + 1 + (p + 2) / 4) + (p + 2) % 4 (items 0 .. 6 came from
    // the prologue) and waits until item g + 1 has landed: of the items issued so far, min(nitems, g + 7), the ones
    // younger than g + 1 may stay in flight
    auto copy_and_wait = [&](auto PT_) __attribute__((always_inline)) {
      constexpr int p = decltype(PT_)::value;
      const int g = g0 + p;
      const int tt = t + 1 + (p + 2) / 4;
      if (!TAIL) { issue_item(tt, std::integral_constant<int, (p + 2) % 4>{}); GEMM_WAIT_VM(10); }
      else {
        if (g + 6 < nitems) issue_item(tt, std::integral_constant<int, (p + 2) % 4>{});
        wait_items(min(g + 7, nitems) - g - 2);
      }
    };
    // ---- phase 1: A0(t)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      af[m][0] = GEMM_LD(a_b0, TB + m * 2048);
      af[m][1] = GEMM_LD(a_b1, TB + m * 2048);
    }
    copy_and_wait(std::integral_constant<int, 1>{});
    __builtin_amdgcn_s_barrier();
    mma_quadrant(0, 0, wf0[DB]);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: W1(t)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      wf1[n][0] = GEMM_LD(w_b0, TB + 3 * HALF_BYTES + n * 2048);
      wf1[n][1] = GEMM_LD(w_b1, TB + 3 * HALF_BYTES + n * 2048);
    }
    copy_and_wait(std::integral_constant<int, 2>{});
    __builtin_amdgcn_s_barrier();
    mma_quadrant(0, 1, wf1);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: A1(t)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      af[m][0] = GEMM_LD(a_b0, TB + HALF_BYTES + m * 2048);
      af[m][1] = GEMM_LD(a_b1, TB + HALF_BYTES + m * 2048);
    }
    copy_and_wait(std::integral_constant<int, 3>{});
    __builtin_amdgcn_s_barrier();
    mma_quadrant(1, 1, wf1);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4: W0(t + 1) into the other register set (behind the last K-tile: a read of stale bytes, never used)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      wf0[DB ^ 1][n][0] = GEMM_LD(w_b0, (TILE_BYTES - TB) + 2 * HALF_BYTES + n * 2048);
      wf0[DB ^ 1][n][1] = GEMM_LD(w_b1, (TILE_BYTES - TB) + 2 * HALF_BYTES + n * 2048);
    }
    copy_and_wait(std::integral_constant<int, 4>{});
    __builtin_amdgcn_s_barrier();
    mma_quadrant(1, 0, wf0[DB]);
    __builtin_amdgcn_s_barrier();
  };
  {
    using std::integral_constant;
    int t = 0;
    for (; t + 3 < nkt; t += 2) {       // phases g <= 4 (nkt - 3) + 4: item g + 7 exists, 5 half-tiles in flight
      ktile(t, integral_constant<int, 0>{}, integral_constant<bool, false>{});
      ktile(t + 1, integral_constant<int, 1>{}, integral_constant<bool, false>{});
    }
    // the last two K-tiles (the launcher takes an even number of K-tiles: t is even here)
    ktile(t, integral_constant<int, 0>{}, integral_constant<bool, true>{});
    ktile(t + 1, integral_constant<int, 1>{}, integral_constant<bool, true>{});
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();          // barrier counts match again

  // ---- epilogue.  A lane holds, for (mh, nh, m-tile, n-tile), columns .. + 4 kq + 0..3 of row .. + lc.  Two m-tiles at a
  // time: v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of the other, after which a
  // lane holds 8 CONSECUTIVE columns (16 n-tile + 8 (kq >> 1) + 0..7) of row 16 (m-tile + (kq & 1)) + lc: 16-byte loads of
  // C / stores instead of 8-byte ones (the epilogue is bound by the issue of its stores, MI355X guide T21).
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4e __attribute__((ext_vector_type(4)));
  const int row_in_pair = 16 * (kq & 1) + lc, col_in_tile = 8 * (kq >> 1);
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col = n0 + 128 * nh + 32 * wc + 16 * n + col_in_tile;
      float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (EPI != EPI_ACCUM && a.bias && col < a.N) {
        if (a.bias_f32) {
          const f32x4 b0 = *(const f32x4*)((const float*)a.bias + col);
          const f32x4 b1 = col + 4 < a.N ? *(const f32x4*)((const float*)a.bias + col + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) { bv[r] = b0[r]; bv[4 + r] = b1[r]; }
        } else {
          const bf16x4 b0 = *(const bf16x4*)((const bf16_t*)a.bias + col);
          const bf16x4 b1 = col + 4 < a.N ? *(const bf16x4*)((const bf16_t*)a.bias + col + 4) : bf16x4{};
#pragma unroll
          for (int r = 0; r < 4; ++r) { bv[r] = (float)b0[r]; bv[4 + r] = (float)b1[r]; }
        }
      }
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mp = 0; mp < 4; mp += 2) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[mh][nh][mp][n][r]),
                                                             __float_as_uint(acc[mh][nh][mp + 1][n][r]), false, false);
            v[r] = __uint_as_float(sw[0]);
            v[4 + r] = __uint_as_float(sw[1]);
          }
          const int row = m0 + 128 * mh + 64 * wr + 16 * mp + row_in_pair;
          if (row >= a.M || col >= a.N) continue;
          bf16_t* cp = a.C + (int64_t)row * a.ldc + col;
          const bool whole = col + 8 <= a.N && (a.ldc % 8 == 0) && (((uintptr_t)a.C & 15) == 0);       // 16-byte piece inside the row
          if (EPI == EPI_ACCUM) {
            if (whole) {
              const bf16x8 c8 = *(const bf16x8*)cp;
#pragma unroll
              for (int r = 0; r < 8; ++r) v[r] += (float)c8[r];
            } else {
              const bf16x4 c4 = *(const bf16x4*)cp;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += (float)c4[r];
              if (col + 4 < a.N) {
                const bf16x4 d4 = *(const bf16x4*)(cp + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 + r] += (float)d4[r];
              }
            }
          } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += bv[r];
          }
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
          if (whole) *(u32x4e*)cp = o;
          else {
            typedef unsigned u32x2e __attribute__((ext_vector_type(2)));
            *(u32x2e*)cp = u32x2e{o[0], o[1]};
            if (col + 4 < a.N) *(u32x2e*)(cp + 4) = u32x2e{o[2], o[3]};
          }
        }
    }
}

template <int EPI>
int launch_gemm(const GemmArgs& a, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) {
    tv_set_error("gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return TV_ERR_LAUNCH;
  }
  gemm_bf16_kernel<EPI><<<dim3((unsigned)(a.tiles_m * a.tiles_n)), 512, LDS_BYTES, st>>>(a);
  TV_LAUNCH_CHECK();
}

}  // namespace
}  // namespace tvgemm

using namespace tvgemm;

extern "C" int tv_gemm_bf16_fwd(const void* A, const void* W, const void* bias, void* C, int64_t M, int N, int K,
                                int64_t lda, int64_t ldw, int64_t ldc, int epilogue, int bias_dtype, void* stream) {
  TV_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm: bad sizes (M %lld N %d K %d)", (long long)M, N, K);
  if (M == 0) return TV_OK;
  TV_CHECK_ARG(A && W && C, "gemm: null pointer");
  TV_CHECK_ARG(epilogue >= 0 && epilogue <= 2, "gemm: epilogue %d", epilogue);
  TV_CHECK_ARG(bias == nullptr || bias_dtype == TV_F32 || bias_dtype == TV_BF16, "gemm: bias dtype %d", bias_dtype);
  if (K % (2 * BK) || N % 4 || lda % 8 || ldw % 8 || ldc % 4 || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 7) ||
      ((uintptr_t)bias & 15))
    TV_UNSUPPORTED("gemm: K must be a multiple of %d, N of 4, rows 16-byte aligned (K %d N %d lda %lld ldw %lld ldc %lld)", 2 * BK, K, N,
                   (long long)lda, (long long)ldw, (long long)ldc);
  if (M > (1ll << 31) - BM || 256 * lda * 2 >= (1ll << 31) || 256 * ldw * 2 >= (1ll << 31))
    TV_UNSUPPORTED("gemm: sizes beyond the 32-bit offsets of the copies");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.bias = bias; a.C = (bf16_t*)C;
  a.M = (int)M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc;
  a.tiles_m = (int)((M + BM - 1) / BM); a.tiles_n = (N + BN - 1) / BN;
  a.bias_f32 = bias_dtype == TV_F32;
  {
    static const int gm_env = [] { const char* e = getenv("TV_GEMM_GROUP_M"); return e ? atoi(e) : 0; }();     // dev tool
    // measured at 1 492 992 rows (bench_gemm_fused.py, gm 1 / 4 / 8 / 16 / 32): N 4 352: 15.8 / 15.3 / 15.7 / 15.4 / 15.7 ms,
    // N 3 584: 10.9 / 10.4 / 10.5 / 10.8 / 11.5, N 1 152 (K 4 352): 12.8 / 12.6 / 12.5 / 13.3 / 16.0
    a.group_m = gm_env > 0 ? gm_env : (a.tiles_n >= 8 ? 4 : 8);
    if (a.group_m > a.tiles_m) a.group_m = a.tiles_m;
  }
  if ((int64_t)a.tiles_m * a.tiles_n >= (1ll << 31)) TV_UNSUPPORTED("gemm: too many tiles");
  hipStream_t st = (hipStream_t)stream;
  // the ViT's big shapes: persistent work-groups — 256 x 192 tiles that leave during the next tile's K loop (gemm_drip.hip),
  // else 256 x 256 tiles with the epilogue by quadrants inside the main loop (gemm_persist.hip)
  if (drip_takes(a, epilogue, persist_grid())) return launch_drip(a, epilogue, persist_grid(), st);
  if (persist_takes(a, epilogue)) return launch_persist(a, epilogue, st);
  switch (epilogue) {
    case EPI_BIAS: return launch_gemm<EPI_BIAS>(a, st);
    case EPI_BIAS_GELU: return launch_gemm<EPI_BIAS_GELU>(a, st);
    default: return launch_gemm<EPI_ACCUM>(a, st);
  }
}
