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
//         phase 1: read A0, W0 | Q(0,0)     phase 2: read W1 | Q(0,1)     phase 3: read A1 | Q(1,1)     phase 4: Q(1,0)
//     (W0 stays in registers for phase 4), and the staging of K-tile t + 2 can start while K-tile t is still computed:
//         phase 1 of t: W1(t+1)   phase 2: A1(t+1)   phase 3: A0(t+2)   phase 4: W0(t+2)
//     each 4 phases ahead of its first read; a counted s_waitcnt vmcnt(8) per phase keeps four half-tiles in flight
//     across the barriers (never a drain inside the loop).
//   * The two wave groups wr = 0 / 1 (the two waves of each SIMD) run ONE BARRIER APART: while one group issues its 16
//     MFMAs of a phase, the other issues the fragment reads, copies and waits of its next phase — the matrix pipe of a
//     SIMD always has one wave feeding it (MI355X guide, "two waves per SIMD": pair matrix with memory).
//     Rules the schedule obeys (derived in DESIGN.md section 3): data retired by the wait in phase j is read in phase
//     j + 1 or later; a half-tile is overwritten no earlier than 2 phases after the phase of its last fragment read.
#include <math.h>
#include "ssd_common.hpp"

namespace {
using namespace ssdk;

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_ACCUM = 2 };

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;          // one half-tile: 128 rows x 128 bytes
constexpr int TILE_BYTES = 4 * HALF_BYTES;        // A0 A1 W0 W1 of one K-tile
constexpr int LDS_BYTES = 2 * TILE_BYTES;         // two K-tiles

struct GemmArgs {
  const bf16_t *A, *W;
  const void* bias;        // fp32 or bf16 (bias_f32), may be NULL
  bf16_t* C;
  int M, N, K;
  int64_t lda, ldw, ldc;
  int tiles_m, tiles_n;
  int bias_f32;
};

// same formula as norms.hip's gelu_erf (A&S 7.1.26): the fused and the two-pass path agree bit for bit
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
  const float erf_abs = __builtin_fmaf(-p, e, 1.f);
  const float hx = 0.5f * x;
  return __builtin_fmaf(hx, copysignf(erf_abs, x), hx);
}

#define GEMM_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

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
  const int tm = id / a.tiles_n, tn = id % a.tiles_n;
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

  // ---- fragment reads: 16 rows x 64 bytes, lane (lc = row, kq = 16-byte chunk of the k-step)
  const unsigned frag_lo = (unsigned)(lc * 128 + ((kq ^ (lc >> 1)) << 4));
  auto ldfrag = [&](unsigned base, int ks) {
    typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
    return *(lds_bf16x8*)(size_t)(base + (frag_lo ^ (unsigned)(ks * 64)));
  };

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[i][j][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 af[4][2];          // A rows of the current m-half: [m-tile][k-step]
  bf16x8 wf0[2][2], wf1[2][2];

  const int nkt = a.K / BK;

  // ---- prologue: K-tile 0 whole, A0 / W0 of K-tile 1
  stage(0, 0, 0); stage(1, 0, 0); stage(1, 1, 0); stage(0, 1, 0);
  if (nkt > 1) { stage(0, 0, 1); stage(1, 0, 1); GEMM_WAIT_VM(4); }
  else GEMM_WAIT_VM(0);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();          // the second wave of every SIMD runs one barrier behind

  // One phase = [reads + copies + wait] barrier [16 MFMAs] barrier
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
  auto read_a = [&](int t, int mh) __attribute__((always_inline)) {
    const unsigned base = lds0 + (unsigned)((t & 1) * TILE_BYTES + mh * HALF_BYTES + wr * 64 * 128);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[m][ks] = ldfrag(base + m * 2048, ks);
  };
  auto read_w = [&](int t, int nh, bf16x8 (&wf)[2][2]) __attribute__((always_inline)) {
    const unsigned base = lds0 + (unsigned)((t & 1) * TILE_BYTES + (2 + nh) * HALF_BYTES + wc * 32 * 128);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wf[n][ks] = ldfrag(base + n * 2048, ks);
  };

  for (int t = 0; t < nkt; ++t) {
    // what the counted waits may leave in flight: 4 half-tiles in the steady state; fewer where the copies run out
    const bool s1 = t + 1 < nkt, s2 = t + 2 < nkt;
    // ---- phase 1: A0, W0 | copy W1(t+1)
    read_w(t, 0, wf0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(t, 0);
    if (s1) stage(1, 1, t + 1);
    if (s1) GEMM_WAIT_VM(8); else GEMM_WAIT_VM(2);                                      // retires W1(t)
    __builtin_amdgcn_s_barrier();
    mma_quadrant(0, 0, wf0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: W1 | copy A1(t+1)
    read_w(t, 1, wf1);
    if (s1) stage(0, 1, t + 1);
    if (s1) GEMM_WAIT_VM(8); else GEMM_WAIT_VM(0);                                      // retires A1(t)
    __builtin_amdgcn_s_barrier();
    mma_quadrant(0, 1, wf1);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: A1 | copy A0(t+2)
    read_a(t, 1);
    if (s2) { stage(0, 0, t + 2); GEMM_WAIT_VM(8); }                                    // retires A0(t+1); (else: phase 4 reads nothing new)
    __builtin_amdgcn_s_barrier();
    mma_quadrant(1, 1, wf1);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4: (W0 still in registers) | copy W0(t+2)
    if (s2) { stage(1, 0, t + 2); GEMM_WAIT_VM(8); }                                    // retires W0(t+1)
    else if (s1) GEMM_WAIT_VM(4);                                                       // A0(t+1), W0(t+1) landed; W1, A1 of t+1 in flight
    __builtin_amdgcn_s_barrier();
    mma_quadrant(1, 0, wf0);
    __builtin_amdgcn_s_barrier();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();          // barrier counts match again

  // ---- epilogue: lane holds, for (mh, nh, m-tile, n-tile), columns n = .. + 4 kq + 0..3 of row m = .. + lc
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2e __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col = n0 + 128 * nh + 32 * wc + 16 * n + 4 * kq;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (EPI != EPI_ACCUM && a.bias && col < a.N) {
        if (a.bias_f32) bv = *(const f32x4*)((const float*)a.bias + col);
        else {
          const bf16x4 b4 = *(const bf16x4*)((const bf16_t*)a.bias + col);
          bv = f32x4{(float)b4[0], (float)b4[1], (float)b4[2], (float)b4[3]};
        }
      }
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int row = m0 + 128 * mh + 64 * wr + 16 * m + lc;
          if (row >= a.M || col >= a.N) continue;
          bf16_t* cp = a.C + (int64_t)row * a.ldc + col;
          f32x4 v = acc[mh][nh][m][n];
          if (EPI == EPI_ACCUM) {
            const bf16x4 c4 = *(const bf16x4*)cp;
            v = f32x4{v[0] + (float)c4[0], v[1] + (float)c4[1], v[2] + (float)c4[2], v[3] + (float)c4[3]};
          } else {
            v = f32x4{v[0] + bv[0], v[1] + bv[1], v[2] + bv[2], v[3] + bv[3]};
          }
          if (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f((float)(bf16_t)v[r]);       // the GEMM's own bf16 rounding first
          }
          const bf16x2 p01 = {(bf16_t)v[0], (bf16_t)v[1]}, p23 = {(bf16_t)v[2], (bf16_t)v[3]};
          *(u32x2e*)cp = u32x2e{__builtin_bit_cast(unsigned, p01), __builtin_bit_cast(unsigned, p23)};
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

extern "C" int tv_gemm_bf16_fwd(const void* A, const void* W, const void* bias, void* C, int64_t M, int N, int K,
                                int64_t lda, int64_t ldw, int64_t ldc, int epilogue, int bias_dtype, void* stream) {
  TV_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm: bad sizes (M %lld N %d K %d)", (long long)M, N, K);
  if (M == 0) return TV_OK;
  TV_CHECK_ARG(A && W && C, "gemm: null pointer");
  TV_CHECK_ARG(epilogue >= 0 && epilogue <= 2, "gemm: epilogue %d", epilogue);
  TV_CHECK_ARG(bias == nullptr || bias_dtype == TV_F32 || bias_dtype == TV_BF16, "gemm: bias dtype %d", bias_dtype);
  if (K % BK || N % 4 || lda % 8 || ldw % 8 || ldc % 4 || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 7) ||
      ((uintptr_t)bias & 15))
    TV_UNSUPPORTED("gemm: K must be a multiple of %d, N of 4, rows 16-byte aligned (K %d N %d lda %lld ldw %lld ldc %lld)", BK, K, N,
                   (long long)lda, (long long)ldw, (long long)ldc);
  if (M > (1ll << 31) - BM || 256 * lda * 2 >= (1ll << 31) || 256 * ldw * 2 >= (1ll << 31))
    TV_UNSUPPORTED("gemm: sizes beyond the 32-bit offsets of the copies");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.bias = bias; a.C = (bf16_t*)C;
  a.M = (int)M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc;
  a.tiles_m = (int)((M + BM - 1) / BM); a.tiles_n = (N + BN - 1) / BN;
  a.bias_f32 = bias_dtype == TV_F32;
  if ((int64_t)a.tiles_m * a.tiles_n >= (1ll << 31)) TV_UNSUPPORTED("gemm: too many tiles");
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case EPI_BIAS: return launch_gemm<EPI_BIAS>(a, st);
    case EPI_BIAS_GELU: return launch_gemm<EPI_BIAS_GELU>(a, st);
    default: return launch_gemm<EPI_ACCUM>(a, st);
  }
}
