// T3 (decode): ONE query token per sequence against a long K / V cache — the attention of a generate() step
// (modeling_nano.py:1198-1209 with q_len == 1; the cache layout of HybridMambaAttentionDynamicCache, :205-360).
//
// The prefill kernel (attention.hip) gives a (batch, q-head) pair to one work-group: with one query row that is
// Hq work-groups walking the whole cache one after the other (32 for Nemotron-Nano-9B: 1/8 of the chip, each streaming
// 32 868 keys serially, the four q-heads of a kv-head each fetching the same K / V).  Here the work is cut the other way:
//   * the q-heads that share a kv-head (<= 16) are the N columns of ONE MFMA tile: K and V are read once per kv-head;
//   * the keys are split over `nsplit` work-groups per kv-head (grid = nsplit x Hkv x batch, >= one per CU), each
//     wave of a work-group walks every fourth 32-key step of its range, two steps of copies in flight;
//   * a second, tiny kernel merges the per-split (max, sum, O) partials (log-sum-exp combine).
// The key count can come from DEVICE memory (`seqlens_k`): a decode step captured in a hipGraph replays with a
// growing cache without re-capture — the grid is sized for the capacity, the ranges are cut in the kernel.
//
// Per 32-key step of a wave (v_mfma_f32_16x16x32_bf16, everything on the matrix pipe):
//   S^T[key][head] = K . Q^T     A = K rows (ds_read_b128 of the wave's 32 x 256-byte K tile), B = Q^T fragments held for
//                                the whole kernel;  2 tiles x 4 k-steps
//   online softmax               a lane owns one head (n = lane % 16) and 8 keys; running max per head, agreed between
//                                the 4 lanes of a head with two v_permlane swaps
//   O^T[d][head] += V^T . P      A = V^T: the V tile read with the transposing ds_read_b64_tr_b16; B = P, which the S^T accumulators are after a bf16 pack — the key
//                                order of the two S^T tiles is chosen so that no lane exchange is needed
// K and V tiles arrive by LDS-DMA (global_load_lds_dwordx4, 16 bytes a lane, no register round trip) into a ring of
// AD_NS stages PRIVATE to the wave, AD_NS - 1 steps ahead of the math behind counted vmcnt waits: no barrier in the
// loop; one at the end for the merge of the waves.  (K fragments as plain 16-byte global loads into registers would be
// the natural A operand, but a register ring filled by asynchronous loads does not survive the compiler: the copies it
// places on the loop's back edge read registers whose loads are still in flight.)
#include "ssd_common.hpp"

namespace {
using namespace ssdk;

#ifndef AD_NW
#define AD_NW 3                               // waves per work-group
#endif
#ifndef AD_NS
#define AD_NS 3                               // ring stages per wave (a stage: the K and the V tile of one step)
#endif
constexpr int AD_STEP = 32;                   // keys per wave step
constexpr int AD_D = 128;                     // head_dim of this kernel
constexpr int AD_ROWB = AD_D * 2;             // bytes of a K / V row in LDS
constexpr int AD_TILE = AD_STEP * AD_ROWB;    // 8 KiB
constexpr int AD_STAGE = 2 * AD_TILE;         // K tile, V tile
constexpr int AD_RING = AD_NW * AD_NS * AD_STAGE;                 // 144 KiB
constexpr int AD_MERGE = AD_NW * 32 * 64 * 4 + 2 * AD_NW * 64 * 4;     // [wave][ct * 4 + i][lane] fp32 O, [wave][lane] max, sum:
constexpr int AD_LDS = AD_RING > AD_MERGE ? AD_RING : AD_MERGE;        // written over the ring once the walk is over
constexpr int AD_OPS = 16;                    // copy instructions per step and wave
constexpr int AD_MAX_SPLIT = 256;

struct DecArgs {
  const bf16_t *q, *k, *v;
  bf16_t* o;
  float *lse, *part_o, *part_ml;
  const int* seqlens;                         // device, one per batch entry; nullptr: Lk for all
  int Lk, Hq, Hkv, gq, hblocks, nsplit;
  int64_t qsb, qsh, ksb, ksl, ksh, vsb, vsl, vsh, osb, osh;
  float scale_log2;
};

// keys of a sequence and the keys per split (a multiple of the step), the same in both kernels
__device__ __forceinline__ void split_of(const DecArgs& a, int b, int& Lk, int& chunk) {
  Lk = a.seqlens ? min(max(a.seqlens[b], 0), a.Lk) : a.Lk;
  chunk = ((Lk + a.nsplit - 1) / a.nsplit + AD_STEP - 1) / AD_STEP * AD_STEP;
  if (chunk == 0) chunk = AD_STEP;
}

__device__ __forceinline__ float xor16_max(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_sum(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__global__ __launch_bounds__(AD_NW * 64) void attn_decode_kernel(DecArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ad_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;                 // MFMA column (q-head / key row / d column) and k-quarter
  const int split = blockIdx.x, b = blockIdx.z;
  const int hk = blockIdx.y / a.hblocks, hb = blockIdx.y % a.hblocks;
  const int nh = min(16, a.gq - 16 * hb);                 // q-heads of this block
  const int head0 = hk * a.gq + 16 * hb;
  int Lk, chunk;
  split_of(a, b, Lk, chunk);
  const int r0 = split * chunk, r1 = min(Lk, r0 + chunk);
  if (r0 >= r1) return;                                   // (the merge kernel skips the same splits)
  const int nsteps = (r1 - r0 + AD_STEP - 1) / AD_STEP;
  const int nst = nsteps > wave ? (nsteps - wave + AD_NW - 1) / AD_NW : 0;      // steps wave, wave + 4, ...

  const unsigned lds0 = lds_addr_of(ad_smem);
  const unsigned ring = lds0 + (unsigned)(wave * AD_NS * AD_STAGE);

  // ---- Q^T fragments (B operand): lane (n, g) holds Q[head0 + n][32 ks + 8 g .. + 7]; heads past the block are zeros
  bf16x8 qf[4];
  {
    const bf16_t* qp = a.q + (int64_t)b * a.qsb + (int64_t)(head0 + min(n, nh - 1)) * a.qsh + 8 * g;
    u32x4 raw[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) raw[ks] = gload16_async(qp + 32 * ks);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      settle(raw[ks]);
      const bf16x8 z = {};
      qf[ks] = n < nh ? __builtin_bit_cast(bf16x8, raw[ks]) : z;
    }
  }

  // ---- K rows (A operand of S^T): tile G row m = key kb + 8 (m >> 2) + 4 G + (m & 3), so that the accumulator rows
  // 4 g + i of tiles 0 / 1 are keys kb + 8 g + i and kb + 8 g + 4 + i: slots 0..7 of P's fragment for k-quarter g
  const bf16_t* kg = a.k + (int64_t)b * a.ksb + (int64_t)hk * a.ksh + (int64_t)r0 * a.ksl;
  const bf16_t* vg = a.v + (int64_t)b * a.vsb + (int64_t)hk * a.vsh + (int64_t)r0 * a.vsl;
  const int last = r1 - 1 - r0;                            // last key of the range, relative to r0
  // ---- copies: instruction i8 of a tile, lane l -> row 4 i8 + (l >> 4), physical 16-byte chunk l & 15, holding logical
  // chunk (l & 15) ^ f(row):  K: f = (row & 3) | ((row >> 1) & 12) (distinct over the 16 rows {0..3, 8..11, 16..19, 24..27} (+ 4)
  // that the 16 lanes of a ds_read_b128 group touch);
  // V: f = 2 ((row & 3) | ((row >> 1) & 4)) (the 32 lanes of one half of a transposing read — rows 8 kq + q4, kq in a
  // pair, 8-byte pieces p4 of the 32 bytes of d-tile ct — touch every bank once)
  const int crow = lane >> 4;
  const unsigned vchunk = (unsigned)((lane & 15) ^ (2 * (lane >> 4)));

  auto issue = [&](int i, int S) __attribute__((always_inline)) {
    const int kb = AD_STEP * (wave + AD_NW * i);           // relative to r0; rows past the range repeat its last key
    const void* kbp = uniform_ptr(kg);
    const void* vbp = uniform_ptr(vg);
    const unsigned dst = ring + (unsigned)(S * AD_STAGE);
#pragma unroll
    for (int i8 = 0; i8 < 8; ++i8) {
      const int row = 4 * i8 + crow, key = min(kb + row, last);
      const unsigned ch = (unsigned)((lane & 15) ^ ((row & 3) | ((row >> 1) & 12)));
      glds16(kbp, (unsigned)key * (unsigned)(a.ksl * 2) + ch * 16u, dst + (unsigned)(i8 * 1024));
    }
#pragma unroll
    for (int i8 = 0; i8 < 8; ++i8) {
      const int key = min(kb + 4 * i8 + crow, last);
      const unsigned ch = vchunk ^ (unsigned)(8 * ((i8 >> 1) & 1));
      glds16(vbp, (unsigned)key * (unsigned)(a.vsl * 2) + ch * 16u, dst + (unsigned)(AD_TILE + i8 * 1024));
    }
  };
  // K fragment reads: lane (m = n, kq = g), tile G, k-step ks: row 8 (m >> 2) + 4 G + (m & 3), logical chunk 4 ks + kq
  const int krow = 8 * (n >> 2) + (n & 3);
  unsigned k_rd[2];
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    const int row = krow + 4 * G;
    k_rd[G] = (unsigned)(row * AD_ROWB + ((g ^ ((row & 3) | ((row >> 1) & 12))) << 4));
  }

  // transposing reads: lane (lc = n, kq = g) gets keys 8 kq + 0..7 of d column 16 ct + lc (two reads: rows 8 kq + q4 and + 4)
  const int q4 = n >> 2, p4 = n & 3;
  const unsigned tr_base = (unsigned)((8 * g + q4) * AD_ROWB + 16 * (p4 >> 1) + 8 * (p4 & 1)) | (unsigned)((q4 | (4 * (g & 1))) << 5);

  f32x4 oacc[8];
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) oacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
  auto compute = [&](int i, int S) __attribute__((always_inline)) {
    const int kb = AD_STEP * (wave + AD_NW * i);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(AD_OPS * (AD_NS - 1)) : "memory");     // the younger steps stay in flight
    const unsigned stage = ring + (unsigned)(S * AD_STAGE);
    f32x4 sacc[2];
#pragma unroll
    for (int G = 0; G < 2; ++G) {
      sacc[G] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // (4 ks + g) ^ f = 4 ks ^ (g ^ f) for f < 16: the k-step flips bits 2..3 of the physical chunk
        const bf16x8 kf = *(lds_bf16x8*)(size_t)(stage + (k_rd[G] ^ (unsigned)(ks << 6)));
        sacc[G] = mfma16(kf, qf[ks], sacc[G]);
      }
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int G = 0; G < 2; ++G)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = kb + 8 * g + 4 * G + j <= last;
        sacc[G][j] = ok ? sacc[G][j] * a.scale_log2 : -INFINITY;
        tmax = fmaxf(tmax, sacc[G][j]);
      }
    tmax = xor32_max(xor16_max(tmax));
    const float m_new = fmaxf(m_run, tmax);                // finite: every step holds at least one key of the range
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float ps = 0.f;
    bf16x8 pf;
#pragma unroll
    for (int G = 0; G < 2; ++G)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p = __builtin_amdgcn_exp2f(sacc[G][j] - m_new);
        ps += p;
        pf[4 * G + j] = from_f32<bf16_t>(p);
      }
    l_run = l_run * alpha + ps;
    if (__builtin_amdgcn_ballot_w64(m_new > m_run)) {
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) oacc[ct] *= alpha;
    }
    m_run = m_new;
    const unsigned st = stage + (unsigned)AD_TILE;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const unsigned p = st + (tr_base ^ (unsigned)(ct << 5));
      const bf16x8 vf = cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(size_t)p),
                             __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(size_t)(p + 4 * AD_ROWB)));
      oacc[ct] = mfma16(vf, pf, oacc[ct]);
    }
  };

  {
#pragma unroll
    for (int j = 0; j < AD_NS - 1; ++j) issue(j, j);
    int S = 0;                                             // ring stage of step i
    for (int i = 0; i < nst; ++i) {
      const int Sn = S == 0 ? AD_NS - 1 : S - 1;           // (S + AD_NS - 1) % AD_NS: the stage step i - 1 has left
      issue(i + AD_NS - 1, Sn);
      compute(i, S);
      S = S == AD_NS - 1 ? 0 : S + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (copies of steps past the range)
  }

  // ---- merge the waves: [wave][ct * 4 + i][lane] partial O, per-lane max / sum (sum over the head's 4 lanes first);
  // the arrays lie over the ring, which every wave has left
  l_run = xor32_sum(xor16_sum(l_run));
  __syncthreads();
  float* mo = (float*)ad_smem;
  float* mm = mo + AD_NW * 32 * 64;
  float* ml = mm + AD_NW * 64;
#pragma unroll
  for (int ct = 0; ct < 8; ++ct)
#pragma unroll
    for (int j = 0; j < 4; ++j) mo[(wave * 32 + ct * 4 + j) * 64 + lane] = oacc[ct][j];
  mm[wave * 64 + lane] = m_run;
  ml[wave * 64 + lane] = l_run;
  __syncthreads();
  float m_tot = -INFINITY;
#pragma unroll
  for (int w = 0; w < AD_NW; ++w) m_tot = fmaxf(m_tot, mm[w * 64 + lane]);       // finite: wave 0 always has a step
  float sc[AD_NW], l_tot = 0.f;
#pragma unroll
  for (int w = 0; w < AD_NW; ++w) {
    sc[w] = __builtin_amdgcn_exp2f(mm[w * 64 + lane] - m_tot);                    // exp2(-inf) = 0 for idle waves
    l_tot += sc[w] * ml[w * 64 + lane];
  }
  if (n < nh) {
    const int64_t slot = ((int64_t)b * a.Hq + head0 + n) * a.nsplit + split;
    for (int ct = wave; ct < 8; ct += AD_NW) {
      f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < AD_NW; ++w)
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] += sc[w] * mo[(w * 32 + ct * 4 + j) * 64 + lane];
      *(f32x4*)(a.part_o + slot * AD_D + 16 * ct + 4 * g) = r;     // O^T rows d = 16 ct + 4 g + j of column n
    }
    if (wave == 0 && g == 0) {
      a.part_ml[slot * 2] = m_tot;
      a.part_ml[slot * 2 + 1] = l_tot;
    }
  }
}

// o[b][h][d] = sum_s 2^(m_s - M) O_s[d] / sum_s 2^(m_s - M) l_s over the splits that hold keys; grid (Hq, batch), 128 threads.
// One pass with a running maximum, the loads of 8 splits issued before their first use (a split at a time the kernel was
// 32 dependent round trips to L2: 12 us for 1 MB).
__global__ __launch_bounds__(AD_D) void attn_decode_merge_kernel(DecArgs a) {
  constexpr int U = 8;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  int Lk, chunk;
  split_of(a, b, Lk, chunk);
  const int nvalid = (Lk + chunk - 1) / chunk;
  const int64_t slot0 = ((int64_t)b * a.Hq + h) * a.nsplit;
  float M = -INFINITY, L = 0.f, O = 0.f;
  for (int s0 = 0; s0 < nvalid; s0 += U) {
    f32x2 ml[U];
    float o[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sidx = min(s0 + u, nvalid - 1);        // (past the end: the last split again, weighted 0 below)
      ml[u] = *(const f32x2*)(a.part_ml + (slot0 + sidx) * 2);
      o[u] = a.part_o[(slot0 + sidx) * AD_D + d];
    }
    float mb = M;
#pragma unroll
    for (int u = 0; u < U; ++u) mb = s0 + u < nvalid ? fmaxf(mb, ml[u][0]) : mb;
    const float sc = __builtin_amdgcn_exp2f(M - mb);   // (M = -inf: 0)
    L *= sc;
    O *= sc;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float w = s0 + u < nvalid ? __builtin_amdgcn_exp2f(ml[u][0] - mb) : 0.f;
      L = fmaf(w, ml[u][1], L);
      O = fmaf(w, o[u], O);
    }
    M = mb;
  }
  a.o[(int64_t)b * a.osb + (int64_t)h * a.osh + d] = from_f32<bf16_t>(nvalid > 0 ? O / L : 0.f);
  if (a.lse && d == 0)
    a.lse[(int64_t)b * a.Hq + h] = nvalid > 0 ? (M + __log2f(L)) * 0.6931471805599453f : -INFINITY;
}

int cu_count() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
      cus = 256;                               // MI355X
    return cus;
  }();
  return n;
}

int pick_splits(int batch, int nheads_kv, int hblocks, int seqlen_k) {
  const int per = batch * nheads_kv * hblocks;
  int want = (cu_count() + per - 1) / per;                          // one work-group per CU
  const int most = (seqlen_k + AD_NW * AD_STEP - 1) / (AD_NW * AD_STEP);          // >= one step per wave
  if (want > most) want = most;
  if (want > AD_MAX_SPLIT) want = AD_MAX_SPLIT;
  return want < 1 ? 1 : want;
}

}  // namespace

extern "C" size_t tv_attn_decode_workspace_bytes(int batch, int nheads_q, int nheads_kv, int seqlen_k) {
  if (batch <= 0 || nheads_q <= 0 || nheads_kv <= 0 || nheads_q % nheads_kv || seqlen_k < 0) return 0;
  const int hblocks = (nheads_q / nheads_kv + 15) / 16;
  const size_t ns = (size_t)pick_splits(batch, nheads_kv, hblocks, seqlen_k);
  return (size_t)batch * nheads_q * ns * (AD_D + 2) * sizeof(float);
}

extern "C" int tv_attn_decode_fwd(const void* q, const void* k, const void* v, void* o, void* lse, int batch,
                                  int seqlen_k, const int* seqlens_k, int nheads_q, int nheads_kv, int headdim,
                                  int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_b, int64_t k_stride_l,
                                  int64_t k_stride_h, int64_t v_stride_b, int64_t v_stride_l, int64_t v_stride_h,
                                  int64_t o_stride_b, int64_t o_stride_h, float softmax_scale, int dtype,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  TV_CHECK_ARG(q && o && (seqlen_k == 0 || (k && v)), "attn_decode: null pointer");
  TV_CHECK_ARG(batch > 0 && seqlen_k >= 0 && nheads_q > 0 && nheads_kv > 0 && nheads_q % nheads_kv == 0 &&
                   softmax_scale > 0.f, "attn_decode: bad sizes (or non-positive softmax scale)");
  if (dtype != TV_BF16) TV_UNSUPPORTED("attn_decode: dtype must be bf16 (tv_flash_attn_fwd takes f16)");
  if (headdim != AD_D) TV_UNSUPPORTED("attn_decode: headdim %d (this kernel: 128; tv_flash_attn_fwd takes the rest)", headdim);
  const int64_t strides[] = {q_stride_b, q_stride_h, k_stride_b, k_stride_l, k_stride_h, v_stride_b, v_stride_l, v_stride_h};
  for (int64_t s : strides)
    if (s % 8) TV_UNSUPPORTED("attn_decode: q/k/v strides must be multiples of 8 elements");
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15))
    TV_UNSUPPORTED("attn_decode: q/k/v must be 16-byte aligned");
  if (k_stride_l < AD_D || v_stride_l < AD_D) TV_UNSUPPORTED("attn_decode: k/v rows must not overlap");
  const int gq = nheads_q / nheads_kv, hblocks = (gq + 15) / 16;
  const int nsplit = pick_splits(batch, nheads_kv, hblocks, seqlen_k);
  // V rows are addressed with 32-bit byte offsets from the first key of a split
  if (((int64_t)seqlen_k / nsplit + 2 * AD_STEP) * v_stride_l * 2 >= (1ll << 31) ||
      ((int64_t)seqlen_k / nsplit + 2 * AD_STEP) * k_stride_l * 2 >= (1ll << 31))
    TV_UNSUPPORTED("attn_decode: k / v row stride too large for a split of %d keys", seqlen_k / nsplit);
  const size_t need = (size_t)batch * nheads_q * nsplit * (AD_D + 2) * sizeof(float);
  TV_CHECK_ARG(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0,
               "attn_decode: workspace of %zu bytes (16-byte aligned) needed, got %zu", need, workspace_bytes);
  DecArgs a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.lse = (float*)lse;
  a.part_o = (float*)workspace;
  a.part_ml = a.part_o + (size_t)batch * nheads_q * nsplit * AD_D;
  a.seqlens = seqlens_k;
  a.Lk = seqlen_k; a.Hq = nheads_q; a.Hkv = nheads_kv; a.gq = gq; a.hblocks = hblocks; a.nsplit = nsplit;
  a.qsb = q_stride_b; a.qsh = q_stride_h; a.ksb = k_stride_b; a.ksl = k_stride_l; a.ksh = k_stride_h;
  a.vsb = v_stride_b; a.vsl = v_stride_l; a.vsh = v_stride_h; a.osb = o_stride_b; a.osh = o_stride_h;
  a.scale_log2 = softmax_scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  static const hipError_t attr = hipFuncSetAttribute((const void*)attn_decode_kernel,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, AD_LDS);
  if (attr != hipSuccess) {
    tv_set_error("attn_decode: cannot reserve %d bytes of LDS: %s", AD_LDS, hipGetErrorString(attr));
    return TV_ERR_LAUNCH;
  }
  if (seqlen_k > 0)
    attn_decode_kernel<<<dim3((unsigned)nsplit, (unsigned)(nheads_kv * hblocks), (unsigned)batch), AD_NW * 64, AD_LDS, st>>>(a);
  attn_decode_merge_kernel<<<dim3((unsigned)nheads_q, (unsigned)batch), AD_D, 0, st>>>(a);
  TV_LAUNCH_CHECK();
}
