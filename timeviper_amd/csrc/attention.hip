// A1 / T3 / ViT: fused softmax attention forward on CDNA4 MFMA tiles.
//
// Structure (wave64, v_mfma_f32_32x32x16): a workgroup of 8 waves owns 256 query
// rows of one (batch, q-head) (4 waves / 128 rows for short queries); each wave keeps
// its 32 query rows as MFMA B fragments in registers for the whole kernel.  K/V tiles
// of 96 keys arrive by LDS-DMA into a 3-stage ring shared by the waves, two tiles ahead
// of the math (counted vmcnt, one barrier per tile).  The score tile is computed
// TRANSPOSED, S^T = K . Q^T, so every lane owns one query column: the softmax
// row statistics are per-lane scalars (one cross-half shuffle per tile), and the
// P^T accumulator registers are, after a bf16 pack, directly the B operand of
// O^T += V^T . P^T — V^T fragments come from the row-major V tile through the
// gfx950 transposing LDS read (ds_read_b64_tr_b16).  No S x S matrix, no
// repeat_kv copy for GQA, exp2-domain online softmax in fp32.
//
// Reference semantics: _flash_attention_forward (modeling_nano.py:1198-1209,
// causal, no positional encoding, scale 1/sqrt(d)), SDPA (:1300-1307,
// cross_attention.py:310-317 non-causal), flash_attn_varlen_qkvpacked_func
// (flash_attention_class.py:59-66).
#include "ssd_common.hpp"
#include <atomic>

namespace {

constexpr int FA_QW = 32;                  // query rows per wave
#ifndef FA_KT
#define FA_KT 3                            // key sub-tiles of 32 per LDS tile (96 keys: +7 % causal d=128 vs 64)
#endif

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// Fragment reads take a 32-bit LDS byte offset: one base register per fragment family and iteration,
// everything compile-time (sub-tile, k-step, row half) in the instruction's offset field.  (With generic
// pointers hipcc spent a v_add per read: 112 of the ~330 VALU instructions of a ViT tile, in a loop
// that is VALU-bound at head_dim 72.)
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ const lds_u8* lds_at(unsigned off) { return (const lds_u8*)(uintptr_t)off; }

using ssdk::u32x4;
using ssdk::gload16_async;
using ssdk::settle;

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  typedef bf16x8 v8; typedef bf16x4 v4;
  static constexpr bool is_bf16 = true;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ v4 tr_read(const lds_u8* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  }
  static __device__ __forceinline__ v8 row_read(const lds_u8* p) {
    return *(const __attribute__((address_space(3))) v8*)p;
  }
};
template <> struct Frag<f16_t> {
  typedef f16x8 v8; typedef f16x4 v4;
  static constexpr bool is_bf16 = false;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ v4 tr_read(const lds_u8* p) {
    const s16x4 raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
    return __builtin_bit_cast(f16x4, raw);
  }
  static __device__ __forceinline__ v8 row_read(const lds_u8* p) {
    return *(const __attribute__((address_space(3))) v8*)p;
  }
};

struct AttnArgs {
  const void *q, *k, *v;
  void* o;
  float* lse;
  int Lq, Lk, Hq, Hkv, D;
  int64_t qsb, qsl, qsh, ksb, ksl, ksh, vsb, vsl, vsh, osb, osl, osh;
  float scale_log2;  // softmax_scale * log2(e)
  int causal;
  int nqb, nb;       // nqb > 0: 1-D XCD-aware grid, nqb query blocks per (batch, head), nb batches; 0: 3-D grid
  int ppx;           // streaming kernel: (batch, head) pairs per XCD
  int o16;           // streaming kernel: o rows are 16-byte aligned (wide epilogue stores)
  int notrim;        // streaming kernel (dev, TV_FA_TRIM=0): the last key tile runs all its sub-tiles
};

// KS = ceil(D/16) k-steps of QK^T, DT = ceil(D/32) d-tiles of PV, NW waves per workgroup.
// K/V tiles of 32*KT keys go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 bytes per
// lane, no VGPR round trip: staging through registers cost a third of the kernel) into a
// ring of 3 stages, two tiles ahead of the math, with counted vmcnt waits and one barrier
// per tile.  LDS rows are 256 bytes (128 elements); the 16-byte chunks of a row are
// XOR-swizzled on the DMA source address so that the row reads of K (ds_read_b128, chunk ^
// row%16) and the transposing reads of V (ds_read_b64_tr_b16, chunk ^ 4(row%4)) are bank
// conflict free.  Chunks past head_dim are never copied (the K ones QK^T reads are zeroed once).
template <typename T, int KS, int DT, int NW, int KT>
__global__ __launch_bounds__(NW * 64) void flash_fwd_kernel(AttnArgs a) {
  constexpr int FA_KB = 32 * KT;     // keys per tile
  constexpr int NPK = FA_KB / 4;     // 1 KiB DMA pieces (4 key rows) of K per tile; as many of V
  typedef typename Frag<T>::v8 v8;
  typedef typename Frag<T>::v4 v4;
  constexpr int QB = NW * FA_QW;     // query rows per workgroup
  constexpr int ROWB = 256;          // LDS bytes per key row
  constexpr int TILEB = FA_KB * ROWB;
  constexpr int NS = 3;              // ring stages
  constexpr int PPW = 2 * NPK / NW;  // DMA pieces per wave and tile
  static_assert(2 * NPK % NW == 0, "pieces must divide over the waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char fa_smem[];
  unsigned char* const sK = fa_smem;
  unsigned char* const sV = fa_smem + NS * TILEB;
  const unsigned sK_off = (unsigned)(uintptr_t)(lds_u8*)fa_smem, sV_off = sK_off + NS * TILEB;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, hh = lane >> 5;
  // causal: launch the heaviest (last) query blocks first
  // 3-D grid (causal: heaviest query blocks first), or — many short sequences (ViT frames) — a 1-D
  // grid whose ids walk the query blocks of ONE (batch, head) on one XCD (id % 8): the blocks that
  // share K / V then share an L2 instead of pulling the same keys into three of them
  int qblk, h, b;
  if (a.nqb > 0) {
    const int slot = blockIdx.x >> 3;
    // an XCD owns a contiguous range of (batch, head) pairs: neighbouring heads of a frame interleave
    // in memory (head_dim 72 = 144-byte rows inside 128-byte lines), so their shared lines hit one L2
    const int pair = (blockIdx.x & 7) * (int)(gridDim.x / (8 * a.nqb)) + slot / a.nqb;
    qblk = slot % a.nqb;
    h = pair % a.Hq;
    b = pair / a.Hq;
    if (b >= a.nb) return;                    // (batch, head) pairs are padded to a multiple of 8
  } else {
    qblk = a.causal ? (gridDim.x - 1 - blockIdx.x) : blockIdx.x;
    h = blockIdx.y;
    b = blockIdx.z;
  }
  const int hk = h / (a.Hq / a.Hkv);
  const int q0 = qblk * QB + wave * FA_QW;      // first query row of this wave
  const int qrow = q0 + r;                      // this lane's query row
  const int D = a.D;
  const int shift = a.Lk - a.Lq;                // bottom-right causal alignment

  const T* qp = (const T*)a.q + (int64_t)b * a.qsb + (int64_t)h * a.qsh;
  const T* kp = (const T*)a.k + (int64_t)b * a.ksb + (int64_t)hk * a.ksh;
  const T* vp = (const T*)a.v + (int64_t)b * a.vsb + (int64_t)hk * a.vsh;

  // K chunks between head_dim and the padded 16*KS columns are read by QK^T (against zero Q
  // columns) but never copied: zero them once so that stale LDS bits cannot be NaN/Inf.  (V
  // chunks past head_dim only feed output rows that are never stored.)
  if (D < 16 * KS) {
    const v8 z = {};
    const int c0 = D / 8, nc = 2 * KS - c0;
    for (int i = tid; i < NS * FA_KB * nc; i += NW * 64) {
      const int row = (i / nc) % FA_KB, st = i / (nc * FA_KB), c = c0 + i % nc;
      *reinterpret_cast<v8*>(sK + st * TILEB + row * ROWB + ((c ^ (row & 15)) << 4)) = z;
    }
  }

  // Q^T fragments (B operand): lane (r,hh) holds Q[qrow][16ks + 8hh + j]; untracked loads, settled
  // behind the first counted wait below
  v8 qf[KS];
  u32x4 qraw[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int d0 = ks * 16 + hh * 8;
    qraw[ks] = gload16_async((qrow < a.Lq && d0 < D) ? qp + (int64_t)qrow * a.qsl + d0 : qp);
  }

  f32x16 oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // key range this workgroup needs
  const int wg_q_last = min(qblk * QB + QB, a.Lq) - 1;
  int k_end = a.causal ? min(a.Lk, wg_q_last + shift + 1) : a.Lk;
  if (k_end < 0) k_end = 0;
  const int ntiles = (k_end + FA_KB - 1) / FA_KB;

  // ---- DMA pieces of this wave: piece id pc = wave + NW*i; pc < 16 -> K rows 4pc..4pc+3,
  // else V rows 4(pc-16)..; lane l carries row 4(pc%16) + l/16, LDS slot l%16 ----
  int p_row[PPW];
  unsigned p_off[PPW];            // byte offset from the tile's first key (full tiles)
  int p_chunk[PPW];               // source 16-byte chunk of the row (>= D/8: nothing to copy)
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = wave + NW * i;
    const bool isK = pc < NPK;
    const int row = 4 * (isK ? pc : pc - NPK) + (lane >> 4);
    const int c = (lane & 15) ^ (isK ? (row & 15) : 4 * (row & 3));
    p_row[i] = row;
    p_chunk[i] = c;
    p_off[i] = (unsigned)(((int64_t)row * (isK ? a.ksl : a.vsl) + c * 8) * (int)sizeof(T));
  }
  const int dchunks = D / 8;
  auto issue_piece = [&](int kt, int i) {          // piece i (0..PPW-1) of tile kt
    const int kbase = kt * FA_KB;
    const int stage = kt % NS;
    const int pc = wave + NW * i;
    const bool isK = pc < NPK;           // wave-uniform
    const void* tb = ssdk::uniform_ptr(isK ? (const void*)(kp + (int64_t)kbase * a.ksl)
                                           : (const void*)(vp + (int64_t)kbase * a.vsl));
    const int left = a.Lk - kbase;          // rows of this tile that exist
    unsigned off = p_off[i];
    if (left < FA_KB) {                  // last tile: rows past the end repeat the last key (masked later)
      const int rr = min(p_row[i], left - 1);
      off = (unsigned)(((int64_t)rr * (isK ? a.ksl : a.vsl) + p_chunk[i] * 8) * (int)sizeof(T));
    }
    const unsigned dst = ssdk::lds_addr_of((isK ? sK : sV) + stage * TILEB + (isK ? pc : pc - NPK) * 1024);
    if (p_chunk[i] < dchunks) ssdk::glds16(tb, off, dst);     // EXEC masks the pad chunks
  };
  auto issue_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(kt, i);
  };

  // fragment read offsets inside a stage
  int k_rd[KS];                   // K row fragment: row t*32 + r, chunk (2ks + hh) ^ (r % 16)
  const int kz = (hh ^ (r & 15)) << 4;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_rd[ks] = r * ROWB + ((32 * ks) ^ kz);
  const int q4 = (lane & 15) >> 2, p4 = lane & 3;
  int v_rd[DT];                   // V^T fragment: row 16s + 4hh + q4 (+8), chunk (4dt + cc) ^ 4 q4
  {
    const int cc = 2 * ((lane >> 4) & 1) + (p4 >> 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      v_rd[dt] = (4 * hh + q4) * ROWB + ((4 * (dt ^ q4) + cc) << 4) + (p4 & 1) * 8;
  }

  __syncthreads();                 // LDS zeroed before the first copy lands
  if (ntiles > 0) issue_tile(0);
  if (ntiles > 1) issue_tile(1);
  if (ntiles > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {      // the Q loads are older than every copy: landed
    settle(qraw[ks]);
    const v8 z = {};
    qf[ks] = (qrow < a.Lq && ks * 16 + hh * 8 < D) ? __builtin_bit_cast(v8, qraw[ks]) : z;
  }

  for (int kt = 0; kt < ntiles; ++kt) {
    const int kbase = kt * FA_KB;
    const bool ahead = kt + 2 < ntiles;
    // stage bases as LDS byte offsets (scalar), folded into one base register per fragment family
    const unsigned stage_off = (unsigned)((kt % NS) * TILEB);
    const unsigned cK = sK_off + stage_off, cV = sV_off + stage_off;

    // wave-uniform skip of tiles entirely above this wave's causal diagonal
    const int wave_q_last = q0 + FA_QW - 1;
    const bool active = !(a.causal && kbase > wave_q_last + shift);
    if (!active && ahead) issue_tile(kt + 2);
    if (active) {
      // ---- S^T = K . Q^T  (2 key sub-tiles of 32); all K fragment reads ahead of the MFMAs;
      // the copies of tile kt+2 are issued between the MFMAs, whose pipe time hides them ----
      f32x16 sacc[KT];
      v8 kf[KT][KS];
      unsigned kb[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kb[ks] = cK + (unsigned)k_rd[ks];
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[t][ks] = Frag<T>::row_read(lds_at(kb[ks]) + t * (32 * ROWB));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[t][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          sacc[t] = Frag<T>::mfma(kf[t][ks], qf[ks], sacc[t]);
          // PPW pieces spread over the 2*KS MFMAs
          constexpr int every = (KT * KS) / PPW > 0 ? (KT * KS) / PPW : 1;
          const int idx = t * KS + ks;
          if (ahead && idx % every == 0 && idx / every < PPW) issue_piece(kt + 2, idx / every);
        }
      }
      if (ahead) {      // pieces that did not fit the spacing (PPW > 2*KS)
#pragma unroll
        for (int i = (KT * KS) / ((KT * KS) / PPW > 0 ? (KT * KS) / PPW : 1); i < PPW; ++i) issue_piece(kt + 2, i);
      }
      // ---- mask, running max on the raw scores (the scale is positive, so it commutes with
      // max), then p = 2^(s*scale - m) as one FMA + v_exp per score; packed fp32 math ----
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const bool need_mask = (kbase + FA_KB > a.Lk) || (a.causal && kbase + FA_KB - 1 > q0 + shift);
      if (need_mask) {
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = kbase + t * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            const bool ok = key < a.Lk && (!a.causal || key <= qrow + shift);
            sacc[t][i] = ok ? sacc[t][i] : -INFINITY;
          }
      }
      float tmax = -INFINITY;
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int i = 0; i < 16; i += 2) tmax = fmaxf(fmaxf(tmax, sacc[t][i]), sacc[t][i + 1]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax * a.scale_log2);
      // rows with nothing visible yet keep m=-inf: use 0 as the exponent base
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);  // m_run=-inf -> 0
      const f32x2 sc2 = {a.scale_log2, a.scale_log2}, nm2 = {-m_use, -m_use};
      f32x2 ps2 = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const f32x2 e = __builtin_elementwise_fma(f32x2{sacc[t][i], sacc[t][i + 1]}, sc2, nm2);
          const f32x2 pp = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
          sacc[t][i] = pp[0];
          sacc[t][i + 1] = pp[1];
          ps2 += pp;
        }
      l_run = l_run * alpha + (ps2[0] + ps2[1]);
      // the accumulators only need rescaling when some row's maximum moved (wave-uniform test)
      if (__builtin_amdgcn_ballot_w64(m_new > m_run)) {
        const f32x2 al2 = {alpha, alpha};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const f32x2 v = f32x2{oacc[dt][i], oacc[dt][i + 1]} * al2;
            oacc[dt][i] = v[0];
            oacc[dt][i + 1] = v[1];
          }
      }
      m_run = m_new;

      // ---- O^T += V^T . P^T over 4 k-steps of 16 keys; the V^T fragments of k-step s+1 are in
      // flight while k-step s multiplies ----
      {
        unsigned vb[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vb[dt] = cV + (unsigned)v_rd[dt];
        auto read_v = [&](int s_, v4 (&lo)[DT], v4 (&hi)[DT]) {
          // element j of this lane is key row 16s + 8(j>>2) + 4hh + (j&3) of the tile
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            lo[dt] = Frag<T>::tr_read(lds_at(vb[dt]) + s_ * (16 * ROWB));
            hi[dt] = Frag<T>::tr_read(lds_at(vb[dt]) + s_ * (16 * ROWB) + 8 * ROWB);
          }
        };
        v4 vlo[2][DT], vhi[2][DT];
        read_v(0, vlo[0], vhi[0]);
#pragma unroll
        for (int s_ = 0; s_ < 2 * KT; ++s_) {
          if (s_ < 2 * KT - 1) read_v(s_ + 1, vlo[(s_ + 1) & 1], vhi[(s_ + 1) & 1]);
          const int t = s_ >> 1, rb = (s_ & 1) * 8;
          v8 pf;
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[j] = from_f32<T>(sacc[t][rb + j]);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            v8 vf;
#pragma unroll
            for (int j = 0; j < 4; ++j) { vf[j] = vlo[s_ & 1][dt][j]; vf[4 + j] = vhi[s_ & 1][dt][j]; }
            oacc[dt] = Frag<T>::mfma(vf, pf, oacc[dt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // tile kt+1 (issued an iteration ago) must have landed; this iteration's copies stay in flight
    if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: normalise and store O[q][d] ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  if (qrow < a.Lq) {
    T* op = (T*)a.o + (int64_t)b * a.osb + (int64_t)qrow * a.osl + (int64_t)h * a.osh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = dt * 32 + 8 * g + 4 * hh;
        if (d0 < D) {
          v4 pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) pk[j] = from_f32<T>(oacc[dt][4 * g + j] * inv);
          *(v4*)(op + d0) = pk;
        }
      }
    if (a.lse && hh == 0) {
      const float lse = l_tot > 0.f ? (m_run * 0.6931471805599453f + logf(l_tot)) : -INFINITY;
      a.lse[((int64_t)b * a.Hq + h) * a.Lq + qrow] = lse;
    }
  }
}


// ssdk::glds16 for the lanes of `live` only (a wave-uniform mask): the other lanes' 16 LDS bytes keep what they hold.
__device__ __forceinline__ void glds16_lanes(const void* sbase, unsigned voff, unsigned lds_dst, unsigned long long live) {
  unsigned keep;
  unsigned long long ex;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, %5\n\t"
               "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(lds_dst), "s"(live) : "memory");
}

// -DTV_FA_STAMP (dev): waves 0 and 4 of work-group 0 of the streaming kernel sum the cycles of each phase
// of tile kt (s_memtime): g_fa_stamps[wave/4][kt < 16][phase < 8]; tv_fa_debug_stamps() copies them out.
#ifdef TV_FA_STAMP
__device__ unsigned long long g_fa_stamps[2 * 16 * 8];
#define FSTAMP(ph) do { if (st_on) { const unsigned long long n__ = clock64(); if (lane == 0) st_acc[((wave >> 2) * 16 + (kt < 16 ? kt : 15)) * 8 + ph] += n__ - st_last; st_last = n__; } } while (0)
#else
#define FSTAMP(ph) do {} while (0)
#endif

// Many short sequences (ViT frames: 729 tokens = 8 key tiles per query block): a work-group that
// handles ONE query block spends a third of its life outside the tile loop — waiting for Q and the
// first two K/V tiles, zeroing LDS, storing O — and at 147 KiB of LDS nothing else is resident on
// the CU to cover that (measured: 27.9 us per work-group of which 8 x 2.2 us in the loop).  This
// kernel keeps the work-group and STREAMS query blocks through it: the K/V ring never drains — the
// copies of the next block's first two tiles are issued under the last two tiles of the current
// one, its Q fragments are loaded under the last tile — and the O stores of a block complete under
// the first tile of the next.  Non-causal, >= 2 key tiles.  Grid: 8 x (work-groups per XCD); the
// work-groups of XCD x walk slots x*... of that XCD's contiguous range of (batch, head) pairs, so the
// query blocks that share K / V still meet in one L2 at about the same time.
// ONES (head_dim a multiple of 8 and < 32 DT: the ViT's 72): the row sums of P come out of the P.V MFMAs.  The pad chunks
// of the rings are written ONCE (zeros; V column head_dim = 1.0) and the copy lanes that would land on them are switched
// off (a constant EXEC mask per wave around each piece: the swizzle depends on row & 15 only), so O^T row head_dim
// accumulates sum_k bf16(P[q][k]) — rescaled by alpha with the rest of the tile — and the per-tile adds for l go away.
// l is then the sum of the ROUNDED weights, the ones the output was built from.
template <typename T, int KS, int DT, int KT, bool ONES>
__global__ __launch_bounds__(512) void flash_fwd_stream_kernel(AttnArgs a) {
  constexpr int NW = 8;
  constexpr int FA_KB = 32 * KT;
  constexpr int NPK = FA_KB / 4;
  typedef typename Frag<T>::v8 v8;
  typedef typename Frag<T>::v4 v4;
  constexpr int QB = NW * FA_QW;
  constexpr int ROWB = 256;
  constexpr int TILEB = FA_KB * ROWB;
  constexpr int NS = 3;
  constexpr int PPW = 2 * NPK / NW;
  static_assert(2 * NPK % NW == 0, "pieces must divide over the waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char fa_smem[];
  const unsigned sK_off = (unsigned)(uintptr_t)(lds_u8*)fa_smem, sV_off = sK_off + NS * TILEB;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int D = a.D;
  const int xcd = blockIdx.x & 7, step = gridDim.x >> 3;
  const int nslots = a.ppx * a.nqb, npairs = a.nb * a.Hq;
  int slot = blockIdx.x >> 3;
  int pair = xcd * a.ppx + slot / a.nqb, qblk = slot % a.nqb;
  if (slot >= nslots || pair >= npairs) return;
  const int gq = a.Hq / a.Hkv;
  auto k_of = [&](int pr) { return (const T*)a.k + (int64_t)(pr / a.Hq) * a.ksb + (int64_t)((pr % a.Hq) / gq) * a.ksh; };
  auto v_of = [&](int pr) { return (const T*)a.v + (int64_t)(pr / a.Hq) * a.vsb + (int64_t)((pr % a.Hq) / gq) * a.vsh; };
  // Q^T fragments (B operand): lane (r,hh) holds Q[qrow][16ks + 8hh + j]
  // untracked loads (gload16_async) + a finishing step behind the counted wait that covers them
  auto load_q = [&](int pr, int qb, u32x4 (&raw)[KS]) {
    const T* qp = (const T*)a.q + (int64_t)(pr / a.Hq) * a.qsb + (int64_t)(pr % a.Hq) * a.qsh;
    const int qrow = qb * QB + wave * FA_QW + r;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 16 + hh * 8;
      raw[ks] = gload16_async((qrow < a.Lq && d0 < D) ? qp + (int64_t)qrow * a.qsl + d0 : qp);
    }
  };
  auto finish_q = [&](int qb, u32x4 (&raw)[KS], v8 (&qf)[KS]) {
    const int qrow = qb * QB + wave * FA_QW + r;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      settle(raw[ks]);
      const v8 z = {};
      qf[ks] = (qrow < a.Lq && ks * 16 + hh * 8 < D) ? __builtin_bit_cast(v8, raw[ks]) : z;
    }
  };
  v8 qf[KS];
  u32x4 qn[KS];
  load_q(pair, qblk, qn);

  // ---- DMA pieces of this wave: K pieces wave + NW i (4 key rows each, i < KPW), as many of V; lane l
  // carries row 4 piece + l/16, LDS slot l%16 (source chunk = slot ^ swizzle).  Lanes whose chunk lies
  // past head_dim re-fetch chunk 0 of their row instead of being masked off: the pad slots then hold
  // finite copies of real values — K pad columns meet Q columns that are exact zeros, V pad columns
  // feed output rows that are never stored — and the issue path needs no EXEC juggling: per piece one
  // scalar add for M0 and the copy itself (the masked, per-piece-addressed form cost ~300 cycles a
  // piece and paced the whole QK^T phase: 2 100 cycles for 480 cycles of MFMA) ----
  static_assert(NPK % NW == 0, "K pieces must divide over the waves");
  constexpr int KPW = NPK / NW;
  const int dchunks = D / 8;
  int p_row[KPW];
  unsigned offK[KPW], offV[KPW], coK[KPW], coV[KPW];
#pragma unroll
  for (int i = 0; i < KPW; ++i) {
    const int row = 4 * (wave + NW * i) + (lane >> 4);
    int ck = (lane & 15) ^ (row & 15), cv = (lane & 15) ^ (4 * (row & 3));
    ck = ck < dchunks ? ck : 0;
    cv = cv < dchunks ? cv : 0;
    p_row[i] = row;
    coK[i] = (unsigned)(ck * 8 * (int)sizeof(T));
    coV[i] = (unsigned)(cv * 8 * (int)sizeof(T));
    offK[i] = (unsigned)(row * (int)a.ksl * (int)sizeof(T)) + coK[i];
    offV[i] = (unsigned)(row * (int)a.vsl * (int)sizeof(T)) + coV[i];
  }
  const unsigned m0_wave = (unsigned)(wave * 1024);
  // ONES: lanes whose slot is a pad chunk stay out of the copies (same lanes for every piece of a wave)
  const unsigned long long liveK = __builtin_amdgcn_ballot_w64(((lane & 15) ^ ((4 * wave + (lane >> 4)) & 15)) < dchunks);
  const unsigned long long liveV = __builtin_amdgcn_ballot_w64(((lane & 15) ^ (4 * ((lane >> 4) & 3))) < dchunks);
  if constexpr (ONES) {
    // pad chunks of every ring row: K zeros (they meet Q columns that are zero, but must be finite), V zeros and the
    // ones column.  16 bytes a chunk; rows are 256 bytes, chunk c of row `row` lives in slot c ^ swizzle(row).
    const v8 one_first = [] { v8 z = {}; z[0] = from_f32<T>(1.f); return z; }();
    const v8 zero8 = {};
    for (int i = tid; i < NS * FA_KB * 16; i += 512) {
      const int row = i >> 4, c = i & 15;
      if (c >= dchunks) {
        *(v8*)(fa_smem + row * ROWB + ((c ^ (row & 15)) << 4)) = zero8;
        *(v8*)(fa_smem + NS * TILEB + row * ROWB + ((c ^ (4 * (row & 3))) << 4)) = c == dchunks ? one_first : zero8;
      }
    }
    __syncthreads();
  }
  // tile `kt` of the sequence at kpx / vpx into ring stage `stage`: piece j (0..PPW-1; K first)
  struct TileCopy { const unsigned char *tk, *tv; unsigned mk; int left; };
  auto tile_copy = [&](const T* kpx, const T* vpx, int kt, int stage) {
    TileCopy c;
    c.tk = (const unsigned char*)(kpx + (int64_t)kt * FA_KB * a.ksl);
    c.tv = (const unsigned char*)(vpx + (int64_t)kt * FA_KB * a.vsl);
    c.mk = sK_off + (unsigned)(stage * TILEB) + m0_wave;
    c.left = a.Lk - kt * FA_KB;                 // rows of this tile that exist
    return c;
  };
  auto issue_piece = [&](const TileCopy& c, int j) {
    const bool isK = j < KPW;                   // compile-time after unrolling
    const int i = isK ? j : j - KPW;
    unsigned off = isK ? offK[i] : offV[i];
    if (c.left < FA_KB) {                       // last tile: rows past the end repeat the last key (masked later)
      const int rr = min(p_row[i], c.left - 1);
      off = (unsigned)(rr * (int)(isK ? a.ksl : a.vsl) * (int)sizeof(T)) + (isK ? coK[i] : coV[i]);
    }
    if constexpr (ONES)
      glds16_lanes(isK ? c.tk : c.tv, off, c.mk + (unsigned)(i * NW * 1024) + (isK ? 0u : (unsigned)(NS * TILEB)),
                         isK ? liveK : liveV);
    else
      ssdk::glds16(isK ? c.tk : c.tv, off, c.mk + (unsigned)(i * NW * 1024) + (isK ? 0u : (unsigned)(NS * TILEB)));
  };

  int k_rd[KS];
  const int kz = (hh ^ (r & 15)) << 4;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_rd[ks] = r * ROWB + ((32 * ks) ^ kz);
  const int q4 = (lane & 15) >> 2, p4 = lane & 3;
  int v_rd[DT];
  {
    const int cc = 2 * ((lane >> 4) & 1) + (p4 >> 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      v_rd[dt] = (4 * hh + q4) * ROWB + ((4 * (dt ^ q4) + cc) << 4) + (p4 & 1) * 8;
  }

  const int ntiles = (a.Lk + FA_KB - 1) / FA_KB;       // >= 2 (launcher)
  const T* kp = k_of(pair);
  const T* vp = v_of(pair);
  {
    const TileCopy c0 = tile_copy(kp, vp, 0, 0), c1 = tile_copy(kp, vp, 1, 1);
#pragma unroll
    for (int j = 0; j < PPW; ++j) issue_piece(c0, j);
#pragma unroll
    for (int j = 0; j < PPW; ++j) issue_piece(c1, j);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
  __builtin_amdgcn_s_barrier();
  finish_q(qblk, qn, qf);

  // the second-dispatched half of the work-group loses every VALU arbitration to the older half (priority, then
  // age): one static priority bump for it, no per-phase flips (-0.8 % here; the causal and fp8 kernels measured
  // 0.5 - 1 % SLOWER with it and do not have it)
  if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  int stage = 0;                   // ring stage of the current tile
#ifdef TV_FA_STAMP
  const bool st_on = blockIdx.x == 0 && (wave == 0 || wave == 4);
  __shared__ unsigned long long st_acc[2 * 16 * 8];
  if (tid < 256) st_acc[tid] = 0;
  __syncthreads();
  unsigned long long st_last = clock64();
#endif
  for (;;) {
    const int slot_n = slot + step;
    const int pair_n = xcd * a.ppx + slot_n / a.nqb, qblk_n = slot_n % a.nqb;
    const bool has_next = slot_n < nslots && pair_n < npairs;
    const T* kp_n = has_next ? k_of(pair_n) : kp;
    const T* vp_n = has_next ? v_of(pair_n) : vp;

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int kt = 0; kt < ntiles; ++kt) {
      const int kbase = kt * FA_KB;
      // the copy that goes out under this tile: two tiles on, in this block or the next one
      const bool wrap = kt + 2 >= ntiles;
      const bool ahead = !wrap || has_next;
      const T* pk = wrap ? kp_n : kp;
      const T* pv = wrap ? vp_n : vp;
      const int pf_kt = wrap ? kt + 2 - ntiles : kt + 2;
      const int pf_stage = stage == 0 ? 2 : stage - 1;
      const TileCopy pf = tile_copy(pk, pv, pf_kt, pf_stage);
      if (kt == ntiles - 1 && has_next) load_q(pair_n, qblk_n, qn);   // older than this tile's copies
      const unsigned stage_off = (unsigned)(stage * TILEB);
      const unsigned cK = sK_off + stage_off, cV = sV_off + stage_off;

      auto tile_body = [&](auto nt_c) __attribute__((always_inline)) {
      constexpr int NT = decltype(nt_c)::value;        // live 32-key sub-tiles of this tile
      f32x16 sacc[NT];
      v8 kf[NT][KS];
      unsigned kb[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kb[ks] = cK + (unsigned)k_rd[ks];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[t][ks] = Frag<T>::row_read(lds_at(kb[ks]) + t * (32 * ROWB));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[t][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          sacc[t] = Frag<T>::mfma(kf[t][ks], qf[ks], sacc[t]);
          // the copies of tile kt+2 go out between the MFMAs (placing them between the exponentials
          // of the softmax instead measured the same)
          constexpr int every = (NT * KS) / PPW > 0 ? (NT * KS) / PPW : 1;
          const int idx = t * KS + ks;
          if (ahead && idx % every == 0 && idx / every < PPW) issue_piece(pf, idx / every);
        }
      }
      if (ahead) {
#pragma unroll
        for (int i = (NT * KS) / ((NT * KS) / PPW > 0 ? (NT * KS) / PPW : 1); i < PPW; ++i)
          issue_piece(pf, i);
      }
      FSTAMP(0);
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      if (kbase + FA_KB > a.Lk) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = kbase + t * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            sacc[t][i] = key < a.Lk ? sacc[t][i] : -INFINITY;
          }
      }
      float tmax = -INFINITY;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; i += 2) tmax = fmaxf(fmaxf(tmax, sacc[t][i]), sacc[t][i + 1]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax * a.scale_log2);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
      const f32x2 sc2 = {a.scale_log2, a.scale_log2}, nm2 = {-m_use, -m_use};
      f32x2 ps2 = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const f32x2 e = __builtin_elementwise_fma(f32x2{sacc[t][i], sacc[t][i + 1]}, sc2, nm2);
          const f32x2 pp = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
          sacc[t][i] = pp[0];
          sacc[t][i + 1] = pp[1];
          if constexpr (!ONES) ps2 += pp;
        }
      if constexpr (!ONES) l_run = l_run * alpha + (ps2[0] + ps2[1]);
      if (__builtin_amdgcn_ballot_w64(m_new > m_run)) {
        const f32x2 al2 = {alpha, alpha};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const f32x2 v = f32x2{oacc[dt][i], oacc[dt][i + 1]} * al2;
            oacc[dt][i] = v[0];
            oacc[dt][i + 1] = v[1];
          }
      }
      m_run = m_new;
      FSTAMP(1);
      {
        unsigned vb[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vb[dt] = cV + (unsigned)v_rd[dt];
        auto read_v = [&](int s_, v4 (&lo)[DT], v4 (&hi)[DT]) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            lo[dt] = Frag<T>::tr_read(lds_at(vb[dt]) + s_ * (16 * ROWB));
            hi[dt] = Frag<T>::tr_read(lds_at(vb[dt]) + s_ * (16 * ROWB) + 8 * ROWB);
          }
        };
        v4 vlo[2][DT], vhi[2][DT];
        read_v(0, vlo[0], vhi[0]);
#pragma unroll
        for (int s_ = 0; s_ < 2 * NT; ++s_) {
          if (s_ < 2 * NT - 1) read_v(s_ + 1, vlo[(s_ + 1) & 1], vhi[(s_ + 1) & 1]);
          const int t = s_ >> 1, rb = (s_ & 1) * 8;
          v8 pf;
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[j] = from_f32<T>(sacc[t][rb + j]);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            v8 vf;
#pragma unroll
            for (int j = 0; j < 4; ++j) { vf[j] = vlo[s_ & 1][dt][j]; vf[4 + j] = vhi[s_ & 1][dt][j]; }
            oacc[dt] = Frag<T>::mfma(vf, pf, oacc[dt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      };
      // a frame's last tile is rarely full (729 keys = 7 x 96 + 57): sub-tiles that hold no key at all are left out of
      // the products, the exponentials and the V reads (the copies still bring the whole tile: the ring's shape is fixed)
      const int live_keys = a.Lk - kbase;
      // (not at 6 k-steps: three bodies beside 24 more fragment registers do not fit the 256)
      if (KT < 3 || KS > 5 || live_keys > 64 || a.notrim) tile_body(std::integral_constant<int, KT>{});
      else if (live_keys > 32) tile_body(std::integral_constant<int, (KT > 2 ? 2 : KT)>{});
      else tile_body(std::integral_constant<int, 1>{});
      FSTAMP(2);
      // the next tile (copied an iteration ago) must have landed; this iteration's copies stay in flight
      if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      FSTAMP(3);
      __builtin_amdgcn_s_barrier();
      FSTAMP(4);
      stage = stage == NS - 1 ? 0 : stage + 1;
    }

    // ---- normalise and store O[q][d] of this block; the stores complete under the next block's first tile ----
    {
      const int h = pair % a.Hq, b = pair / a.Hq;
      const int qrow = qblk * QB + wave * FA_QW + r;
      float l_tot;
      if constexpr (ONES) {
        // O^T row head_dim of query r: d-tile DT - 1, local row D - 32 (DT - 1) (a multiple of 8: element 4 (row / 8) of
        // the lane with hh = 0)
        const int rl8 = (D - 32 * (DT - 1)) >> 3;
        float lv = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) lv = g == rl8 ? oacc[DT - 1][4 * g] : lv;
        const float lo = __shfl_xor(lv, 32, 64);
        l_tot = hh ? lo : lv;
      } else {
        l_tot = l_run + __shfl_xor(l_run, 32, 64);
      }
      const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
      // A lane holds 4 output columns (8 bytes) per (d-tile, group g): columns 32 dt + 8 g + 4 hh .. + 3.  The halves
      // hh = 0 / 1 of a lane pair (r, r + 32) own the two halves of 8 consecutive columns, so one
      // v_permlane32_swap per dword gives each lane 16 contiguous bytes (groups g, g + 1 -> the lower lane
      // stores columns 16 gp .. + 7, the upper lane the next 8): half the store instructions — the epilogue of
      // a block is store-ISSUE bound.  Needs 16-byte aligned rows (a.o16); the last lone group stays 8 bytes.
      typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
      typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
      T* op = (T*)a.o + (int64_t)b * a.osb + (int64_t)min(qrow, a.Lq - 1) * a.osl + (int64_t)h * a.osh;
      const bool rowok = qrow < a.Lq;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int d_lo = dt * 32 + 16 * gp;          // first column of the pair of groups
          if (d_lo >= D) continue;
          v4 pa, pb;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            pa[j] = from_f32<T>(oacc[dt][4 * (2 * gp) + j] * inv);
            pb[j] = from_f32<T>(oacc[dt][4 * (2 * gp + 1) + j] * inv);
          }
          if (a.o16 && d_lo + 8 < D) {                 // (wave-uniform)
            const u32x2s ua = __builtin_bit_cast(u32x2s, pa), ub = __builtin_bit_cast(u32x2s, pb);
            const auto s0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
            const u32x4s w = {s0[0], s1[0], s0[1], s1[1]};
            if (rowok) *(u32x4s*)(op + d_lo + 8 * hh) = w;
          } else {
            if (rowok) *(v4*)(op + d_lo + 4 * hh) = pa;
            if (rowok && d_lo + 8 + 4 * hh < D) *(v4*)(op + d_lo + 8 + 4 * hh) = pb;
          }
        }
      if (rowok) {
        if (a.lse && hh == 0) {
          const float lse = l_tot > 0.f ? (m_run * 0.6931471805599453f + logf(l_tot)) : -INFINITY;
          a.lse[((int64_t)b * a.Hq + h) * a.Lq + qrow] = lse;
        }
      }
    }
    { const int kt = 15; FSTAMP(5); (void)kt; }
    if (!has_next) break;
    slot = slot_n; pair = pair_n; qblk = qblk_n; kp = kp_n; vp = vp_n;
    finish_q(qblk, qn, qf);        // loaded under the last tile, older than the copies its wait left in flight
  }
#ifdef TV_FA_STAMP
  __syncthreads();
  if (blockIdx.x == 0 && tid < 256) g_fa_stamps[tid] = st_acc[tid];
#endif
}


#include "attention_vit.hpp"

std::atomic<int> g_fa_variant{[] { const char* v = getenv("TV_FA_W64"); return v ? atoi(v) : 0; }()};

int tv_cu_count() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
      cus = 256;                               // MI355X
    return cus;
  }();
  return n;
}

template <typename T, int KS, int DT>
int launch_fa_d(const AttnArgs& a, int B, hipStream_t st) {
  constexpr int KT = FA_KT;
  constexpr int lds = 2 * 3 * 32 * KT * 256;   // K and V rings: 3 stages x 32 KT rows x 256 B
  hipError_t e;
  if (a.Lq > 128) {
    e = hipFuncSetAttribute((const void*)flash_fwd_kernel<T, KS, DT, 8, KT>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) {
      const int nqb = (a.Lq + 255) / 256;
      static const int xcd_ = [] { const char* v = getenv("TV_FA_XCD"); return v ? atoi(v) : 1; }();
      if (xcd_ && !a.causal && nqb <= 8 && (int64_t)B * a.Hq >= 64) {
        AttnArgs ax = a;
        ax.nqb = nqb;
        ax.nb = B;
        const int64_t pairs = ((int64_t)B * a.Hq + 7) / 8 * 8;
        ax.ppx = (int)(pairs / 8);
        ax.o16 = ((uintptr_t)a.o % 16 == 0 && a.osb % 8 == 0 && a.osl % 8 == 0 && a.osh % 8 == 0) ? 1 : 0;
        static const int stream_ = [] { const char* v = getenv("TV_FA_STREAM"); return v ? atoi(v) : 1; }();
        const int64_t slots = (int64_t)ax.ppx * nqb;          // query blocks per XCD
        if constexpr (KS > 6) {                                // head_dim 128: the second Q set does not fit the registers
          flash_fwd_kernel<T, KS, DT, 8, KT><<<dim3((unsigned)(pairs * nqb), 1, 1), 512, lds, st>>>(ax);
        } else if (stream_ && a.Lk > 32 * KT && slots > tv_cu_count() / 8) {
          // one resident work-group per CU streams its share of the query blocks
          static const int ones_ = [] { const char* v = getenv("TV_FA_ONES"); return v ? atoi(v) : 1; }();
          static const int trim_ = [] { const char* v = getenv("TV_FA_TRIM"); return v ? atoi(v) : 1; }();
          ax.notrim = !trim_;
          const dim3 grid_s((unsigned)(8 * (tv_cu_count() / 8)), 1, 1);
          const int variant = g_fa_variant.load(std::memory_order_relaxed);
          if ((variant == 0 || variant == 4) && std::is_same<T, bf16_t>::value && KS == 5 && DT == 3 && KT == 3 && ones_ && trim_ &&
              a.D % 8 == 0 && a.D < 32 * DT && a.Lk > 2 * 32 * KT) {
            // the generated tile loop (attention_vit.hpp): one wave per SIMD, 64 query rows a wave
            ax.notrim = 0;
            e = hipFuncSetAttribute((const void*)flash_fwd_vit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess) flash_fwd_vit_kernel<<<grid_s, 256, lds, st>>>(ax);
          } else if (ones_ && variant != 3 && a.D % 8 == 0 && a.D < 32 * DT) {
            e = hipFuncSetAttribute((const void*)flash_fwd_stream_kernel<T, KS, DT, KT, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess) flash_fwd_stream_kernel<T, KS, DT, KT, true><<<grid_s, 512, lds, st>>>(ax);
          } else {
            e = hipFuncSetAttribute((const void*)flash_fwd_stream_kernel<T, KS, DT, KT, false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess) flash_fwd_stream_kernel<T, KS, DT, KT, false><<<grid_s, 512, lds, st>>>(ax);
          }
        } else {
          flash_fwd_kernel<T, KS, DT, 8, KT><<<dim3((unsigned)(pairs * nqb), 1, 1), 512, lds, st>>>(ax);
        }
      } else {
        dim3 grid(nqb, a.Hq, B);
        flash_fwd_kernel<T, KS, DT, 8, KT><<<grid, 512, lds, st>>>(a);
      }
    }
  } else {
    e = hipFuncSetAttribute((const void*)flash_fwd_kernel<T, KS, DT, 4, KT>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) {
      dim3 grid((a.Lq + 127) / 128, a.Hq, B);
      flash_fwd_kernel<T, KS, DT, 4, KT><<<grid, 256, lds, st>>>(a);
    }
  }
  if (e != hipSuccess) {
    tv_set_error("flash_attn: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return TV_ERR_LAUNCH;
  }
  TV_LAUNCH_CHECK();
}

template <typename T>
int launch_fa(const AttnArgs& a, int B, hipStream_t st) {
  const int D = a.D;
  if (D <= 64) return launch_fa_d<T, 4, 2>(a, B, st);
  if (D <= 80) return launch_fa_d<T, 5, 3>(a, B, st);
  if (D <= 96) return launch_fa_d<T, 6, 3>(a, B, st);
  if (D <= 128) return launch_fa_d<T, 8, 4>(a, B, st);
  TV_UNSUPPORTED("flash_attn: headdim %d > 128", D);
}

}  // namespace

#ifdef TV_FA_STAMP
extern "C" int tv_fa_debug_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fa_stamps), sizeof(g_fa_stamps));
}
#endif

extern "C" void tv_flash_attn_set_variant(int variant) { g_fa_variant.store(variant, std::memory_order_relaxed); }

extern "C" int tv_flash_attn_fwd(const void* q, const void* k, const void* v, void* o, void* lse,
                                 int batch, int seqlen_q, int seqlen_k, int nheads_q,
                                 int nheads_kv, int headdim, int64_t q_stride_b,
                                 int64_t q_stride_l, int64_t q_stride_h, int64_t k_stride_b,
                                 int64_t k_stride_l, int64_t k_stride_h, int64_t v_stride_b,
                                 int64_t v_stride_l, int64_t v_stride_h, int64_t o_stride_b,
                                 int64_t o_stride_l, int64_t o_stride_h, float softmax_scale,
                                 int causal, int dtype, void* stream) {
  TV_CHECK_ARG((seqlen_q == 0 || (q && o)) && (seqlen_q == 0 || seqlen_k == 0 || (k && v)), "flash_attn: null pointer");
  TV_CHECK_ARG(batch > 0 && seqlen_q >= 0 && seqlen_k >= 0 && nheads_q > 0 && nheads_kv > 0 &&
                   nheads_q % nheads_kv == 0 && headdim > 0 && softmax_scale > 0.f,
               "flash_attn: bad sizes (or non-positive softmax scale)");
  if (dtype != TV_BF16 && dtype != TV_F16) TV_UNSUPPORTED("flash_attn: dtype must be bf16/f16");
  if (headdim % 8) TV_UNSUPPORTED("flash_attn: headdim %d not a multiple of 8", headdim);
  const int64_t strides[] = {q_stride_b, q_stride_l, q_stride_h, k_stride_b, k_stride_l,
                             k_stride_h, v_stride_b, v_stride_l, v_stride_h};
  for (int64_t s : strides)
    if (s % 8) TV_UNSUPPORTED("flash_attn: q/k/v strides must be multiples of 8 elements");
  if (o_stride_b % 4 || o_stride_l % 4 || o_stride_h % 4 || ((uintptr_t)o & 7))
    TV_UNSUPPORTED("flash_attn: o strides must be multiples of 4 elements");
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15))
    TV_UNSUPPORTED("flash_attn: q/k/v must be 16-byte aligned");
  // the K/V copies address a 64-key tile with 32-bit byte offsets from its first key
  if (128 * k_stride_l * 2 >= (1ll << 31) || 128 * v_stride_l * 2 >= (1ll << 31))
    TV_UNSUPPORTED("flash_attn: k/v row stride too large");
  if (seqlen_q == 0) return TV_OK;
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.o = o; a.lse = (float*)lse;
  a.Lq = seqlen_q; a.Lk = seqlen_k; a.Hq = nheads_q; a.Hkv = nheads_kv; a.D = headdim;
  a.qsb = q_stride_b; a.qsl = q_stride_l; a.qsh = q_stride_h;
  a.ksb = k_stride_b; a.ksl = k_stride_l; a.ksh = k_stride_h;
  a.vsb = v_stride_b; a.vsl = v_stride_l; a.vsh = v_stride_h;
  a.osb = o_stride_b; a.osl = o_stride_l; a.osh = o_stride_h;
  a.scale_log2 = softmax_scale * 1.4426950408889634f;
  a.causal = causal;
  a.nqb = 0;
  a.nb = batch;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TV_BF16) return launch_fa<bf16_t>(a, batch, st);
  return launch_fa<f16_t>(a, batch, st);
}
