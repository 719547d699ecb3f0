// A1 / T3 / ViT: fused softmax attention forward on CDNA4 MFMA tiles.
//
// Structure (wave64, v_mfma_f32_32x32x16): a workgroup of 8 waves owns 256 query
// rows of one (batch, q-head) (4 waves / 128 rows for short queries); each wave keeps
// its 32 query rows as MFMA B fragments in registers for the whole kernel.  K/V tiles
// of 64 keys are double-buffered row-major in LDS and shared by the waves; the next
// tile's global loads are in flight while the current one is multiplied.  The score tile is computed
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
#include "common.hpp"

namespace {

constexpr int FA_QW = 32;                  // query rows per wave
constexpr int FA_KB = 64;                  // keys per tile

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  typedef bf16x8 v8; typedef bf16x4 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ v4 tr_read(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  }
};
template <> struct Frag<f16_t> {
  typedef f16x8 v8; typedef f16x4 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ v4 tr_read(const f16_t* p) {
    const s16x4 raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
    return __builtin_bit_cast(f16x4, raw);
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
};

// KS = ceil(D/16) k-steps of QK^T, DT = ceil(D/32) d-tiles of PV, NW waves per workgroup.
// K/V tiles are double-buffered in LDS: the global loads of tile j+1 are issued before the
// math of tile j and land in registers while it runs; they are written to the other buffer
// after it, so there is ONE barrier per tile and HBM/L2 latency hides under the MFMAs.
template <typename T, int KS, int DT, int NW>
__global__ __launch_bounds__(NW * 64) void flash_fwd_kernel(AttnArgs a) {
  typedef typename Frag<T>::v8 v8;
  typedef typename Frag<T>::v4 v4;
  constexpr int THREADS = NW * 64;
  constexpr int QB = NW * FA_QW;     // query rows per workgroup
  constexpr int DKP = KS * 16;       // padded K row (elements)
  constexpr int DVP = DT * 32;       // padded V row
  constexpr int KSTR = DKP + 8;      // +16 B: conflict-free ds_read_b128 across rows
  // V row stride == 16 or 48 dwords (mod 64): the 4 rows of a tr-read block land
  // on disjoint bank quarters
  constexpr int VSTR = (DVP % 64 == 32) ? DVP : DVP + 32;
  constexpr int KCH = DKP / 8, VCH = DVP / 8;           // 16-byte chunks per row
  constexpr int NKR = (FA_KB * KCH + THREADS - 1) / THREADS;
  constexpr int NVR = (FA_KB * VCH + THREADS - 1) / THREADS;
  __shared__ __attribute__((aligned(16))) T sK[2][FA_KB * KSTR];
  __shared__ __attribute__((aligned(16))) T sV[2][FA_KB * VSTR];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, hh = lane >> 5;
  // causal: launch the heaviest (last) query blocks first
  const int qblk = a.causal ? (gridDim.x - 1 - blockIdx.x) : blockIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  const int hk = h / (a.Hq / a.Hkv);
  const int q0 = qblk * QB + wave * FA_QW;      // first query row of this wave
  const int qrow = q0 + r;                      // this lane's query row
  const int D = a.D;
  const int shift = a.Lk - a.Lq;                // bottom-right causal alignment

  const T* qp = (const T*)a.q + (int64_t)b * a.qsb + (int64_t)h * a.qsh;
  const T* kp = (const T*)a.k + (int64_t)b * a.ksb + (int64_t)hk * a.ksh;
  const T* vp = (const T*)a.v + (int64_t)b * a.vsb + (int64_t)hk * a.vsh;

  // Q^T fragments (B operand): lane (r,hh) holds Q[qrow][16ks + 8hh + j]
  v8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int d0 = ks * 16 + hh * 8;
    v8 z = {};
    qf[ks] = (qrow < a.Lq && d0 < D) ? *(const v8*)(qp + (int64_t)qrow * a.qsl + d0) : z;
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

  // ---- staging: global -> registers (issue) ... registers -> LDS (commit) ----
  v8 rk[NKR], rv[NVR];
  auto stage_issue = [&](int kt) {
    const int kbase = kt * FA_KB;
#pragma unroll
    for (int i = 0; i < NKR; ++i) {
      const int idx = tid + i * THREADS;
      const int row = idx / KCH, c = idx % KCH;
      const int key = kbase + row;
      v8 z = {};
      rk[i] = (idx < FA_KB * KCH && key < a.Lk && c * 8 < D)
                  ? *(const v8*)(kp + (int64_t)key * a.ksl + c * 8) : z;
    }
#pragma unroll
    for (int i = 0; i < NVR; ++i) {
      const int idx = tid + i * THREADS;
      const int row = idx / VCH, c = idx % VCH;
      const int key = kbase + row;
      v8 z = {};
      rv[i] = (idx < FA_KB * VCH && key < a.Lk && c * 8 < D)
                  ? *(const v8*)(vp + (int64_t)key * a.vsl + c * 8) : z;
    }
  };
  auto stage_commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NKR; ++i) {
      const int idx = tid + i * THREADS;
      if (idx < FA_KB * KCH) *(v8*)(sK[buf] + (idx / KCH) * KSTR + (idx % KCH) * 8) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < NVR; ++i) {
      const int idx = tid + i * THREADS;
      if (idx < FA_KB * VCH) *(v8*)(sV[buf] + (idx / VCH) * VSTR + (idx % VCH) * 8) = rv[i];
    }
  };

  if (ntiles > 0) {
    stage_issue(0);
    stage_commit(0);
  }
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int kbase = kt * FA_KB;
    const int buf = kt & 1;
    const bool more = kt + 1 < ntiles;
    if (more) stage_issue(kt + 1);
    const T* cK = sK[buf];
    const T* cV = sV[buf];

    // wave-uniform skip of tiles entirely above this wave's causal diagonal
    const int wave_q_last = q0 + FA_QW - 1;
    if (!(a.causal && kbase > wave_q_last + shift)) {
      // ---- S^T = K . Q^T  (2 key sub-tiles of 32) ----
      f32x16 sacc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[t][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const v8 kf = *(const v8*)(cK + (t * 32 + r) * KSTR + ks * 16 + hh * 8);
          sacc[t] = Frag<T>::mfma(kf, qf[ks], sacc[t]);
        }
      }
      // ---- mask, running max on the raw scores (the scale is positive, so it commutes with
      // max), then p = 2^(s*scale - m) as one FMA + v_exp per score; packed fp32 math ----
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const bool need_mask = (kbase + FA_KB > a.Lk) || (a.causal && kbase + FA_KB - 1 > q0 + shift);
      if (need_mask) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = kbase + t * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            const bool ok = key < a.Lk && (!a.causal || key <= qrow + shift);
            sacc[t][i] = ok ? sacc[t][i] : -INFINITY;
          }
      }
      float tmax = -INFINITY;
#pragma unroll
      for (int t = 0; t < 2; ++t)
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
      for (int t = 0; t < 2; ++t)
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

      // ---- O^T += V^T . P^T over 4 k-steps of 16 keys ----
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int t = s >> 1, rb = (s & 1) * 8;
        v8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = from_f32<T>(sacc[t][rb + j]);
        // element j of this lane is key row 16s + 8(j>>2) + 4hh + (j&3) of the tile
        const int key0 = s * 16 + 4 * hh;
        const int q4 = (lane & 15) >> 2, p4 = lane & 3;
        const int cb = 16 * ((lane >> 4) & 1);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const T* base = cV + (key0 + q4) * VSTR + dt * 32 + cb + 4 * p4;
          const v4 lo = Frag<T>::tr_read(base);
          const v4 hi = Frag<T>::tr_read(base + 8 * VSTR);
          v8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
          oacc[dt] = Frag<T>::mfma(vf, pf, oacc[dt]);
        }
      }
    }
    if (more) stage_commit(buf ^ 1);
    __syncthreads();
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

template <typename T, int KS, int DT>
int launch_fa_d(const AttnArgs& a, int B, hipStream_t st) {
  if (a.Lq > 128) {
    dim3 grid((a.Lq + 255) / 256, a.Hq, B);
    flash_fwd_kernel<T, KS, DT, 8><<<grid, 512, 0, st>>>(a);
  } else {
    dim3 grid((a.Lq + 127) / 128, a.Hq, B);
    flash_fwd_kernel<T, KS, DT, 4><<<grid, 256, 0, st>>>(a);
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
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TV_BF16) return launch_fa<bf16_t>(a, batch, st);
  return launch_fa<f16_t>(a, batch, st);
}
