// V1/V2: ViT patch embedding (Conv2d / Conv3d with kernel == stride == patch) as an
// im2col-free MFMA GEMM:  out[m, n] = sum_k W[n, k] * patch(m)[k] + bias[n] + pos[m % P, n]
// with m = (frame, py, px), k = (c, dy, dx).  Patch pixels are gathered straight
// from the NCHW frame into an LDS tile whose (c,dy) rows are padded from `patch`
// to 16 columns, so every MFMA k-step of 16 is one image row of one patch; the
// weight tile gets the same padding on the fly.  The product is formed as
// W . patches^T (32x32x16 MFMA) so the patch index sits on the lane and each lane
// stores 4 consecutive output channels.
// Reference: timm PatchEmbed via TimmViTBackbone (base_vision.py:146-170,:274-278);
// InternVideo2 PatchEmbed Conv3d k=s=(1,14,14) (vit_scale_clean.py:445-461).
#include "common.hpp"

namespace {

constexpr int PE_TM = 128, PE_TN = 128;
constexpr int PE_RPS = 6;                 // (c,dy) rows per K step
constexpr int PE_KS = PE_RPS * 16;        // 96
constexpr int PE_STR = PE_KS + 8;         // LDS row stride (elements)
constexpr int PE_THREADS = 256;

struct PeArgs {
  const void *pix, *w, *wp, *bias, *pos;   // wp: packed weight [dout][rows_pad][16] or null
  void* out;
  int64_t M;                // frames * gh * gw
  int cin, H, W, p, gh, gw, dout;
  int fpg;                  // frames per group (T for Conv3d layout, else 1)
  int rows_pad;             // (c,dy) rows of the packed weight (multiple of PE_RPS)
  int64_t group_stride, frame_stride, chan_stride;
};

template <typename T> struct PeMma;
template <> struct PeMma<bf16_t> {
  typedef bf16x8 v8; typedef bf16x4 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct PeMma<f16_t> {
  typedef f16x8 v8; typedef f16x4 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// Conv weight (dout, cin*p*p) -> [dout][rows_pad][16]: every (c,dy) row of p taps padded to one
// MFMA k-step of 16 with zeros, rows padded to a multiple of PE_RPS.  Without it every lane walks
// its own 2*cin*p*p-byte weight row with 4-byte loads (64 cache lines per load instruction).
template <typename T>
__global__ void pe_pack_weight_kernel(const T* __restrict__ w, T* __restrict__ wp, int dout, int rows,
                                      int rows_pad, int p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one (n, row, tap)
  if (i >= (int64_t)dout * rows_pad * 16) return;
  const int t = (int)(i % 16), row = (int)((i / 16) % rows_pad);
  const int64_t n = i / (16 * (int64_t)rows_pad);
  wp[i] = (row < rows && t < p) ? w[n * rows * p + (int64_t)row * p + t] : from_f32<T>(0.f);
}

template <typename T, bool PACKED, int HP>
__global__ __launch_bounds__(PE_THREADS) void patch_embed_kernel(PeArgs a) {
  typedef typename PeMma<T>::v8 v8;
  typedef typename PeMma<T>::v4 v4;
  __shared__ __attribute__((aligned(16))) T sP[PE_TM * PE_STR];  // patches  [m][k]
  __shared__ __attribute__((aligned(16))) T sW[PE_TN * PE_STR];  // weights  [n][k]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;       // 2x2 waves, 64x64 each
  // XCD-aware tile order: work-group id -> XCD id % 8 (round-robin dispatch).  An XCD walks the
  // channel tiles of one patch tile back to back, so the pixels of a patch tile are fetched from
  // HBM once and re-read from that XCD's L2 (with channel tiles on grid.y they came back ~dout/128
  // times, launches apart).
  const int ntn = (a.dout + PE_TN - 1) / PE_TN;
  const int64_t wg = blockIdx.x, slot = wg >> 3;
  const int64_t m_tile = (slot / ntn) * 8 + (wg & 7);
  const int64_t m0 = m_tile * PE_TM;
  const int n0 = (int)(slot % ntn) * PE_TN;
  if (m0 >= a.M) return;
  const int npatch = a.gh * a.gw;
  const int rows_total = a.cin * a.p;            // (c,dy) rows
  const int kw = a.cin * a.p * a.p;              // weight row length

  f32x16 acc[2][2];  // [nt][mt]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const T* pix = (const T*)a.pix;
  const T* wgt = (const T*)a.w;

  // Staging.  A thread always stages the same patch (ml = tid % 128) and the same output
  // channel (nl = tid % 128): their base pointers are resolved once.  With an even patch size
  // (14, 16) and even image width every patch row starts 4-byte aligned, so a row is p/2 dword
  // loads and two 16-byte LDS writes instead of 16 scalar loads, selects and 2-byte writes.
  constexpr int PE_IT = PE_TM * PE_RPS / PE_THREADS;     // (row) items per thread and k-step
  static_assert(PE_TM == PE_TN && PE_THREADS % PE_TM == 0, "staging map");
  const int sl = tid % PE_TM, srl0 = tid / PE_TM;        // item j: row row0 + srl0 + j * (256/128)
  const int64_t sm_ = m0 + sl;
  const bool m_ok = sm_ < a.M;
  const T* pbase = pix;
  if (m_ok) {
    const int64_t f = sm_ / npatch;
    const int pi = (int)(sm_ % npatch);
    pbase = pix + (f / a.fpg) * a.group_stride + (f % a.fpg) * a.frame_stride +
            (int64_t)(pi / a.gw) * a.p * a.W + (pi % a.gw) * a.p;
  }
  const int sn = n0 + sl;
  const bool n_ok = sn < a.dout;
  const T* wbase = wgt + (int64_t)(n_ok ? sn : 0) * kw;
  const bool vec_ok = (a.p % 2 == 0) && (a.W % 2 == 0) && ((a.chan_stride | a.frame_stride | a.group_stride) % 2 == 0) &&
                      (((uintptr_t)pix | (uintptr_t)wgt) % 4 == 0);
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  struct Row { u32x4 lo, hi; };
  auto load_row = [&](const T* src, bool ok) {
    unsigned u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ok) {
      if (HP > 0) {            // launcher guarantees patch == 2*HP and dword-aligned rows
        const unsigned* s32 = reinterpret_cast<const unsigned*>(src);
#pragma unroll
        for (int d = 0; d < HP; ++d) u[d] = s32[d];
      } else if (vec_ok) {
        const unsigned* s32 = reinterpret_cast<const unsigned*>(src);
#pragma unroll
        for (int d = 0; d < 8; ++d)
          if (2 * d < a.p) u[d] = s32[d];
      } else {
        for (int dx = 0; dx < a.p; ++dx) {
          const unsigned v = __builtin_bit_cast(unsigned short, src[dx]);
          u[dx >> 1] |= v << (16 * (dx & 1));
        }
      }
    }
    return Row{u32x4{u[0], u[1], u[2], u[3]}, u32x4{u[4], u[5], u[6], u[7]}};
  };
  auto store_row = [&](const Row& rw, T* dst) {
    *(u32x4*)dst = rw.lo;
    *(u32x4*)(dst + 8) = rw.hi;
  };
  // The global loads of k-step i+1 are issued before the MFMAs of k-step i and written to LDS
  // after them: their latency hides under the MFMA phase instead of standing between two barriers.
  constexpr int WCH = PE_TN * (PE_KS / 8) / PE_THREADS;     // packed-weight chunks per thread
  Row prow[PE_IT], wrow[PACKED ? 1 : PE_IT];
  v8 wch[PACKED ? WCH : 1];
  auto fetch = [&](int row0) {
#pragma unroll
    for (int j = 0; j < PE_IT; ++j) {
      const int rl = srl0 + j * (PE_THREADS / PE_TM);
      const int row = row0 + rl;
      const int pp = HP > 0 ? 2 * HP : a.p;          // compile-time divisor on the common paths
      const int c = row / pp, dy = row - c * pp;
      const bool row_ok = row < rows_total;
      prow[j] = load_row(pbase + (int64_t)c * a.chan_stride + (int64_t)dy * a.W, m_ok && row_ok);
      if (!PACKED) wrow[j] = load_row(wbase + (int64_t)row * a.p, n_ok && row_ok);
    }
    if (PACKED) {   // 128 channels x 12 chunks of 16 bytes: consecutive lanes, consecutive chunks
      const T* wp = (const T*)a.wp;
#pragma unroll
      for (int j = 0; j < WCH; ++j) {
        const int q = tid + PE_THREADS * j;
        const int nl = q / (PE_KS / 8), ck = q % (PE_KS / 8);
        const v8 z = {};
        wch[j] = (n0 + nl < a.dout) ? *(const v8*)(wp + ((int64_t)(n0 + nl) * a.rows_pad + row0) * 16 + ck * 8) : z;
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < PE_IT; ++j) {
      const int rl = srl0 + j * (PE_THREADS / PE_TM);
      store_row(prow[j], sP + sl * PE_STR + rl * 16);
      if (!PACKED) store_row(wrow[j], sW + sl * PE_STR + rl * 16);
    }
    if (PACKED) {
#pragma unroll
      for (int j = 0; j < WCH; ++j) {
        const int q = tid + PE_THREADS * j;
        *(v8*)(sW + (q / (PE_KS / 8)) * PE_STR + (q % (PE_KS / 8)) * 8) = wch[j];
      }
    }
  };

  fetch(0);
  for (int row0 = 0; row0 < rows_total; row0 += PE_RPS) {
    __syncthreads();                 // the previous k-step's fragment reads are done
    commit();
    __syncthreads();
    if (row0 + PE_RPS < rows_total) fetch(row0 + PE_RPS);
#pragma unroll
    for (int ks = 0; ks < PE_RPS; ++ks) {
      v8 wf[2], pf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        wf[t] = *(const v8*)(sW + (wn * 64 + t * 32 + r) * PE_STR + ks * 16 + hh * 8);
        pf[t] = *(const v8*)(sP + (wm * 64 + t * 32 + r) * PE_STR + ks * 16 + hh * 8);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[nt][mt] = PeMma<T>::mfma(wf[nt], pf[mt], acc[nt][mt]);
    }
  }

  // epilogue: lane owns patch m (column), 4 consecutive channels per register group; bias and
  // pos-embed come in as 8-byte vectors (dout % 4 == 0)
  const T* bias = (const T*)a.bias;
  const T* pos = (const T*)a.pos;
  T* out = (T*)a.out;
  float bz[2][4][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + wn * 64 + nt * 32 + 8 * g + 4 * hh;
      v4 bv = {};
      if (bias && n < a.dout) bv = *(const v4*)(bias + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) bz[nt][g][j] = to_f32(bv[j]);
    }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int64_t m = m0 + wm * 64 + mt * 32 + r;
    if (m >= a.M) continue;
    const int pi = (int)(m % npatch);
    v4 pv[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + nt * 32 + 8 * g + 4 * hh;
        const v4 z = {};
        pv[nt][g] = (pos && n < a.dout) ? *(const v4*)(pos + (int64_t)pi * a.dout + n) : z;
      }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + nt * 32 + 8 * g + 4 * hh;
        if (n < a.dout) {
          v4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = from_f32<T>(acc[nt][mt][4 * g + j] + bz[nt][g][j] + to_f32(pv[nt][g][j]));
          *(v4*)(out + m * a.dout + n) = o;
        }
      }
  }
}

}  // namespace

extern "C" size_t tv_patch_embed_workspace_bytes(int dout, int cin, int patch) {
  if (dout <= 0 || cin <= 0 || patch <= 0) return 0;
  const int rows = cin * patch, rows_pad = (rows + PE_RPS - 1) / PE_RPS * PE_RPS;
  return (size_t)dout * rows_pad * 16 * 2;
}

// extended entry used by the Conv3d (B,C,T,H,W) layout; tv_patch_embed_fwd wraps it
extern "C" int tv_patch_embed_strided_fwd(const void* pixels, const void* weight,
                                          const void* bias, const void* pos, void* out,
                                          int frames, int cin, int height, int width, int patch,
                                          int dout, int frames_per_group, int64_t group_stride,
                                          int64_t frame_stride, int64_t chan_stride, int dtype,
                                          void* workspace, void* stream) {
  TV_CHECK_ARG(weight && (frames == 0 || (pixels && out)), "patch_embed: null pointer");
  TV_CHECK_ARG(frames >= 0 && cin > 0 && height > 0 && width > 0 && patch > 0 && dout > 0 &&
                   frames_per_group > 0,
               "patch_embed: bad sizes");
  if (patch > 16) TV_UNSUPPORTED("patch_embed: patch %d > 16", patch);
  if (dtype != TV_BF16 && dtype != TV_F16) TV_UNSUPPORTED("patch_embed: dtype must be bf16/f16");
  if (dout % 4 || ((uintptr_t)out & 7)) TV_UNSUPPORTED("patch_embed: dout must be a multiple of 4");
  if (workspace && ((uintptr_t)workspace & 15)) TV_UNSUPPORTED("patch_embed: workspace must be 16-byte aligned");
  if (((uintptr_t)bias | (uintptr_t)pos) & 7) TV_UNSUPPORTED("patch_embed: bias / pos must be 8-byte aligned");
  if (frames == 0) return TV_OK;
  PeArgs a;
  a.pix = pixels; a.w = weight; a.wp = workspace; a.bias = bias; a.pos = pos; a.out = out;
  a.cin = cin; a.H = height; a.W = width; a.p = patch;
  a.gh = height / patch; a.gw = width / patch; a.dout = dout;
  a.M = (int64_t)frames * a.gh * a.gw;
  a.fpg = frames_per_group; a.group_stride = group_stride; a.frame_stride = frame_stride;
  a.chan_stride = chan_stride;
  const int rows = cin * patch;
  a.rows_pad = (rows + PE_RPS - 1) / PE_RPS * PE_RPS;
  const int64_t ntm = (a.M + PE_TM - 1) / PE_TM, ntn = (dout + PE_TN - 1) / PE_TN;
  const int64_t nwg = (ntm + 7) / 8 * 8 * ntn;
  if (nwg >= (1ll << 31)) TV_UNSUPPORTED("patch_embed: too many tiles for one launch");
  dim3 grid((unsigned)nwg);
  hipStream_t st = (hipStream_t)stream;
  // dword-per-lane row loads need an even patch and image width and 4-byte aligned rows
  const bool vec = (patch % 2 == 0) && (width % 2 == 0) && ((chan_stride | frame_stride | group_stride) % 2 == 0) &&
                   (((uintptr_t)pixels | (uintptr_t)weight) % 4 == 0);
  const int hp = vec && (patch == 14 || patch == 16) ? patch / 2 : 0;
#define PE_LAUNCH(TT, PK)                                                                   \
  do {                                                                                      \
    if (hp == 7) patch_embed_kernel<TT, PK, 7><<<grid, PE_THREADS, 0, st>>>(a);             \
    else if (hp == 8) patch_embed_kernel<TT, PK, 8><<<grid, PE_THREADS, 0, st>>>(a);        \
    else patch_embed_kernel<TT, PK, 0><<<grid, PE_THREADS, 0, st>>>(a);                     \
  } while (0)
  if (workspace) {   // pack the weight (a few microseconds), then the GEMM reads it with 16-byte loads
    const int64_t n = (int64_t)dout * a.rows_pad * 16;
    const unsigned pg = (unsigned)((n + 255) / 256);
    if (dtype == TV_BF16) {
      pe_pack_weight_kernel<bf16_t><<<pg, 256, 0, st>>>((const bf16_t*)weight, (bf16_t*)workspace, dout, rows, a.rows_pad, patch);
      PE_LAUNCH(bf16_t, true);
    } else {
      pe_pack_weight_kernel<f16_t><<<pg, 256, 0, st>>>((const f16_t*)weight, (f16_t*)workspace, dout, rows, a.rows_pad, patch);
      PE_LAUNCH(f16_t, true);
    }
  } else {
    if (dtype == TV_BF16) PE_LAUNCH(bf16_t, false);
    else PE_LAUNCH(f16_t, false);
  }
#undef PE_LAUNCH
  TV_LAUNCH_CHECK();
}

extern "C" int tv_patch_embed_fwd(const void* pixels, const void* weight, const void* bias,
                                  const void* pos, void* out, int frames, int cin, int height,
                                  int width, int patch, int dout, int dtype, void* workspace,
                                  void* stream) {
  return tv_patch_embed_strided_fwd(pixels, weight, bias, pos, out, frames, cin, height, width,
                                    patch, dout, 1, (int64_t)cin * height * width, 0,
                                    (int64_t)height * width, dtype, workspace, stream);
}
