// ViT attention (many short non-causal sequences, bf16, head_dim 65 .. 80: the SigLIP frames) with the key-tile loop as
// generated instruction streams (attention_vit_tile.inc <- devtools/gen_fa_vit.py).  Included by attention.hip inside its
// anonymous namespace.
//
// Same decomposition as flash_fwd_stream_kernel — 256 resident work-groups, one per CU, each walking the 256-row query
// blocks of its XCD's (frame, head) pairs; K / V tiles of 96 keys by LDS-DMA into a ring of 3 (source-side swizzle, pad
// lanes switched off, the ones column in V's pad chunk gives P's row sums on the matrix pipe); S^T = K Q^T so that the
// softmax statistics are per lane; O^T += V^T P^T — but ONE wave per SIMD with 64 query rows (units X, Y), and the loop
// software-pipelined across tiles: statement i = softmax of tile i (vector pipe) beside P.V of tile i - 1 and Q.K^T of tile
// i + 1 (matrix pipe), K / V fragment reads shared by the two units.  The compiled kernel runs its two waves per SIMD
// phase-aligned (a barrier a tile): every wave multiplies, then every wave exponentiates; here the pipes overlap inside a
// wave.  The exponentials use a LAZY maximum (it follows the true running maximum when that has grown by more than 2^8),
// so the O accumulators, which live in accumulation registers, are rescaled only then (a block's first tile sets it
// without a rescale: O is zero).  Copies: a statement issues K of tile i + 3 and V of tile i + 1 (the next block's tiles
// near the end of a block: the ring never drains) and waits for the previous statement's.
#include "attention_vit_tile.inc"

#define FAV_ACC_WRITE(idx, val) asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(val), "n"(idx))
#define FAV_ACC_READ(dst, idx) asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(dst) : "n"(idx))

__global__ __launch_bounds__(256) void flash_fwd_vit_kernel(AttnArgs a) {
  typedef bf16_t T;
  typedef Frag<T>::v8 v8;
  typedef Frag<T>::v4 v4;
  constexpr int KS = 5, DT = 3, NW = 4, FA_KB = 96, ROWB = 256, TILEB = FA_KB * ROWB, NS = 3, QB = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char fa_smem[];
  const unsigned sK_off = (unsigned)(uintptr_t)(lds_u8*)fa_smem, sV_off = sK_off + NS * TILEB;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int D = a.D, dchunks = D / 8;
  const int xcd = blockIdx.x & 7, step = gridDim.x >> 3;
  const int nslots = a.ppx * a.nqb, npairs = a.nb * a.Hq;
  int slot = blockIdx.x >> 3;
  int pair = xcd * a.ppx + slot / a.nqb, qblk = slot % a.nqb;
  if (slot >= nslots || pair >= npairs) return;
  const int gq = a.Hq / a.Hkv;
  auto k_of = [&](int pr) { return (const T*)a.k + (int64_t)(pr / a.Hq) * a.ksb + (int64_t)((pr % a.Hq) / gq) * a.ksh; };
  auto v_of = [&](int pr) { return (const T*)a.v + (int64_t)(pr / a.Hq) * a.vsb + (int64_t)((pr % a.Hq) / gq) * a.vsh; };
  auto q_of = [&](int pr) { return (const T*)a.q + (int64_t)(pr / a.Hq) * a.qsb + (int64_t)(pr % a.Hq) * a.qsh; };

  // ---- the ring starts as finite values everywhere (a statement may multiply a stage that no copy has filled yet by a
  // zero P): zeros, and 1.0 in the first element of V's pad chunk `dchunks` (the ones column)
  {
    const v8 one_first = [] { v8 z = {}; z[0] = from_f32<T>(1.f); return z; }();
    const v8 zero8 = {};
    for (int i = tid; i < NS * FA_KB * 16; i += 256) {
      const int row = i >> 4, c = i & 15;
      *(v8*)(fa_smem + row * ROWB + (c << 4)) = zero8;
      *(v8*)(fa_smem + NS * TILEB + row * ROWB + (c << 4)) = (c == (dchunks ^ (4 * (row & 3)))) ? one_first : zero8;
    }
    __syncthreads();
  }

  // ---- lane constants of the copies: piece i of a wave = rows row0 + 16 i of the tile, LDS slot lane % 16 of row lane / 16
  const int row0 = 4 * wave + (lane >> 4);
  const int ckr = (lane & 15) ^ (row0 & 15), cvr = (lane & 15) ^ (4 * (row0 & 3));
  const unsigned kco = (unsigned)((ckr < dchunks ? ckr : 0) * 16), vco = (unsigned)((cvr < dchunks ? cvr : 0) * 16);
  const unsigned long long klive = __builtin_amdgcn_ballot_w64(ckr < dchunks), vlive = __builtin_amdgcn_ballot_w64(cvr < dchunks);
  const unsigned kstride = (unsigned)(a.ksl * (int)sizeof(T)), vstride = (unsigned)(a.vsl * (int)sizeof(T));
  const unsigned koff0 = (unsigned)row0 * kstride + kco, voff0 = (unsigned)row0 * vstride + vco;      // whole tiles: one lane offset,
  const unsigned kstep = 16 * kstride, vstep = 16 * vstride;                                           // the source pointer walks 16 rows a piece
  // ---- lane parts of the fragment reads (as in flash_fwd_stream_kernel)
  const int kz = (hh ^ (r & 15)) << 4;
  unsigned kr[KS], vr[DT];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kr[ks] = (unsigned)(r * ROWB + ((32 * ks) ^ kz));
  {
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, cc = 2 * ((lane >> 4) & 1) + (p4 >> 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vr[dt] = (unsigned)((4 * hh + q4) * ROWB + ((4 * (dt ^ q4) + cc) << 4) + (p4 & 1) * 8);
  }
  const int ntiles = (a.Lk + FA_KB - 1) / FA_KB;       // >= 3 (launcher)
  const int64_t ktile = (int64_t)FA_KB * a.ksl, vtile = (int64_t)FA_KB * a.vsl;
  // byte offsets of this lane's Q^T fragments: unit u = rows wave 64 + 32 u + r; k-step ks = elements 16 ks + 8 hh (.. + 7); a
  // k-step past head_dim re-reads elements 0 .. 7 (finite; they meet K's zero pad chunks)
  auto q_off = [&](int qb, int u, bool last) {
    const int qrow = min(qb * QB + wave * 64 + 32 * u + r, a.Lq - 1);
    const int d0 = last ? ((64 + 8 * hh < D) ? 64 + 8 * hh : 0) : 8 * hh;
    return (unsigned)((qrow * (int)a.qsl + d0) * (int)sizeof(T));
  };

  // ---- first block: its first copies and its Q from here
  const T* kp = k_of(pair);
  const T* vp = v_of(pair);
  auto copy_tile = [&](const T* base, int kt, int stage, bool isK) {
    const unsigned char* src = (const unsigned char*)ssdk::uniform_ptr(base + (int64_t)kt * (isK ? ktile : vtile));
    const int left1 = a.Lk - kt * FA_KB - 1;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int rr = min(row0 + 16 * i, left1);
      glds16_lanes(src, (unsigned)rr * (isK ? kstride : vstride) + (isK ? kco : vco),
                   (isK ? sK_off : sV_off) + (unsigned)(stage * TILEB) + (unsigned)((wave + NW * i) * 1024), isK ? klive : vlive);
    }
  };
  copy_tile(kp, 0, 0, true);
  copy_tile(kp, 1, 1, true);
  copy_tile(kp, 2, 2, true);
  copy_tile(vp, 0, 0, false);
  // (the accumulation-register indices must be literals: written out per unit and k-step)
  {
    const unsigned char* qp = (const unsigned char*)q_of(pair);
#define FAV_QSET(u, ks, off)                                                        \
    do {                                                                            \
      const u32x4 raw = *(const u32x4*)(qp + (off));                                \
      FAV_ACC_WRITE(192 + 20 * (u) + 4 * (ks) + 0, raw[0]);                         \
      FAV_ACC_WRITE(192 + 20 * (u) + 4 * (ks) + 1, raw[1]);                         \
      FAV_ACC_WRITE(192 + 20 * (u) + 4 * (ks) + 2, raw[2]);                         \
      FAV_ACC_WRITE(192 + 20 * (u) + 4 * (ks) + 3, raw[3]);                         \
    } while (0)
    const unsigned ox = q_off(qblk, 0, false), ox4 = q_off(qblk, 0, true), oy = q_off(qblk, 1, false), oy4 = q_off(qblk, 1, true);
    FAV_QSET(0, 0, ox); FAV_QSET(0, 1, ox + 32); FAV_QSET(0, 2, ox + 64); FAV_QSET(0, 3, ox + 96); FAV_QSET(0, 4, ox4);
    FAV_QSET(1, 0, oy); FAV_QSET(1, 1, oy + 32); FAV_QSET(1, 2, oy + 64); FAV_QSET(1, 3, oy + 96); FAV_QSET(1, 4, oy4);
#undef FAV_QSET
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

#ifdef TV_FA_STAMP
  unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = clock64();
#define VSTAMP(ph) do { const unsigned long long n__ = clock64(); st[ph] += n__ - st_last; st_last = n__; } while (0)
#else
#define VSTAMP(ph) do {} while (0)
#endif
  int g = 0;                       // ring stage of the current block's tile 0
  // the walk over this XCD's slots without a division per block: slot -> (pair, qblk), pair -> (batch, head)
  const int sd = step / a.nqb, sm = step % a.nqb;
  int pb = pair / a.Hq, ph = pair % a.Hq;
  auto kv_base = [&](const void* base, int64_t sb, int64_t sh, int b_, int h_) {
    return (const T*)base + (int64_t)b_ * sb + (int64_t)(gq == 1 ? h_ : h_ / gq) * sh;
  };
  for (;;) {
    const int slot_n = slot + step;
    int qblk_n = qblk + sm, pair_n = pair + sd;
    if (qblk_n >= a.nqb) { qblk_n -= a.nqb; ++pair_n; }
    int pb_n = pb, ph_n = ph + (pair_n - pair);
    while (ph_n >= a.Hq) { ph_n -= a.Hq; ++pb_n; }
    const bool has_next = slot_n < nslots && pair_n < npairs;
    const T* kp_n = has_next ? kv_base(a.k, a.ksb, a.ksh, pb_n, ph_n) : kp;
    const T* vp_n = has_next ? kv_base(a.v, a.vsb, a.vsh, pb_n, ph_n) : vp;
    const int qb_n = has_next ? qblk_n : qblk;
    const void* qsrc = ssdk::uniform_ptr((const T*)a.q + (int64_t)(has_next ? pb_n : pb) * a.qsb + (int64_t)(has_next ? ph_n : ph) * a.qsh);
    const unsigned qox = q_off(qb_n, 0, false), qox4 = q_off(qb_n, 0, true), qoy = q_off(qb_n, 1, false), qoy4 = q_off(qb_n, 1, true);

    asm volatile(TV_FAV_BEGIN_ASM ::: TV_FAV_CLOBBERS);
    {
      const unsigned sk = sK_off + (unsigned)(g * TILEB);
      asm volatile(TV_FAV_PRO_ASM
                   :: [sk] "s"(sk), [kr0] "v"(kr[0]), [kr1] "v"(kr[1]), [kr2] "v"(kr[2]), [kr3] "v"(kr[3]), [kr4] "v"(kr[4])
                   : TV_FAV_CLOBBERS);
    }
    __builtin_amdgcn_s_barrier();
    VSTAMP(0);
    int sg = g;                    // stage of tile kt
    for (int kt = 0; kt < ntiles; ++kt) {
      const int s1 = sg == 2 ? 0 : sg + 1, s2 = sg == 0 ? 2 : sg - 1;      // stages of tiles kt + 1 and kt - 1 (= kt + 2)
      // the copies of this statement: K of tile kt + 3 -> stage of kt; V of tile kt + 1 -> stage s1
      const int kk = kt + 3, kv = kt + 1;
      const bool kwrap = kk >= ntiles, vwrap = kv >= ntiles;
      const int kkt = kwrap ? kk - ntiles : kk, kvt = vwrap ? kv - ntiles : kv;
      const void* ksrc = ssdk::uniform_ptr((kwrap ? kp_n : kp) + (int64_t)kkt * ktile);
      const void* vsrc = ssdk::uniform_ptr((vwrap ? vp_n : vp) + (int64_t)kvt * vtile);
      const int kleft = a.Lk - kkt * FA_KB - 1, vleft = a.Lk - kvt * FA_KB - 1;
      const unsigned km = sK_off + (unsigned)(sg * TILEB) + (unsigned)(wave * 1024);
      const unsigned vm = sV_off + (unsigned)(s1 * TILEB) + (unsigned)(wave * 1024);
      const unsigned sk = sK_off + (unsigned)(s1 * TILEB), sv = sV_off + (unsigned)(s2 * TILEB);
      unsigned flag;
      const bool full_copies = kleft >= FA_KB - 1 && vleft >= FA_KB - 1;
#define FAV_CLAMPED_OPS [kleft] "s"(kleft), [vleft] "s"(vleft), [kstride] "s"(kstride), [vstride] "s"(vstride), [row0] "v"(row0), [kco] "v"(kco), [vco] "v"(vco)
#define FAV_COMMON_OPS [sv] "s"(sv), [km] "s"(km), [vm] "s"(vm), [ksrc] "s"(ksrc), [vsrc] "s"(vsrc), [klive] "s"(klive), [vlive] "s"(vlive), \
                       [scale] "s"(a.scale_log2), [vr0] "v"(vr[0]), [vr1] "v"(vr[1]), [vr2] "v"(vr[2])
#define FAV_K_OPS [sk] "s"(sk), [kr0] "v"(kr[0]), [kr1] "v"(kr[1]), [kr2] "v"(kr[2]), [kr3] "v"(kr[3]), [kr4] "v"(kr[4])
      if (kt == 0) {
        asm volatile(TV_FAV_TILE0_ASM : [flag] "=s"(flag) : FAV_COMMON_OPS, FAV_K_OPS, FAV_CLAMPED_OPS : TV_FAV_CLOBBERS);
        VSTAMP(1);
      } else if (kt + 1 < ntiles) {
        if (full_copies)
          asm volatile(TV_FAV_TILE_ASM : [flag] "=s"(flag)
                       : FAV_COMMON_OPS, FAV_K_OPS, [koff] "v"(koff0), [voff] "v"(voff0), [kstep] "s"(kstep), [vstep] "s"(vstep) : TV_FAV_CLOBBERS);
        else
          asm volatile(TV_FAV_TILEC_ASM : [flag] "=s"(flag) : FAV_COMMON_OPS, FAV_K_OPS, FAV_CLAMPED_OPS : TV_FAV_CLOBBERS);
        VSTAMP(1);
      } else {
        const int live = a.Lk - kt * FA_KB;
        const int lim = live - 4 * hh;
        asm volatile(TV_FAV_LAST_ASM : [flag] "=s"(flag)
                     : FAV_COMMON_OPS, FAV_CLAMPED_OPS, [qsrc] "s"(qsrc), [live] "s"(live), [lim] "v"(lim), [qox] "v"(qox), [qox4] "v"(qox4),
                       [qoy] "v"(qoy), [qoy4] "v"(qoy4)
                     : TV_FAV_CLOBBERS);
        VSTAMP(3);
      }
#undef FAV_CLAMPED_OPS
#undef FAV_COMMON_OPS
#undef FAV_K_OPS
      if (flag) {
        asm volatile(TV_FAV_RESCALE_ASM ::: TV_FAV_CLOBBERS);
#ifdef TV_FA_STAMP
        st[7] += 1;
#endif
        VSTAMP(4);
      }
      __builtin_amdgcn_s_barrier();
      VSTAMP(2);
      sg = s1;
    }
    {
      const int sl = sg == 0 ? 2 : sg - 1;             // stage of the last tile
      const unsigned sv = sV_off + (unsigned)(sl * TILEB);
      asm volatile(TV_FAV_EPI_ASM :: [sv] "s"(sv), [vr0] "v"(vr[0]), [vr1] "v"(vr[1]), [vr2] "v"(vr[2]) : TV_FAV_CLOBBERS);
      VSTAMP(5);
    }

    // ---- normalise and store O[q][d] of this block (both units); the stores complete under the next block's first tiles
    if (a.o16 && !a.lse && (D == 72 || D == 80)) {
      const void* obase = ssdk::uniform_ptr((const T*)a.o + (int64_t)pb * a.osb + (int64_t)ph * a.osh);
      const int qx = qblk * QB + wave * 64 + r, qy = qx + 32;
      const unsigned rbx = (unsigned)(min(qx, a.Lq - 1) * (int)a.osl * (int)sizeof(T)), rby = (unsigned)(min(qy, a.Lq - 1) * (int)a.osl * (int)sizeof(T));
      const unsigned oax = rbx + 16 * hh, obx = rbx + 8 * hh, oay = rby + 16 * hh, oby = rby + 8 * hh;
      const unsigned long long maskx = __builtin_amdgcn_ballot_w64(qx < a.Lq), masky = __builtin_amdgcn_ballot_w64(qy < a.Lq);
      if (D == 72)
        asm volatile(TV_FAV_STORE72_ASM :: [obase] "s"(obase), [oax] "v"(oax), [obx] "v"(obx), [oay] "v"(oay), [oby] "v"(oby),
                     [maskx] "s"(maskx), [masky] "s"(masky) : TV_FAV_CLOBBERS);
      else
        asm volatile(TV_FAV_STORE80_ASM :: [obase] "s"(obase), [oax] "v"(oax), [obx] "v"(obx), [oay] "v"(oay), [oby] "v"(oby),
                     [maskx] "s"(maskx), [masky] "s"(masky) : TV_FAV_CLOBBERS);
    } else {
      float mu_x, mu_y;              // the maxima the exponentials used (first: the store phase may take these registers)
      asm volatile("v_mov_b32 %0, v[%c2]\n\tv_mov_b32 %1, v[%c3]" : "=v"(mu_x), "=v"(mu_y) : "n"(TV_FAV_MU_X), "n"(TV_FAV_MU_Y));
      const int h = ph, b = pb;
      typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
      typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
#define FAV_STORE_UNIT(U, MUREG)                                                                                       \
      do {                                                                                                             \
        float oc[DT][16];                                                                                              \
        FAV_READ16(oc[0], 48 * (U)); FAV_READ16(oc[1], 48 * (U) + 16); FAV_READ16(oc[2], 48 * (U) + 32);               \
        const int qrow = qblk * QB + wave * 64 + 32 * (U) + r;                                                         \
        const int rl8 = (D - 32 * (DT - 1)) >> 3;                                                                      \
        float lv = 0.f;                                                                                                \
        _Pragma("unroll") for (int gg = 0; gg < 4; ++gg) lv = gg == rl8 ? oc[DT - 1][4 * gg] : lv;                     \
        const float lo = __shfl_xor(lv, 32, 64);                                                                       \
        const float l_tot = hh ? lo : lv;                                                                              \
        const float inv = __builtin_amdgcn_rcpf(l_tot);          /* (as the generated store block; l >= 1) */                \
        T* op = (T*)a.o + (int64_t)b * a.osb + (int64_t)min(qrow, a.Lq - 1) * a.osl + (int64_t)h * a.osh;              \
        const bool rowok = qrow < a.Lq;                                                                                \
        _Pragma("unroll") for (int dt = 0; dt < DT; ++dt)                                                              \
          _Pragma("unroll") for (int gp = 0; gp < 2; ++gp) {                                                           \
            const int d_lo = dt * 32 + 16 * gp;                                                                        \
            if (d_lo >= D) continue;                                                                                   \
            v4 pa, pb;                                                                                                 \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                            \
              pa[j] = from_f32<T>(oc[dt][4 * (2 * gp) + j] * inv);                                                     \
              pb[j] = from_f32<T>(oc[dt][4 * (2 * gp + 1) + j] * inv);                                                 \
            }                                                                                                          \
            if (a.o16 && d_lo + 8 < D) {                                                                               \
              const u32x2s ua = __builtin_bit_cast(u32x2s, pa), ub = __builtin_bit_cast(u32x2s, pb);                   \
              const auto s0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);                            \
              const auto s1_ = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);                           \
              const u32x4s w = {s0[0], s1_[0], s0[1], s1_[1]};                                                         \
              if (rowok) *(u32x4s*)(op + d_lo + 8 * hh) = w;                                                           \
            } else {                                                                                                   \
              if (rowok) *(v4*)(op + d_lo + 4 * hh) = pa;                                                              \
              if (rowok && d_lo + 8 + 4 * hh < D) *(v4*)(op + d_lo + 8 + 4 * hh) = pb;                                 \
            }                                                                                                          \
          }                                                                                                            \
        if (rowok && a.lse && hh == 0)                                                                                 \
          a.lse[((int64_t)b * a.Hq + h) * a.Lq + qrow] = l_tot > 0.f ? ((MUREG) * 0.6931471805599453f + logf(l_tot)) : -INFINITY; \
      } while (0)
#define FAV_READ16(dst, base)                                                                                          \
      do {                                                                                                             \
        FAV_ACC_READ(dst[0], (base) + 0); FAV_ACC_READ(dst[1], (base) + 1); FAV_ACC_READ(dst[2], (base) + 2); FAV_ACC_READ(dst[3], (base) + 3);     \
        FAV_ACC_READ(dst[4], (base) + 4); FAV_ACC_READ(dst[5], (base) + 5); FAV_ACC_READ(dst[6], (base) + 6); FAV_ACC_READ(dst[7], (base) + 7);     \
        FAV_ACC_READ(dst[8], (base) + 8); FAV_ACC_READ(dst[9], (base) + 9); FAV_ACC_READ(dst[10], (base) + 10); FAV_ACC_READ(dst[11], (base) + 11); \
        FAV_ACC_READ(dst[12], (base) + 12); FAV_ACC_READ(dst[13], (base) + 13); FAV_ACC_READ(dst[14], (base) + 14); FAV_ACC_READ(dst[15], (base) + 15); \
      } while (0)
      FAV_STORE_UNIT(0, mu_x);
      FAV_STORE_UNIT(1, mu_y);
#undef FAV_STORE_UNIT
#undef FAV_READ16
    }
    VSTAMP(6);
#ifdef TV_FA_STAMP
    st[8] += 1;
#endif
    if (!has_next) break;
    g = sg;                        // (sg has advanced by ntiles stages)
    slot = slot_n; pair = pair_n; qblk = qblk_n; kp = kp_n; vp = vp_n; pb = pb_n; ph = ph_n;
  }
#ifdef TV_FA_STAMP
  if (blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 10; ++i) g_fa_stamps[i] = st[i];
#endif
}
