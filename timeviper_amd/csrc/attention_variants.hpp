// Two measured-slower variants of the ViT attention kernel ("What bounds the ViT attention"):
// flash_fwd_w64_kernel (4 waves x 64 query rows, one wave per SIMD) and flash_fwd_w32_kernel (8 waves x 32 rows on
// 16x16x32 tiles, the in-wave skewed pipeline at two waves per SIMD).  Both are parity-green and 15 - 20 % slower than
// flash_fwd_stream_kernel at head_dim 72; they are kept as measured negative results and are only compiled into
// builds with -DTV_FA_VARIANTS (TV_FA_VARIANTS=1 python -m timeviper_amd.build), where tv_flash_attn_set_variant(1 / 2)
// selects them.  Included by attention.hip inside its anonymous namespace.
#pragma once
// ---------------------------------------------------------------------------------------------------------------------
// ViT-sized heads (head_dim <= 80), non-causal, many short sequences: 64 QUERY ROWS PER WAVE at ONE wave per SIMD.
// The streaming kernel above runs two waves of 32 rows per SIMD; its three phases (QK^T, softmax, PV) then share the
// matrix pipe, the VALU and the LDS one after the other (6 000 cycles a tile for 2 100 of MFMA).  Here a
// work-group is 4 waves, a wave owns two 32-row query halves A and B and the whole 512-register file, and the halves
// run half a tile apart so that every segment of the instruction stream pairs the MFMAs of one half with the softmax of
// the other, placed gap by gap (sched_barrier fences):
//     segment 1 of tile n:  MFMA  S_B(n) = K(n) Q_B^T,  O_B += V(n-1) P_B(n-1)   |  VALU  P_A(n) = exp2(S_A(n) c - m_A)
//     segment 2 of tile n:  MFMA  S_A(n+1) = K(n+1) Q_A^T,  O_A += V(n) P_A(n)   |  VALU  P_B(n) = exp2(S_B(n) c - m_B)
// * the row sums ride on the matrix pipe: one extra MFMA per 16 keys with an all-ones A operand adds sum_k P[k][q] (of
//   the bf16-rounded P, the values the PV product uses) into an accumulator tile — no VALU adds, and the sums are
//   rescaled with O;
// * lazy rescale: the reference maximum m of a query row only moves when a tile's maximum exceeds it by more than
//   2^TV_FA_W64_LAZY (P stays <= 2^8, fp32 sums and bf16 P hold that without loss); the decision is taken between
//   two segments (wave-uniform branch), O is touched only then;
// * K / V tiles of 64 keys by LDS-DMA into rings of 4 (K three tiles ahead, V two), one barrier per tile, the
//   stream of tiles and the Q fragments run on across query blocks (the epilogue of half A sits one segment before
//   that of half B).
// Same LDS image, swizzles and fragment reads as the kernels above.  bf16 only.
#ifndef TV_FA_W64_LAZY
#define TV_FA_W64_LAZY 8.0f
#endif
#ifndef TV_FA_W64_KGAP
#define TV_FA_W64_KGAP 3       // the gap of segment 1 that carries the four K copies
#endif
#ifndef TV_FA_W64_VGAP
#define TV_FA_W64_VGAP 3       // the gap of segment 2 that carries the four V copies
#endif
__device__ __forceinline__ float w64_max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float w64_fma(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
  return d;
}
// (no inline-asm consumer may follow a transcendental directly: gfx950 needs a wait state between v_exp_f32 and a VALU
// that reads its result, and hipcc's hazard recogniser does not look into asm statements — an asm v_cvt_pk_bf16_f32
// behind the exponentials packed the exponent instead of the power; the conversions below are plain C++)
__device__ __forceinline__ unsigned w64_pk(float lo, float hi) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const b2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

// TV_FA_W64_ASM = 1: the MFMAs are asm statements with the accumulator TIED to its register ("+a" / "+v"); 0: builtins.
// With asm the hazard recogniser does not see the MFMAs: nothing may write an A / B operand with a VALU in the two
// instructions before one, and VALU reads of a result must sit >= 18 wait states behind it (`mfma_settle`).
#ifndef TV_FA_W64_ASM
#define TV_FA_W64_ASM 1
#endif
__device__ __forceinline__ void w64_mfma_v(f32x16& c, bf16x8 a, const ssdk::u32x4& b) {
#if TV_FA_W64_ASM
  asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#else
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}
__device__ __forceinline__ void w64_mfma_v0(f32x16& c, bf16x8 a, const ssdk::u32x4& b) {
#if TV_FA_W64_ASM
  asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
#else
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b), z, 0, 0, 0);
#endif
}
__device__ __forceinline__ void w64_mfma_a(f32x16& c, bf16x8 a, bf16x8 b) {
#if TV_FA_W64_ASM
  asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#else
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ void w64_mfma_a1(f32x16& c, const ssdk::u32x4& ones, bf16x8 b) {
#if TV_FA_W64_ASM
  asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(ones), "v"(b));
#else
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ void mfma_settle() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }

#ifndef W64_ABL
#define W64_ABL 0        // dev: timing ablations (wrong results): 1 no copies, 2 no tile maximum, 4 no softmax, 8 no epilogue stores
#endif
#ifdef TV_FA_STAMP
#define W64STAMP(ph) do { if (st_on) { const unsigned long long n__ = clock64(); st_acc[ph] += n__ - st_last; st_last = n__; } } while (0)
#else
#define W64STAMP(ph) do {} while (0)
#endif
template <int KS, int DT>
__global__ __launch_bounds__(256) void flash_fwd_w64_kernel(AttnArgs a) {
  typedef bf16_t T;
  typedef bf16x8 v8;
  typedef bf16x4 v4;
  typedef Frag<bf16_t> F;
  typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
  constexpr int NW = 4, KB = 64, ROWB = 256, TILEB = KB * ROWB, NS = 4, QB = 256, QW = 64;
  constexpr int NM = 2 * KS + 4 * (DT + 1);          // MFMAs per segment

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
  const float c_ = a.scale_log2;
  auto k_of = [&](int pr) { return (const T*)a.k + (int64_t)(pr / a.Hq) * a.ksb + (int64_t)((pr % a.Hq) / gq) * a.ksh; };
  auto v_of = [&](int pr) { return (const T*)a.v + (int64_t)(pr / a.Hq) * a.vsb + (int64_t)((pr % a.Hq) / gq) * a.vsh; };

  // ---- Q^T fragments of half X (B operand): lane (r, hh) holds Q[row][16 ks + 8 hh + j].  They come through a
  // wave-private 8 KiB staging tile in LDS (32 rows in the K tile's image and swizzle: 8 copy pieces, then five
  // ds_read_b128): every global access of the kernel is then a hand-counted LDS-DMA copy — an ordinary load here made
  // hipcc put its own vmcnt waits in front of the MFMAs of every segment, which drained the copy queue each time.
  // Rows past Lq repeat the last row (never stored); columns past head_dim are zeroed in the registers.
  const unsigned sQ_off = sV_off + NS * TILEB + (unsigned)(wave * 8192);
  auto copy_q = [&](int pr, int qb, int X) __attribute__((always_inline)) {
    const T* qp = (const T*)a.q + (int64_t)(pr / a.Hq) * a.qsb + (int64_t)(pr % a.Hq) * a.qsh;
    const void* sp = ssdk::uniform_ptr(qp);
    const int row0 = qb * QB + wave * QW + 32 * X;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = 4 * i + (lane >> 4);
      int cq = (lane & 15) ^ (rl & 15);
      cq = cq < D / 8 ? cq : 0;
      ssdk::glds16(sp, (unsigned)(min(row0 + rl, a.Lq - 1) * (int)a.qsl * 2 + cq * 16), sQ_off + 1024u * i);
    }
  };
  // ---- copies: this wave's four pieces of a tile = key rows 16 wave .. + 15 (4 rows a piece, lane l: row + l / 16,
  // LDS chunk l % 16, source chunk = chunk ^ swizzle; chunks past head_dim re-fetch chunk 0: finite pad).  Offsets are
  // relative to the tile's first key and pre-corrected for the instruction offsets of the grouped copy; a second set
  // serves the sequence's last tile, whose rows past the end repeat the last key (masked in the softmax).
  const int ntiles = (a.Lk + KB - 1) / KB;             // >= 4 (launcher)
  const int left_last = a.Lk - (ntiles - 1) * KB;
  const int dchunks = D / 8;
  unsigned offK[4], offV0;
  const unsigned dV = (unsigned)(4 * (int)a.vsl * 2 - 1024);       // piece i + 1 against piece i (V's swizzle repeats every 4 rows)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 16 * wave + 4 * i + (lane >> 4);
    int ck = (lane & 15) ^ (row & 15);
    ck = ck < dchunks ? ck : 0;
    offK[i] = (unsigned)(row * (int)a.ksl * 2 + ck * 16 - 1024 * i);
  }
  {
    const int row = 16 * wave + (lane >> 4);
    int cv = (lane & 15) ^ (4 * (row & 3));
    cv = cv < dchunks ? cv : 0;
    offV0 = (unsigned)(row * (int)a.vsl * 2 + cv * 16);
  }
  const unsigned m0_wave = (unsigned)(wave * 4096);
  // (the grouped copy adds 1024 j to the source address of piece j; in the sequence's last tile a clamped row may sit
  // below that, so that tile — one per query block — goes piece by piece with its offsets computed on the spot)
  auto tile_copy_k = [&](const T* base, int kt, int stage) __attribute__((always_inline)) {
    const void* sp = ssdk::uniform_ptr(base + (int64_t)kt * KB * a.ksl);
    const unsigned dst = sK_off + stage * TILEB + m0_wave;
    if (kt == ntiles - 1 && left_last < KB) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 16 * wave + 4 * i + (lane >> 4);
        int ck = (lane & 15) ^ (row & 15);
        ck = ck < dchunks ? ck : 0;
        ssdk::glds16(sp, (unsigned)(min(row, left_last - 1) * (int)a.ksl * 2 + ck * 16), dst + 1024u * i);
      }
    } else {
      ssdk::glds16x4(sp, offK[0], offK[1], offK[2], offK[3], dst);
    }
  };
  auto tile_copy_v = [&](const T* base, int kt, int stage) __attribute__((always_inline)) {
    const void* sp = ssdk::uniform_ptr(base + (int64_t)kt * KB * a.vsl);
    const unsigned dst = sV_off + stage * TILEB + m0_wave;
    if (kt == ntiles - 1 && left_last < KB) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 16 * wave + 4 * i + (lane >> 4);
        int cv = (lane & 15) ^ (4 * (row & 3));
        cv = cv < dchunks ? cv : 0;
        ssdk::glds16(sp, (unsigned)(min(row, left_last - 1) * (int)a.vsl * 2 + cv * 16), dst + 1024u * i);
      }
    } else {
      ssdk::glds16x4(sp, offV0, offV0 + dV, offV0 + 2 * dV, offV0 + 3 * dV, dst);
    }
  };

  int k_rd[KS];
  const int kz = (hh ^ (r & 15)) << 4;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_rd[ks] = r * ROWB + ((32 * ks) ^ kz);
  auto read_q = [&](int X, u32x4 (&qf)[KS]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      const u32x4 v = __builtin_bit_cast(u32x4, F::row_read(lds_at(sQ_off + (unsigned)k_rd[ks])));
      qf[ks] = (ks * 16 + hh * 8 < D) ? v : z;
    }
  };
  const int q4 = (lane & 15) >> 2, p4 = lane & 3;
  int v_rd[DT];
  {
    const int cc = 2 * ((lane >> 4) & 1) + (p4 >> 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      v_rd[dt] = (4 * hh + q4) * ROWB + ((4 * (dt ^ q4) + cc) << 4) + (p4 & 1) * 8;
  }

  // ---- state of the two halves
  f32x16 O[2][DT], Ls[2], S[2][2];
  u32x4 Qf[2][KS];            // Q fragments (bit patterns of 8 bf16)
  u32x4 Pf[2][4];             // P^T fragments of the tile in flight: k-step s = keys 16 s .. + 15 (in the accumulator's key order)
  float mref[2] = {-INFINITY, -INFINITY}, tmax[2] = {-INFINITY, -INFINITY};
  bool first[2] = {true, true};
  v8 kf[3];
  // four bf16 ones per dword, made opaque: a constant is re-materialised (v_mov) right in front of the asm MFMA that
  // reads it, inside the wait states a VALU write needs before an MFMA reads the register
  u32x4 ones_u;
  asm volatile("v_mov_b32 %0, 0x3f803f80\n\tv_mov_b32 %1, 0x3f803f80\n\tv_mov_b32 %2, 0x3f803f80\n\tv_mov_b32 %3, 0x3f803f80"
               : "=v"(ones_u[0]), "=v"(ones_u[1]), "=v"(ones_u[2]), "=v"(ones_u[3]));
#pragma unroll
  for (int X = 0; X < 2; ++X) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) O[X][dt][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) Ls[X][i] = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) Pf[X][s_] = u32x4{0u, 0u, 0u, 0u};
  }

  // ---- prologue: zero the rings (the first PV of half B multiplies P = 0 with whatever the slot holds), first tiles, Q
  {
    u32x4s* z = (u32x4s*)fa_smem;
    for (int i = tid; i < 2 * NS * TILEB / 16; i += 256) z[i] = u32x4s{0u, 0u, 0u, 0u};
  }
  __syncthreads();
  const T* kp = k_of(pair);
  const T* vp = v_of(pair);
  tile_copy_k(kp, 0, 0);
  tile_copy_k(kp, 1, 1);
  tile_copy_k(kp, 2, 2);
  tile_copy_v(vp, 0, 0);
  tile_copy_v(vp, 1, 1);
  copy_q(pair, qblk, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  read_q(0, Qf[0]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  copy_q(pair, qblk, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  read_q(1, Qf[1]);
  __builtin_amdgcn_s_barrier();

  auto kfrag = [&](unsigned cK, int m) __attribute__((always_inline)) {     // K fragment of QK^T MFMA m: sub-tile m & 1, k-step m >> 1
    return F::row_read(lds_at(cK + (unsigned)k_rd[m >> 1]) + (m & 1) * (32 * ROWB));
  };
  // tile maximum of half X (raw scores), both key halves of the lane pair
  auto tile_max = [&](int X) __attribute__((always_inline)) {
    float m = fmaxf(S[X][0][0], S[X][0][1]);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = (t == 0 ? 2 : 0); i < 16; i += 2) m = w64_max3(m, S[X][t][i], S[X][t][i + 1]);
    return fmaxf(m, __shfl_xor(m, 32, 64));
  };
  auto mask_tail = [&](int X, int kvalid) __attribute__((always_inline)) {   // keys >= kvalid of the tile -> -inf
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        S[X][t][i] = key < kvalid ? S[X][t][i] : -INFINITY;
      }
  };
  // between two segments: does half X's reference maximum have to move before its softmax?
  auto decide = [&](int X) __attribute__((always_inline)) {
    const float tm = tmax[X] * c_;
    if (__builtin_amdgcn_ballot_w64(tm > mref[X] + TV_FA_W64_LAZY)) {
      if (first[X]) {
        mref[X] = tm;
        first[X] = false;
      } else {
        const float mn = fmaxf(mref[X], tm);
        const float al = __builtin_amdgcn_exp2f(mref[X] - mn);
        mref[X] = mn;
        mfma_settle();
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) O[X][dt][i] *= al;
#pragma unroll
        for (int i = 0; i < 16; ++i) Ls[X][i] *= al;
      }
    }
  };

  // One segment: MFMAs of half X (scores of the tile in K slot cK, PV of its previous P with the tile in V slot cV),
  // elementwise softmax of half Y = 1 - X, tile maximum of X's new scores; COPY: 1 the K group, 2 the V group, 0 none.
  // cKn: K slot whose first two fragments the NEXT segment starts with.
  auto segment = [&](const int X, const bool masked, const int kvalid, unsigned cK, unsigned cV, unsigned cKn,
                     const T* cbase, int ckt, int cstage) __attribute__((always_inline)) {
    const int COPY = X == 1 ? 1 : 2;
    float ew[32], pw[32], mx = 0.f;
    const int Y = 1 - X;
    const float nm = -mref[Y];
    unsigned vb[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vb[dt] = cV + (unsigned)v_rd[dt];
    v4 vlo[2][DT], vhi[2][DT];
#pragma clang loop unroll(full)
    for (int m = 0; m < NM; ++m) {
      // ---- the MFMA of this gap
      if (m < 2 * KS) {
        const int t = m & 1, ks = m >> 1;
        if (ks == 0) w64_mfma_v0(S[X][t], kf[m % 3], Qf[X][ks]);
        else w64_mfma_v(S[X][t], kf[m % 3], Qf[X][ks]);
      } else {
        const int j = m - 2 * KS, s_ = j / (DT + 1), w = j % (DT + 1);
        const v8 pv = __builtin_bit_cast(v8, Pf[X][s_]);
        if (w < DT) {
          v8 vf;
#pragma unroll
          for (int e = 0; e < 4; ++e) { vf[e] = vlo[s_ & 1][w][e]; vf[4 + e] = vhi[s_ & 1][w][e]; }
          w64_mfma_a(O[X][w], vf, pv);
        } else {
          w64_mfma_a1(Ls[X], ones_u, pv);
        }
      }
      // ---- fillers
      if (!(W64_ABL & 32) && m + 2 < 2 * KS) kf[(m + 2) % 3] = kfrag(cK, m + 2);
      if (!(W64_ABL & 32) && m >= NM - 2) kf[m - (NM - 2)] = kfrag(cKn, m - (NM - 2));        // the next segment's first two
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          if (!(W64_ABL & 16) && m == 2 * KS - 4 + (DT + 1) * s_ + dt) {
            vlo[s_ & 1][dt] = F::tr_read(lds_at(vb[dt]) + s_ * (16 * ROWB));
            vhi[s_ & 1][dt] = F::tr_read(lds_at(vb[dt]) + s_ * (16 * ROWB) + 8 * ROWB);
          }
      // softmax of Y, one element = three stages a gap apart (no v_exp_f32 result is read in the gap that makes it: no
      // hazard nops): fma in gap f(e) = 3 e / 4, exp2 in f(e) + 1, the pair's bf16 pack in f(odd e) + 2 — at most two
      // of each per gap.  The empty asm statements pin every stage to its gap (LLVM otherwise sinks them to the use).
      if (!(W64_ABL & 4)) {
        // elements with f(e) = g:  (4 g + 2) / 3 <= e <= (4 g + 3) / 3
        if (m >= 2) {
#pragma unroll
          for (int e = (4 * (m - 2) + 2) / 3; e <= (4 * (m - 2) + 3) / 3; ++e)
            if (e < 32 && (e & 1)) {
              unsigned pk = w64_pk(pw[e - 1], pw[e]);
              asm volatile("" : "+v"(pk));
              Pf[Y][e >> 3][(e & 7) >> 1] = pk;
            }
        }
        if (m >= 1) {
#pragma unroll
          for (int e = (4 * (m - 1) + 2) / 3; e <= (4 * (m - 1) + 3) / 3; ++e)
            if (e < 32) {
              pw[e] = __builtin_amdgcn_exp2f(ew[e]);
              asm volatile("" : "+v"(pw[e]));
            }
        }
#pragma unroll
        for (int e = (4 * m + 2) / 3; e <= (4 * m + 3) / 3; ++e)
          if (e < 32) {
            ew[e] = w64_fma(S[Y][e >> 4][e & 15], c_, nm);
            asm volatile("" : "+v"(ew[e]));
          }
      }
      // tile maximum of X's new scores (complete two gaps behind the last QK^T MFMA): the 16 register pairs over the
      // gaps 2 KS + 2 .. NM - 1 (two in each of the first ones), one v_max3_f32 a pair
      if (!(W64_ABL & 2) && m >= 2 * KS + 2) {
        constexpr int G0 = 2 * KS + 2, NG = NM - G0, DBL = 16 - NG;      // DBL gaps take two pairs
        static_assert(NG >= 8 && NG <= 16, "tile-maximum schedule");
        const int g = m - G0;
        const int u0 = g < DBL ? 2 * g : g + DBL, u1 = g < DBL ? u0 + 2 : u0 + 1;
#pragma unroll
        for (int u = u0; u < u1; ++u) {
          const int t = u >> 3, i2 = 2 * (u & 7);
          if (u == 0) asm volatile("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(S[X][t][i2]), "v"(S[X][t][i2 + 1]));
          else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(S[X][t][i2]), "v"(S[X][t][i2 + 1]));
        }
      }
      if (!(W64_ABL & 1) && m == (COPY == 1 ? TV_FA_W64_KGAP : TV_FA_W64_VGAP)) {
        if (COPY == 1) tile_copy_k(cbase, ckt, cstage);
        else tile_copy_v(cbase, ckt, cstage);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (masked) {                          // (wave-uniform; one tile in a sequence; the asm keeps it a branch)
      asm volatile("; masked tail");
      mask_tail(X, kvalid);
      tmax[X] = tile_max(X);
    } else if (!(W64_ABL & 2)) {
      tmax[X] = fmaxf(mx, __shfl_xor(mx, 32, 64));
    }
  };

  // epilogue of half X of query block (pr, qb): O / l -> global, reset the half
  auto epilogue = [&](int X, int pr, int qb) __attribute__((always_inline)) {
    const int h = pr % a.Hq, b = pr / a.Hq;
    const int qrow = qb * QB + wave * QW + 32 * X + r;
    mfma_settle();
    const float l_tot = Ls[X][0];
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    T* op = (T*)a.o + (int64_t)b * a.osb + (int64_t)min(qrow, a.Lq - 1) * a.osl + (int64_t)h * a.osh;
    const bool rowok = qrow < a.Lq;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const int d_lo = dt * 32 + 16 * gp;
        if (d_lo >= D) continue;
        const u32x2s ua = {w64_pk(O[X][dt][8 * gp + 0] * inv, O[X][dt][8 * gp + 1] * inv), w64_pk(O[X][dt][8 * gp + 2] * inv, O[X][dt][8 * gp + 3] * inv)};
        const u32x2s ub = {w64_pk(O[X][dt][8 * gp + 4] * inv, O[X][dt][8 * gp + 5] * inv), w64_pk(O[X][dt][8 * gp + 6] * inv, O[X][dt][8 * gp + 7] * inv)};
        if (a.o16 && d_lo + 8 < D) {
          const auto s0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
          const u32x4s w = {s0[0], s1[0], s0[1], s1[1]};
          if (rowok && !(W64_ABL & 8)) *(u32x4s*)(op + d_lo + 8 * hh) = w;
        } else {
          if (rowok && !(W64_ABL & 8)) *(u32x2s*)(op + d_lo + 4 * hh) = ua;
          if (rowok && !(W64_ABL & 8) && d_lo + 8 + 4 * hh < D) *(u32x2s*)(op + d_lo + 8 + 4 * hh) = ub;
        }
      }
    if (rowok && a.lse && hh == 0)
      a.lse[((int64_t)b * a.Hq + h) * a.Lq + qrow] = l_tot > 0.f ? (mref[X] * 0.6931471805599453f + logf(l_tot)) : -INFINITY;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) O[X][dt][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) Ls[X][i] = 0.f;
    mref[X] = -INFINITY;
    first[X] = true;
  };

  // ---- segment 0 of the stream: S_A of tile 0
  {
    const unsigned cK = sK_off;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const v8 kx = kfrag(cK, 2 * ks + t);
        if (ks == 0) w64_mfma_v0(S[0][t], kx, Qf[0][ks]);
        else w64_mfma_v(S[0][t], kx, Qf[0][ks]);
      }
    }
    mfma_settle();
    tmax[0] = tile_max(0);
    kf[0] = kfrag(cK, 0);
    kf[1] = kfrag(cK, 1);
  }

  const bool tail = left_last < KB;        // the last tile of a sequence has masked keys
#ifdef TV_FA_STAMP
  const bool st_on = blockIdx.x == 0 && wave == 0;
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = clock64();
#endif
  int n = 0;                               // tile counter of the stream (ring slots)
  int pair_p = pair, qblk_p = qblk;        // half B's block (one segment behind at the seams)
  bool have_prev = false, valid = true;    // valid = false: the phantom block behind the last one (half B's last PV only)
  for (;;) {
    const int slot_n = slot + step;
    const int pair_n = xcd * a.ppx + slot_n / a.nqb, qblk_n = slot_n % a.nqb;
    const bool has_next = valid && slot_n < nslots && pair_n < npairs;
    const T* kp_n = has_next ? k_of(pair_n) : kp;
    const T* vp_n = has_next ? v_of(pair_n) : vp;
    bool done = false;
    for (int kt = 0; kt < ntiles; ++kt, ++n) {
      const bool last = kt == ntiles - 1;
      const unsigned cK0 = sK_off + (unsigned)((n & 3) * TILEB), cK1 = sK_off + (unsigned)(((n + 1) & 3) * TILEB);
      const unsigned cV0 = sV_off + (unsigned)((n & 3) * TILEB), cVm = sV_off + (unsigned)(((n + 3) & 3) * TILEB);
      // copies of this iteration: K three tiles ahead, V two (past the last block: harmless re-fetches)
      const bool kw = kt + 3 >= ntiles, vw = kt + 2 >= ntiles;
      const T* ck = kw ? kp_n : kp;
      const T* cv = vw ? vp_n : vp;
      const int ckt = kw ? kt + 3 - ntiles : kt + 3, vkt = vw ? kt + 2 - ntiles : kt + 2;
      // ---- segment 1
      W64STAMP(0);
      decide(0);
      if (last && has_next) copy_q(pair_n, qblk_n, 0);             // Q_A's last use was the previous segment
      W64STAMP(1);
      segment(1, last && tail, left_last, cK0, cVm, cK1, ck, ckt, (n + 3) & 3);
      W64STAMP(2);
      if (kt == 0 && have_prev) epilogue(1, pair_p, qblk_p);
      if (!valid) { done = true; break; }
      if (last && has_next) {        // the 8 Q pieces are older than this segment's 4 K pieces
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        read_q(0, Qf[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the staging tile is written again below
      }
      // ---- segment 2
      W64STAMP(3);
      decide(1);
      if (last && has_next) copy_q(pair_n, qblk_n, 1);             // Q_B's last use was segment 1
      W64STAMP(4);
      segment(0, kt == ntiles - 2 && tail, left_last, cK1, cV0, cK1, cv, vkt, (n + 2) & 3);
      W64STAMP(5);
      if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      W64STAMP(6);
      __builtin_amdgcn_s_barrier();
      W64STAMP(7);
      if (last) {
        epilogue(0, pair, qblk);
        if (has_next) {
          read_q(1, Qf[1]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    }
    if (done) break;
    pair_p = pair; qblk_p = qblk; have_prev = true;
    if (has_next) { slot = slot_n; pair = pair_n; qblk = qblk_n; kp = kp_n; vp = vp_n; }
    else valid = false;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the phantom segment's copies
#ifdef TV_FA_STAMP
  if (st_on && lane == 0) {
    for (int i = 0; i < 8; ++i) g_fa_stamps[i] = st_acc[i];
    g_fa_stamps[8] = (unsigned long long)n;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Variant 2: the same in-wave pipeline at TWO waves per SIMD — 8 waves x 32 query rows, a wave's two halves are 16 rows
// on v_mfma_f32_16x16x32_bf16 tiles (head_dim 65..80 = two k-steps of 32 and one of 16: no padding to 96), <= 256
// registers a wave.  flash_fwd_w64_kernel showed that at head_dim 72 a wave is ISSUE-bound when it mixes its MFMA and
// softmax streams; here two such streams share a SIMD's issue slots, and (unlike the streaming kernel) neither runs its
// phases in lock-step with the other.  Layouts (lc = lane & 15, kq = lane >> 4):
//   S^T tile [16 keys][16 q]   = K . Q^T: A = K rows (key lc, d = 32 ks + 8 kq ..+7; tail d = 64 + 4 kq ..+3), B = Q rows
//                                likewise; accumulator register r of lane (q = lc, kq) = key 4 kq + r
//   P^T fragment of 32 keys    = the packed scores of key tiles 2 s, 2 s + 1 of the lane itself: k slot 8 kq + j = key
//                                (j < 4: 32 s + 4 kq + j, else 32 s + 16 + 4 kq + j - 4) — V^T fragments follow that order:
//   V^T tile [16 d][32 keys]   : two ds_read_b64_tr_b16 (rows 32 s + 4 kq + (lc >> 2) and + 16, columns 16 dt + 4 (lane & 3))
//   O^T tile [16 d][16 q]      : lane (q, kq) register r = d 16 dt + 4 kq + r  -> 8-byte stores
// LDS image as above; the V tile's chunk swizzle also takes bit 2 of the row (^ 2) so that the kq = 0 / 1 halves of a
// transposing read fall on different banks.
// Wait states in front of every asm MFMA: without them the second wave of each SIMD (and only it) computed wrong values
// on some launches — an operand written or loaded just before the MFMA (cdna_hip_programming.md, inline asm rule 2:
// "a just-written operand -> MFMA operand: s_nop 1"); one state was enough in every run, two are kept.
#ifndef W32_PRE_N
#define W32_PRE_N 1
#endif
#define W32_STR2(x) #x
#define W32_STR(x) W32_STR2(x)
#if W32_PRE_N >= 0
#define W32_PRE "s_nop " W32_STR(W32_PRE_N) "\n\t"
#else
#define W32_PRE
#endif
__device__ __forceinline__ void w32_mfma_v(f32x4& c, const ssdk::u32x4& a, const ssdk::u32x4& b) {
  asm(W32_PRE "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void w32_mfma_v0(f32x4& c, const ssdk::u32x4& a, const ssdk::u32x4& b) {
  asm(W32_PRE "v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
typedef unsigned w32_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void w32_mfma_vt(f32x4& c, const w32_u2& a, const w32_u2& b) {      // k-step of 16
  asm(W32_PRE "v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void w32_mfma_a(f32x4& c, const ssdk::u32x4& a, const ssdk::u32x4& b) {
  asm(W32_PRE "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
#ifndef TV_FA_W32_KGAP
#define TV_FA_W32_KGAP 2
#endif
#ifndef TV_FA_W32_VGAP
#define TV_FA_W32_VGAP 2
#endif

__global__ __launch_bounds__(512) void flash_fwd_w32_kernel(AttnArgs a) {
  typedef bf16_t T;
  typedef bf16x4 v4;
  typedef Frag<bf16_t> F;
  typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
  constexpr int NW = 8, KB = 64, ROWB = 256, TILEB = KB * ROWB, NS = 4, QB = 256, QW = 32, DT = 5;
  constexpr int NQK = 12, NM = NQK + 2 * (DT + 1);          // MFMAs per segment: 4 key tiles x (2 + 1) k-steps, 2 x (5 + 1)
  extern __shared__ __attribute__((aligned(16))) unsigned char fa_smem[];
  const unsigned sK_off = (unsigned)(uintptr_t)(lds_u8*)fa_smem, sV_off = sK_off + NS * TILEB;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lc = lane & 15, kq = lane >> 4, q4 = lc >> 2, p4 = lane & 3;
  const int D = a.D;
  const int xcd = blockIdx.x & 7, step = gridDim.x >> 3;
  const int nslots = a.ppx * a.nqb, npairs = a.nb * a.Hq;
  int slot = blockIdx.x >> 3;
  int pair = xcd * a.ppx + slot / a.nqb, qblk = slot % a.nqb;
  if (slot >= nslots || pair >= npairs) return;
  const int gq = a.Hq / a.Hkv;
  const float c_ = a.scale_log2;
  auto k_of = [&](int pr) { return (const T*)a.k + (int64_t)(pr / a.Hq) * a.ksb + (int64_t)((pr % a.Hq) / gq) * a.ksh; };
  auto v_of = [&](int pr) { return (const T*)a.v + (int64_t)(pr / a.Hq) * a.vsb + (int64_t)((pr % a.Hq) / gq) * a.vsh; };
  const int dchunks = D / 8;

  // ---- Q of half X through a wave-private 4 KiB staging tile (16 rows, the K tile's image): 4 copy pieces
  const unsigned sQ_off = sV_off + NS * TILEB + (unsigned)(wave * 4096);
  auto copy_q = [&](int pr, int qb, int X) __attribute__((always_inline)) {
    const T* qp = (const T*)a.q + (int64_t)(pr / a.Hq) * a.qsb + (int64_t)(pr % a.Hq) * a.qsh;
    const void* sp = ssdk::uniform_ptr(qp);
    const int row0 = qb * QB + wave * QW + 16 * X;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = 4 * i + (lane >> 4);
      int cq = (lane & 15) ^ (rl & 15);
      cq = cq < dchunks ? cq : 0;
      ssdk::glds16(sp, (unsigned)(min(row0 + rl, a.Lq - 1) * (int)a.qsl * 2 + cq * 16), sQ_off + 1024u * i);
    }
  };
  // fragment addresses inside a 16-row tile: k-steps of 32 (chunk 4 ks + kq, 16 bytes) and the tail of 16 (8 bytes)
  const int k_rd0 = lc * ROWB + (((0 + kq) ^ lc) << 4), k_rd1 = lc * ROWB + (((4 + kq) ^ lc) << 4);
  const int k_rdt = lc * ROWB + (((8 + (kq >> 1)) ^ lc) << 4) + (kq & 1) * 8;
  u32x4 Qa[2][2];
  w32_u2 Qt[2];
  auto read_q = [&](int X) __attribute__((always_inline)) {
    Qa[X][0] = *(const __attribute__((address_space(3))) u32x4*)lds_at(sQ_off + (unsigned)k_rd0);
    Qa[X][1] = *(const __attribute__((address_space(3))) u32x4*)lds_at(sQ_off + (unsigned)k_rd1);
    const w32_u2 t = *(const __attribute__((address_space(3))) w32_u2*)lds_at(sQ_off + (unsigned)k_rdt);
    const w32_u2 z = {0u, 0u};
    Qt[X] = (64 + 4 * kq < D) ? t : z;             // columns past head_dim meet finite pad values of K: zero here
  };

  // ---- copies: this wave's two pieces of a tile = key rows 8 wave .. + 7
  const int ntiles = (a.Lk + KB - 1) / KB;             // >= 4 (launcher)
  const int left_last = a.Lk - (ntiles - 1) * KB;
  auto chunk_k = [&](int j) { const int row = 8 * wave + 4 * j + (lane >> 4); const int c = (lane & 15) ^ (row & 15); return c < dchunks ? c : 0; };
  auto chunk_v = [&](int j) { const int c = (lane & 15) ^ (4 * (lane >> 4)) ^ (2 * (j & 1)); return c < dchunks ? c : 0; };   // row & 3 = lane >> 4, bit 2 of the row = j
  unsigned offK[2], offV[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 8 * wave + 4 * j + (lane >> 4);
    offK[j] = (unsigned)(row * (int)a.ksl * 2 + chunk_k(j) * 16 - 1024 * j);
    offV[j] = (unsigned)(row * (int)a.vsl * 2 + chunk_v(j) * 16 - 1024 * j);
  }
  const unsigned m0_wave = (unsigned)(wave * 2048);
  auto copy2 = [&](const void* sp, unsigned o0, unsigned o1, unsigned dst) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(o0), "v"(o1), "s"(sp), "s"(dst) : "memory");
  };
  auto tile_copy = [&](const T* base, int kt, int stage, bool isK) __attribute__((always_inline)) {
    const int64_t sl = isK ? a.ksl : a.vsl;
    const void* sp = ssdk::uniform_ptr(base + (int64_t)kt * KB * sl);
    const unsigned dst = (isK ? sK_off : sV_off) + stage * TILEB + m0_wave;
    if (kt == ntiles - 1 && left_last < KB) {          // clamped rows (masked in the softmax): piece by piece
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = min(8 * wave + 4 * j + (lane >> 4), left_last - 1);
        ssdk::glds16(sp, (unsigned)(row * (int)sl * 2 + (isK ? chunk_k(j) : chunk_v(j)) * 16), dst + 1024u * j);
      }
    } else {
      copy2(sp, isK ? offK[0] : offV[0], isK ? offK[1] : offV[1], dst);
    }
  };
  int v_rd[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
    v_rd[dt] = (4 * kq + q4) * ROWB + (((2 * dt + (p4 >> 1)) ^ (4 * q4) ^ (2 * (kq & 1))) << 4) + (p4 & 1) * 8;

  // ---- state of the two halves
  f32x4 O[2][DT], Ls[2], S[2][4];
  u32x4 Pf[2][2];
  float mref[2] = {-INFINITY, -INFINITY}, tmax[2] = {-INFINITY, -INFINITY};
  bool first[2] = {true, true};
  u32x4 kf[3];
  u32x4 ones_u;
  asm volatile("v_mov_b32 %0, 0x3f803f80\n\tv_mov_b32 %1, 0x3f803f80\n\tv_mov_b32 %2, 0x3f803f80\n\tv_mov_b32 %3, 0x3f803f80"
               : "=v"(ones_u[0]), "=v"(ones_u[1]), "=v"(ones_u[2]), "=v"(ones_u[3]));
#pragma unroll
  for (int X = 0; X < 2; ++X) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) O[X][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    Ls[X] = f32x4{0.f, 0.f, 0.f, 0.f};
    Pf[X][0] = Pf[X][1] = u32x4{0u, 0u, 0u, 0u};
  }

  // ---- prologue
  {
    u32x4s* z = (u32x4s*)fa_smem;
    for (int i = tid; i < 2 * NS * TILEB / 16; i += 512) z[i] = u32x4s{0u, 0u, 0u, 0u};
  }
  __syncthreads();
  const T* kp = k_of(pair);
  const T* vp = v_of(pair);
  tile_copy(kp, 0, 0, true);
  tile_copy(kp, 1, 1, true);
  tile_copy(kp, 2, 2, true);
  tile_copy(vp, 0, 0, false);
  tile_copy(vp, 1, 1, false);
  copy_q(pair, qblk, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  read_q(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  copy_q(pair, qblk, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  read_q(1);
  __builtin_amdgcn_s_barrier();

  // K fragment of QK^T MFMA m (k-step m >> 2: 0, 1 = 32 wide, 2 = the tail of 16; key tile m & 3)
  auto kfrag = [&](unsigned cK, int m) __attribute__((always_inline)) {
    const int ks = m >> 2, kt = m & 3;
    if (ks < 2) return *(const __attribute__((address_space(3))) u32x4*)(lds_at(cK + (unsigned)(ks ? k_rd1 : k_rd0)) + kt * (16 * ROWB));
    const w32_u2 t = *(const __attribute__((address_space(3))) w32_u2*)(lds_at(cK + (unsigned)k_rdt) + kt * (16 * ROWB));
    return u32x4{t[0], t[1], 0u, 0u};
  };
  auto tile_max = [&](int X) __attribute__((always_inline)) {
    float m = fmaxf(S[X][0][0], S[X][0][1]);
#pragma unroll
    for (int u = 1; u < 8; ++u) m = w64_max3(m, S[X][u >> 1][2 * (u & 1)], S[X][u >> 1][2 * (u & 1) + 1]);
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    return fmaxf(m, __shfl_xor(m, 32, 64));
  };
  auto mask_tail = [&](int X, int kvalid) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) S[X][t][i] = (16 * t + 4 * kq + i < kvalid) ? S[X][t][i] : -INFINITY;
  };
  auto decide = [&](int X) __attribute__((always_inline)) {
    const float tm = tmax[X] * c_;
    if (__builtin_amdgcn_ballot_w64(tm > mref[X] + TV_FA_W64_LAZY)) {
      if (first[X]) {
        mref[X] = tm;
        first[X] = false;
      } else {
        const float mn = fmaxf(mref[X], tm);
        const float al = __builtin_amdgcn_exp2f(mref[X] - mn);
        mref[X] = mn;
        mfma_settle();
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) O[X][dt] *= al;
        Ls[X] *= al;
      }
    }
  };

  auto segment = [&](const int X, const bool masked, const int kvalid, unsigned cK, unsigned cV, unsigned cKn,
                     const T* cbase, int ckt, int cstage) __attribute__((always_inline)) {
    const int Y = 1 - X;
    const float nm = -mref[Y];
    unsigned vb[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vb[dt] = cV + (unsigned)v_rd[dt];
    v4 vlo[2][DT], vhi[2][DT];
    float ew[16], pw[16], mx = 0.f;
#pragma clang loop unroll(full)
    for (int m = 0; m < NM; ++m) {
      if (m < NQK) {
        const int ks = m >> 2, kt = m & 3;
        if (ks == 0) w32_mfma_v0(S[X][kt], kf[m % 3], Qa[X][0]);
        else if (ks == 1) w32_mfma_v(S[X][kt], kf[m % 3], Qa[X][1]);
        else w32_mfma_vt(S[X][kt], w32_u2{kf[m % 3][0], kf[m % 3][1]}, Qt[X]);
      } else {
        const int j = m - NQK, s_ = j / (DT + 1), w = j % (DT + 1);
        if (w < DT) {
          const bf16x8 vf8 = ssdk::cat4(vlo[s_][w], vhi[s_][w]);
          w32_mfma_a(O[X][w], __builtin_bit_cast(u32x4, vf8), Pf[X][s_]);
        } else {
          w32_mfma_a(Ls[X], ones_u, Pf[X][s_]);
        }
      }
      // ---- fillers
      if (m + 2 < NQK) kf[(m + 2) % 3] = kfrag(cK, m + 2);
      if (m >= NM - 2) kf[m - (NM - 2)] = kfrag(cKn, m - (NM - 2));        // the next segment's first two
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          if (m == NQK - 6 + (DT + 1) * s_ + dt) {
            vlo[s_][dt] = F::tr_read(lds_at(vb[dt]) + s_ * (32 * ROWB));
            vhi[s_][dt] = F::tr_read(lds_at(vb[dt]) + s_ * (32 * ROWB) + 16 * ROWB);
          }
      // softmax of Y: element e = 4 (key tile) + register; fma in gap f(e) = 22 e / 16, exp2 one gap later, the pair's
      // pack one more (elements of gap g: (16 g + 21) / 22 <= e < (16 (g + 1) + 21) / 22)
      {
        if (m >= 2) {
#pragma unroll
          for (int e = (16 * (m - 2) + 21) / 22; e < (16 * (m - 1) + 21) / 22; ++e)
            if (e < 16 && (e & 1)) {
              unsigned pk = w64_pk(pw[e - 1], pw[e]);
              asm volatile("" : "+v"(pk));
              Pf[Y][e >> 3][(e >> 1) & 3] = pk;
            }
        }
        if (m >= 1) {
#pragma unroll
          for (int e = (16 * (m - 1) + 21) / 22; e < (16 * m + 21) / 22; ++e)
            if (e < 16) {
              pw[e] = __builtin_amdgcn_exp2f(ew[e]);
              asm volatile("" : "+v"(pw[e]));
            }
        }
#pragma unroll
        for (int e = (16 * m + 21) / 22; e < (16 * (m + 1) + 21) / 22; ++e)
          if (e < 16) {
            ew[e] = w64_fma(S[Y][e >> 2][e & 3], c_, nm);
            asm volatile("" : "+v"(ew[e]));
          }
      }
      // tile maximum of X's new scores: the 8 register pairs in the gaps NQK + 2 .. NQK + 9
      if (m >= NQK + 2 && m < NQK + 10) {
        const int u = m - (NQK + 2);
        if (u == 0) asm volatile("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(S[X][0][0]), "v"(S[X][0][1]));
        else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(S[X][u >> 1][2 * (u & 1)]), "v"(S[X][u >> 1][2 * (u & 1) + 1]));
      }
      if (m == (X == 1 ? TV_FA_W32_KGAP : TV_FA_W32_VGAP)) tile_copy(cbase, ckt, cstage, X == 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (masked) {
      asm volatile("; masked tail");
      mask_tail(X, kvalid);
      tmax[X] = tile_max(X);
    } else {
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      tmax[X] = fmaxf(mx, __shfl_xor(mx, 32, 64));
    }
  };

  auto epilogue = [&](int X, int pr, int qb) __attribute__((always_inline)) {
    const int h = pr % a.Hq, b = pr / a.Hq;
    const int qrow = qb * QB + wave * QW + 16 * X + lc;
    mfma_settle();
    const float l_tot = Ls[X][0];
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    T* op = (T*)a.o + (int64_t)b * a.osb + (int64_t)min(qrow, a.Lq - 1) * a.osl + (int64_t)h * a.osh;
    const bool rowok = qrow < a.Lq;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d0 = 16 * dt + 4 * kq;
      const u32x2s u = {w64_pk(O[X][dt][0] * inv, O[X][dt][1] * inv), w64_pk(O[X][dt][2] * inv, O[X][dt][3] * inv)};
      if (rowok && d0 < D) *(u32x2s*)(op + d0) = u;
    }
    if (rowok && a.lse && kq == 0)
      a.lse[((int64_t)b * a.Hq + h) * a.Lq + qrow] = l_tot > 0.f ? (mref[X] * 0.6931471805599453f + logf(l_tot)) : -INFINITY;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) O[X][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    Ls[X] = f32x4{0.f, 0.f, 0.f, 0.f};
    mref[X] = -INFINITY;
    first[X] = true;
  };

  // ---- segment 0 of the stream: S_A of tile 0
  {
    asm volatile("s_nop 4" ::: "memory");     // (the Q fragments' zero-select is a VALU write: keep it away from the asm MFMAs)
#pragma unroll
    for (int m = 0; m < NQK; ++m) {
      const int ks = m >> 2, kt = m & 3;
      const u32x4 kx = kfrag(sK_off, m);
      if (ks == 0) w32_mfma_v0(S[0][kt], kx, Qa[0][0]);
      else if (ks == 1) w32_mfma_v(S[0][kt], kx, Qa[0][1]);
      else w32_mfma_vt(S[0][kt], w32_u2{kx[0], kx[1]}, Qt[0]);
    }
    mfma_settle();
    tmax[0] = tile_max(0);
    kf[0] = kfrag(sK_off, 0);
    kf[1] = kfrag(sK_off, 1);
  }

  const bool tail = left_last < KB;
#ifdef TV_FA_STAMP
  const bool st_on = blockIdx.x == 0 && (wave == 0 || wave == 4);
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = clock64();
#endif
  int n = 0;
  int pair_p = pair, qblk_p = qblk;
  bool have_prev = false, valid = true;
  for (;;) {
    const int slot_n = slot + step;
    const int pair_n = xcd * a.ppx + slot_n / a.nqb, qblk_n = slot_n % a.nqb;
    const bool has_next = valid && slot_n < nslots && pair_n < npairs;
    const T* kp_n = has_next ? k_of(pair_n) : kp;
    const T* vp_n = has_next ? v_of(pair_n) : vp;
    bool done = false;
    for (int kt = 0; kt < ntiles; ++kt, ++n) {
      const bool last = kt == ntiles - 1;
      const unsigned cK0 = sK_off + (unsigned)((n & 3) * TILEB), cK1 = sK_off + (unsigned)(((n + 1) & 3) * TILEB);
      const unsigned cV0 = sV_off + (unsigned)((n & 3) * TILEB), cVm = sV_off + (unsigned)(((n + 3) & 3) * TILEB);
      const bool kw = kt + 3 >= ntiles, vw = kt + 2 >= ntiles;
      const T* ck = kw ? kp_n : kp;
      const T* cv = vw ? vp_n : vp;
      const int ckt = kw ? kt + 3 - ntiles : kt + 3, vkt = vw ? kt + 2 - ntiles : kt + 2;
      // ---- segment 1
      W64STAMP(0);
      decide(0);
      if (last && has_next) copy_q(pair_n, qblk_n, 0);
      W64STAMP(1);
      segment(1, last && tail, left_last, cK0, cVm, cK1, ck, ckt, (n + 3) & 3);
      W64STAMP(2);
      if (kt == 0 && have_prev) epilogue(1, pair_p, qblk_p);
      if (!valid) { done = true; break; }
      if (last && has_next) {        // the 4 Q pieces are older than this segment's 2 K pieces
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        read_q(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      // ---- segment 2
      W64STAMP(3);
      decide(1);
      if (last && has_next) copy_q(pair_n, qblk_n, 1);
      W64STAMP(4);
      segment(0, kt == ntiles - 2 && tail, left_last, cK1, cV0, cK1, cv, vkt, (n + 2) & 3);
      W64STAMP(5);
      if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      W64STAMP(6);
      __builtin_amdgcn_s_barrier();
      W64STAMP(7);
      if (last) {
        epilogue(0, pair, qblk);
        if (has_next) {
          read_q(1);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    }
    if (done) break;
    pair_p = pair; qblk_p = qblk; have_prev = true;
    if (has_next) { slot = slot_n; pair = pair_n; qblk = qblk_n; kp = kp_n; vp = vp_n; }
    else valid = false;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TV_FA_STAMP
  if (st_on && lane == 0) {
    for (int i = 0; i < 8; ++i) g_fa_stamps[(wave >> 2) * 16 + i] = st_acc[i];
    g_fa_stamps[(wave >> 2) * 16 + 8] = (unsigned long long)n;
  }
#endif
}
