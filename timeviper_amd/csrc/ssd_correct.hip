// S3 companion: carried-in state correction of a scan that started from a zero state.
//
//   y_t += exp(cs_t) * C_t . S_in        cs_t = sum_{j <= t} dt_j A_h  (inclusive, from the range start)
//
// — SURVEY.md Appendix A's "sequence sharding" identity: a shard (another GPU's, or the second
// segment of one GPU's sequence) scans from a zero state, and once the state that enters it is
// known the outputs are completed by this term.  It is the Y_off term of modeling_nano.py:833-836
// with the chunk-local decay replaced by the decay from the range start.
//
// The factor exp(cs_t) only shrinks along the sequence (dt >= 0, A < 0); once cs_t * log2(e) < -32 the
// term is below 2.5e-10 of |C_t . S_in| — far under the fp32 rounding of the sum it would enter, let alone
// a bf16 ulp of y — and the kernel stops (C_UNDERFLOW; the reference's fp32 state passing loses such terms
// in its own additions).  Heads that forget within a few hundred tokens cost a few chunks; a head that
// never forgets costs the full range.
//
// Three launches: per-chunk log-decays (from dt), their exclusive prefix per head, and the
// correction itself: work-groups (slot k of 64, head, batch) walk the chunks k, k+64, ... of their
// head until the prefix underflows; per 64-token chunk C . S_in^T on MFMA 16x16x32 (S_in as bf16
// B fragments in registers for the whole walk, like the march's state snapshot), scaled rows
// through LDS, 16-byte read-modify-write of y.  The all-segments pass of the segmented march (tv_ssd_correct_all_launch)
// does not launch that grid: its prefix kernel lists one walker per CPW chunks of each (head, boundary)'s horizon and
// persistent work-groups take them off the list (ssd_correct_list_kernel) — no work-group is launched to find nothing
// to do, and a head that forgets slowly gets up to 32 walkers instead of 6.
#include <stdlib.h>
#include "ssd_common.hpp"

namespace {
using namespace ssdk;

constexpr int CQ = 64;            // tokens per chunk
constexpr int CN = 128;           // d_state
constexpr int MAXW = 32;          // walkers per (head, range) of the all-segments pass, at most
constexpr int CPW = 4;            // chunks of its horizon per walker (the 20 KB of S_in a walker loads: 14 % of what it moves)
constexpr int CSLOTS = 64;        // work-groups per (batch, head): a head that never forgets is walked by all of them
                                  // (measured with 8: the slowest heads set the launch time, 808 us in the 9B model)
// The walk of a head ends where the factor 2^(cs_t log2 e) has fallen below 2^-32: the term is then < 2.5e-10 of
// |C_t . S_in| — two orders of magnitude under the fp32 rounding of the sum it would be added to and seven
// under a bf16 ulp of y; round 2 walked on to 2^-160 (exactly 0 in fp32), five times the distance.
constexpr float C_UNDERFLOW = -32.f;

struct CorrArgs {
  bf16_t* y;
  const bf16_t *dt, *Cm;
  const float *A, *dt_bias, *state;
  float *tot, *pre;
  const bf16_t* state16;           // S_in as bf16 (the all-segments pass: written by ssd_seg_chain_kernel), or NULL: `state` (fp32)
  int L, H, P, G, nchunks;         // nchunks: chunks of the whole sequence (row length of `pre`)
  // the ranges that are corrected: nsegc of them per batch entry (blockIdx.z = b * nsegc + si), range si covering the
  // chunks [(si + first_seg) * seg_chunks, + seg_chunks) of the sequence, its entering state at state[si][b][h];
  // the stand-alone operator has one range = the whole sequence (nsegc 1, first_seg 0, seg_chunks = nchunks)
  int nsegc, first_seg, seg_chunks;
  int64_t tot_stride;              // elements between the (b, h) rows of `tot`
  int64_t ysb, ysl, dsb, dsl, dsh, csb, csl, csg;      // dsh: elements between the heads of dt (1: token-major rows)
  int softplus, group_map;
  float dt_min, dt_max;
  // the all-segments pass: a list of walkers (one per CPW chunks of a (head, range)'s horizon) built by the prefix kernel and
  // taken from a counter by persistent work-groups; NULL: one work-group per (slot, head, range) of the grid
  unsigned* items;
  unsigned* counters;              // [0] items written, [1] items taken
};

__device__ __forceinline__ float disc_dt(const CorrArgs& a, float raw, int h) {
  float d = raw + (a.dt_bias ? a.dt_bias[h] : 0.f);
  if (a.softplus) d = softplus_fast(d);
  return fminf(fmaxf(d, a.dt_min), a.dt_max);
}

// grid (nchunks, ceil(H / 4), B), 4 waves: wave = head, lane = token of the chunk
// tot[b][h][c] = log2(e) * sum_{t in chunk} dt_t A_h
__global__ __launch_bounds__(256) void ssd_chunk_decay_kernel(CorrArgs a) {
  const int c = blockIdx.x, b = blockIdx.z;
  const int h = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (h >= a.H) return;
  const int t = c * CQ + lane;
  const float d = t < a.L ? disc_dt(a, (float)a.dt[(int64_t)b * a.dsb + (int64_t)t * a.dsl + (int64_t)h * a.dsh], h) : 0.f;
  const float s = wave_sum(d);
  if (lane == 0) a.tot[((int64_t)b * a.H + h) * a.nchunks + c] = s * a.A[h] * 1.4426950408889634f;
}

// one wave per (head h, range z = b * nsegc + si): pre[b][h][c] = sum_{first chunk of c's range <= c' < c} tot[b][h][c'], and
// (all-segments pass) this (head, range)'s walkers onto the list
__device__ __forceinline__ void decay_prefix_wave(const CorrArgs& a, const int h, const int z, const int lane) {
  const int b = z / a.nsegc, si = z % a.nsegc;
  const int cbeg = (si + a.first_seg) * a.seg_chunks;
  const int nch = min(a.seg_chunks, a.nchunks - cbeg);
  const float* t = a.tot + ((int64_t)b * a.H + h) * a.tot_stride + cbeg;
  float* p = a.pre + ((int64_t)b * a.H + h) * a.nchunks + cbeg;
  float carry = 0.f;
  int hor = 0;                      // chunks whose factor has not underflowed (the prefix only decreases)
  for (int c0 = 0; c0 < nch; c0 += 64) {
    const float v = c0 + lane < nch ? t[c0 + lane] : 0.f;
    const float inc = wave_incl_scan_dpp(v);
    const float pv = carry + inc - v;
    if (c0 + lane < nch) p[c0 + lane] = pv;
    hor += __popcll(__ballot(c0 + lane < nch && !(pv < C_UNDERFLOW)));
    carry += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc), 63));
  }
  if (a.items && nch > 0) {         // (range 12 bits | head 10 | walker 5 | walkers - 1 5; the launcher keeps z < 4 096, h < 1 024)
    const int nwalk = min(MAXW, max(1, (hor + CPW - 1) / CPW));
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&a.counters[0], (unsigned)nwalk);
    base = __builtin_amdgcn_readfirstlane(base);
    if (lane < nwalk) a.items[base + lane] = ((unsigned)z << 20) | ((unsigned)h << 10) | ((unsigned)lane << 5) | (unsigned)(nwalk - 1);
  }
}
// grid (H, B * nsegc), one wave (the stand-alone operator)
__global__ __launch_bounds__(64) void ssd_decay_prefix_kernel(CorrArgs a) {
  decay_prefix_wave(a, blockIdx.x, blockIdx.y, threadIdx.x);
}

// One walker: chunks w, w + nwalk, ... of range (b, si) of head h, until the factor underflows.
template <int PT>
__device__ __forceinline__ void correct_walk(const CorrArgs& a, const int h, const int b, const int si, const int nbatch,
                                             const int w, const int nslots, float* ef, float* tile) {
  constexpr int LDW = PT * 16 + 4;                  // padded fp32 row of the staging tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, kq = lane >> 4;
  const int hpg = a.H / a.G;
  const int g = a.group_map ? (h % a.G) : (h / hpg);
  const int cbeg = (si + a.first_seg) * a.seg_chunks;          // first chunk / token of the range
  const int tbeg = cbeg * CQ;
  const int nch = min(a.seg_chunks, a.nchunks - cbeg);
  if (nch <= 0 || w >= nch) return;
  const int L = min(a.L - tbeg, nch * CQ);
  const float* pre = a.pre + ((int64_t)b * a.H + h) * a.nchunks + cbeg;
  if (pre[w] < C_UNDERFLOW) return;

  // S_in as B operand: lane (col lc, kq) of tile ct, k-step ks holds S[p = 16 ct + lc][n = 32 ks + 8 kq + 0..7]
  bf16x8 sf[PT][4];
  const int64_t sbase = (((int64_t)si * nbatch + b) * a.H + h) * a.P * CN;
#pragma unroll
  for (int ct = 0; ct < PT; ++ct) {
    const int p = 16 * ct + lc;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 v = {};
      if (p < a.P) {
        if (a.state16) {
          v = *(const bf16x8*)(a.state16 + sbase + (int64_t)p * CN + 32 * ks + 8 * kq);
        } else {
          const float* sp = a.state + sbase;
          const f32x4 lo = *(const f32x4*)(sp + (int64_t)p * CN + 32 * ks + 8 * kq);
          const f32x4 hi = *(const f32x4*)(sp + (int64_t)p * CN + 32 * ks + 8 * kq + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] = (bf16_t)lo[j]; v[4 + j] = (bf16_t)hi[j]; }
        }
      }
      sf[ct][ks] = v;
    }
  }
  const float Ah = a.A[h] * 1.4426950408889634f;
  const bf16_t* dp = a.dt + (int64_t)b * a.dsb + (int64_t)tbeg * a.dsl + (int64_t)h * a.dsh;
  const bf16_t* cp = a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg + (int64_t)tbeg * a.csl;
  bf16_t* yp = a.y + (int64_t)b * a.ysb + (int64_t)tbeg * a.ysl + (int64_t)h * a.P;
  const int nvec = a.P / 8;                          // 16-byte pieces per y row

  // One chunk per iteration; the C rows, the dt value and the prefix of the NEXT chunk of this work-group and the y
  // rows of the current one are requested before the current chunk's arithmetic, so an iteration waits for one
  // round trip to memory instead of three in a row (C -> MFMA -> y read -> y write; round 2: 80 % of the
  // wave-cycles of this kernel were parked).
  constexpr int NY = (CQ * 2 * PT + 255) / 256;      // y pieces per thread (rows of at most 16 PT columns)
  auto fetch = [&](int c, bf16x8 (&cf)[4], bf16x8 (&yv)[NY], float& draw, float& p0) {
    const int t0 = c * CQ;
    const int trow = min(t0 + 16 * wave + lc, L - 1);         // rows past the end repeat the last row: finite
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) cf[ks] = *(const bf16x8*)(cp + (int64_t)trow * a.csl + 32 * ks + 8 * kq);
    const int t = t0 + lane;
    draw = (wave == 0 && t < L) ? (float)dp[(int64_t)t * a.dsl] : 0.f;
    p0 = pre[c];
#pragma unroll
    for (int k = 0; k < NY; ++k) {       // the y rows this chunk's term is added to (requested a chunk ahead, like C)
      const int i = tid + 256 * k, row = i / nvec, ch = i % nvec;
      if (i < CQ * nvec && t0 + row < L) yv[k] = *(const bf16x8*)(yp + (int64_t)(t0 + row) * a.ysl + 8 * ch);
    }
  };
  bf16x8 cf[4], cfn[4], yv[NY], yvn[NY];
  float draw, drawn = 0.f, p0, p0n = 0.f;
  fetch(w, cf, yv, draw, p0);
  for (int c = w; c < nch; c += nslots) {
    if (p0 < C_UNDERFLOW) break;                     // the prefix only decreases: nothing left for this head
    const int t0 = c * CQ;
    const bool more = c + nslots < nch;
    if (more) fetch(c + nslots, cfn, yvn, drawn, p0n);
    if (wave == 0) {
      const float d = t0 + lane < L ? disc_dt(a, draw, h) : 0.f;
      const float cs = wave_incl_scan_dpp(d * Ah);
      ef[lane] = __builtin_amdgcn_exp2f(p0 + cs);
    }
    f32x4 acc[PT];
#pragma unroll
    for (int ct = 0; ct < PT; ++ct) {
      acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc[ct] = mfma16(cf[ks], sf[ct][ks], acc[ct]);
    }
    __syncthreads();                                 // ef ready; previous iteration's tile reads done
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * wave + 4 * kq + r;
      const float e = ef[row];
#pragma unroll
      for (int ct = 0; ct < PT; ++ct) tile[row * LDW + 16 * ct + lc] = acc[ct][r] * e;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NY; ++k) {
      const int i = tid + 256 * k, row = i / nvec, ch = i % nvec;
      if (i < CQ * nvec && t0 + row < L) {
        bf16x8 v = yv[k];
        const f32x4 lo = *(const f32x4*)(tile + row * LDW + 8 * ch);
        const f32x4 hi = *(const f32x4*)(tile + row * LDW + 8 * ch + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = (bf16_t)((float)v[j] + lo[j]);
          v[4 + j] = (bf16_t)((float)v[4 + j] + hi[j]);
        }
        *(bf16x8*)(yp + (int64_t)(t0 + row) * a.ysl + 8 * ch) = v;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) cf[ks] = cfn[ks];
#pragma unroll
    for (int k = 0; k < NY; ++k) yv[k] = yvn[k];
    draw = drawn;
    p0 = p0n;
  }
}

// grid (slots, H, B * nsegc), 256 threads.  PT = ceil(P / 16) column tiles.  Work-group `slot` of a (head, range)
// walks the chunks slot, slot + gridDim.x, ... of its range.
template <int PT>
__global__ __launch_bounds__(256) void ssd_correct_kernel(CorrArgs a) {
  __shared__ float ef[CQ];
  __shared__ __attribute__((aligned(16))) float tile[CQ * (PT * 16 + 4)];
  correct_walk<PT>(a, blockIdx.y, blockIdx.z / a.nsegc, blockIdx.z % a.nsegc, gridDim.z / a.nsegc, blockIdx.x, gridDim.x, ef, tile);
}

// The all-segments pass: persistent work-groups take walkers off the list the prefix kernel wrote.  A grid of
// (slots, heads, boundaries) work-groups paid ~10 ns per work-group that found nothing to do — 5 376 of them at 6 slots,
// 55 us a call where every head forgets within a chunk — and left a head that forgets slowly to 6 walkers of 50 chunks
// each while the rest of the chip idled (202 us a launch inside the 9B forward, whose tokens are correlated).
template <int PT>
__global__ __launch_bounds__(256) void ssd_correct_list_kernel(CorrArgs a, int nbatch) {
  __shared__ float ef[CQ];
  __shared__ __attribute__((aligned(16))) float tile[CQ * (PT * 16 + 4)];
  __shared__ unsigned next;
  const unsigned nitems = a.counters[0];
  for (;;) {
    __syncthreads();                                 // the previous walker's LDS reads are done
    if (threadIdx.x == 0) next = atomicAdd(&a.counters[1], 1u);
    __syncthreads();
    const unsigned i = next;
    if (i >= nitems) return;
    const unsigned it = a.items[i];
    const int z = (int)(it >> 20), h = (int)((it >> 10) & 1023u), w = (int)((it >> 5) & 31u), nwalk = (int)(it & 31u) + 1;
    correct_walk<PT>(a, h, z / a.nsegc, z % a.nsegc, nbatch, w, nwalk, ef, tile);
  }
}

// Chain of the per-segment results of a segmented march (segments > 0 marched from a zero state):
//   run = seg_state[0];  for s >= 1:  S_in(s) = run (stored as bf16: the correction's MFMA operand);
//   run = exp(decay[s]) run + seg_state[s];  final state = run;  total decay = sum of the segments'
__device__ __forceinline__ void seg_chain_block(const float* __restrict__ seg_state, const float* __restrict__ seg_decay,
                                                bf16_t* __restrict__ sin16, float* __restrict__ final_state,
                                                float* __restrict__ total_decay, int nseg, int64_t bh, int64_t per_head,
                                                const int64_t i) {             // i: float4 index
  const int64_t n4 = bh * per_head / 4;
  if (i < n4) {
    const int64_t head = (i * 4) / per_head;
    f32x4 run = ((const f32x4*)seg_state)[i];
    for (int s = 1; s < nseg; ++s) {
      bf16x4 o = {(bf16_t)run[0], (bf16_t)run[1], (bf16_t)run[2], (bf16_t)run[3]};
      ((bf16x4*)sin16)[(int64_t)(s - 1) * n4 + i] = o;
      const float e = __expf(seg_decay[(int64_t)s * bh + head]);
      const f32x4 cur = ((const f32x4*)seg_state)[(int64_t)s * n4 + i];
      run = f32x4{e * run[0] + cur[0], e * run[1] + cur[1], e * run[2] + cur[2], e * run[3] + cur[3]};
    }
    if (final_state) ((f32x4*)final_state)[i] = run;
  }
  if (total_decay && i < bh) {
    float t = 0.f;
    for (int s = 0; s < nseg; ++s) t += seg_decay[(int64_t)s * bh + i];
    total_decay[i] = t;
  }
}
// One launch for the two small passes between the march and the correction: blocks [0, nb_chain) chain the segment states, the
// rest take four (head, range) pairs each and write their decay prefixes and walkers (a launch less per call: ~12 us of ~1 360
// inside the 9B forward, of ~420 at 33 k tokens).  The list's counters are zeroed by the caller's FIRST launch (the dt transpose).
__global__ __launch_bounds__(256) void ssd_chain_prefix_kernel(const float* __restrict__ seg_state, const float* __restrict__ seg_decay,
                                                               bf16_t* __restrict__ sin16, float* __restrict__ final_state,
                                                               float* __restrict__ total_decay, int nseg, int64_t bh,
                                                               int64_t per_head, int nb_chain, CorrArgs a, int npairs) {
  if ((int)blockIdx.x < nb_chain) {
    seg_chain_block(seg_state, seg_decay, sin16, final_state, total_decay, nseg, bh, per_head, (int64_t)blockIdx.x * 256 + threadIdx.x);
  } else {
    const int pair = ((int)blockIdx.x - nb_chain) * 4 + (int)(threadIdx.x >> 6);
    if (pair < npairs) decay_prefix_wave(a, pair % a.H, pair / a.H, threadIdx.x & 63);
  }
}

template <int PT> void launch_correct(const CorrArgs& a, dim3 grid, hipStream_t st) {
  ssd_correct_kernel<PT><<<grid, 256, 0, st>>>(a);
}
template <int PT> void launch_correct_list(const CorrArgs& a, int nwg, int nbatch, hipStream_t st) {
  ssd_correct_list_kernel<PT><<<dim3(nwg), 256, 0, st>>>(a, nbatch);
}
void launch_correct_list_pt(const CorrArgs& a, int nwg, int nbatch, int headdim, hipStream_t st) {
  switch ((headdim + 15) / 16) {
    case 1: launch_correct_list<1>(a, nwg, nbatch, st); break;
    case 2: launch_correct_list<2>(a, nwg, nbatch, st); break;
    case 3: launch_correct_list<3>(a, nwg, nbatch, st); break;
    case 4: launch_correct_list<4>(a, nwg, nbatch, st); break;
    case 5: launch_correct_list<5>(a, nwg, nbatch, st); break;
    case 6: launch_correct_list<6>(a, nwg, nbatch, st); break;
    case 7: launch_correct_list<7>(a, nwg, nbatch, st); break;
    default: launch_correct_list<8>(a, nwg, nbatch, st); break;
  }
}
void launch_correct_pt(const CorrArgs& a, dim3 grid, int headdim, hipStream_t st) {
  switch ((headdim + 15) / 16) {
    case 1: launch_correct<1>(a, grid, st); break;
    case 2: launch_correct<2>(a, grid, st); break;
    case 3: launch_correct<3>(a, grid, st); break;
    case 4: launch_correct<4>(a, grid, st); break;
    case 5: launch_correct<5>(a, grid, st); break;
    case 6: launch_correct<6>(a, grid, st); break;
    case 7: launch_correct<7>(a, grid, st); break;
    default: launch_correct<8>(a, grid, st); break;
  }
}

}  // namespace

size_t tv_ssd_correct_workspace_bytes(int batch, int seqlen, int nheads) {
  const size_t nchunks = (size_t)(seqlen + CQ - 1) / CQ;
  return 2 * (size_t)batch * nheads * nchunks * sizeof(float);
}

bool tv_ssd_correct_supported(int headdim, int dstate, int dtype, int nheads) {
  return dtype == TV_BF16 && dstate == CN && headdim % 8 == 0 && headdim <= 128 && nheads <= 1024;
}

// chunk_tot: per-chunk log2-decays (B, H, nchunks_total) already produced by the scan kernel for the
// chunks [chunk0, chunk0 + nchunks) of this range (row stride chunk_tot_stride), or NULL: computed here.
int tv_ssd_correct_launch(void* y, const void* dt, const void* A, const void* Cm, const void* dt_bias,
                          const void* state_in, int batch, int seqlen, int nheads, int headdim,
                          int ngroups, int64_t ysb, int64_t ysl, int64_t dsb, int64_t dsl, int64_t csb,
                          int64_t csl, int64_t csg, int dt_softplus, float dt_min, float dt_max,
                          int group_map, void* workspace, const float* chunk_tot, int64_t chunk_tot_stride,
                          hipStream_t st) {
  CorrArgs a;
  a.y = (bf16_t*)y; a.dt = (const bf16_t*)dt; a.Cm = (const bf16_t*)Cm;
  a.A = (const float*)A; a.dt_bias = (const float*)dt_bias; a.state = (const float*)state_in;
  a.L = seqlen; a.H = nheads; a.P = headdim; a.G = ngroups;
  a.nchunks = (seqlen + CQ - 1) / CQ;
  a.tot = (float*)workspace;
  a.pre = a.tot + (size_t)batch * nheads * a.nchunks;
  a.ysb = ysb; a.ysl = ysl; a.dsb = dsb; a.dsl = dsl; a.dsh = 1; a.csb = csb; a.csl = csl; a.csg = csg;
  a.softplus = dt_softplus; a.group_map = group_map; a.dt_min = dt_min; a.dt_max = dt_max;
  a.tot_stride = a.nchunks;
  a.state16 = nullptr;
  a.items = nullptr; a.counters = nullptr;
  a.nsegc = 1; a.first_seg = 0; a.seg_chunks = a.nchunks;
  if (chunk_tot) {
    a.tot = const_cast<float*>(chunk_tot);
    a.tot_stride = chunk_tot_stride;
  } else {
    ssd_chunk_decay_kernel<<<dim3(a.nchunks, (nheads + 3) / 4, batch), 256, 0, st>>>(a);
  }
  ssd_decay_prefix_kernel<<<dim3(nheads, batch), 64, 0, st>>>(a);
  launch_correct_pt(a, dim3(CSLOTS, nheads, batch), headdim, st);
  TV_LAUNCH_CHECK();
}

// All segments of a segmented march in three launches (chain, prefixes, correction): the head-per-wave march
// (ssd_head.hip) cuts a sequence into up to 16 segments; one launch per boundary cost a fixed ~140 us each
// (every one of 64 x heads work-groups loading the 40 KB fp32 state before looking at its first chunk).
size_t tv_ssd_correct_all_workspace_bytes(int batch, int nheads, int nchunks, int nseg, int headdim) {
  // pre (B, H, nchunks) fp32 + S_in of the nseg - 1 boundaries as bf16 + the walker list (MAXW per head and boundary) + its counters
  const size_t nb = (size_t)(nseg > 1 ? nseg - 1 : 0);
  return ((size_t)batch * nheads * nchunks * sizeof(float) + 255) / 256 * 256 +
         (nb * batch * nheads * headdim * CN * sizeof(bf16_t) + 255) / 256 * 256 +
         nb * batch * nheads * MAXW * sizeof(unsigned) + 256;
}

// the walker list's two counters inside `workspace` (zeroed by the caller before tv_ssd_correct_all_launch)
unsigned* tv_ssd_correct_all_counters(void* workspace, int batch, int nheads, int nchunks, int nseg, int headdim) {
  const size_t nb = (size_t)(nseg > 1 ? nseg - 1 : 0);
  unsigned char* sin16 = (unsigned char*)workspace + ((size_t)batch * nheads * nchunks * sizeof(float) + 255) / 256 * 256;
  unsigned char* after = sin16 + (nb * batch * nheads * headdim * CN * sizeof(bf16_t) + 255) / 256 * 256;
  return (unsigned*)(after + nb * batch * nheads * MAXW * sizeof(unsigned));
}

int tv_ssd_correct_all_launch(void* y, const void* dt, const void* A, const void* Cm, const void* dt_bias,
                              const float* seg_state, const float* seg_decay, float* final_state,
                              float* total_decay, const float* chunk_tot, int batch, int seqlen, int nheads,
                              int headdim, int ngroups, int nseg, int seg_chunks, int64_t ysb, int64_t ysl,
                              int64_t dsb, int64_t dsl, int64_t dsh, int64_t csb, int64_t csl, int64_t csg, int dt_softplus,
                              float dt_min, float dt_max, int group_map, void* workspace, hipStream_t st) {
  CorrArgs a;
  a.y = (bf16_t*)y; a.dt = (const bf16_t*)dt; a.Cm = (const bf16_t*)Cm;
  a.A = (const float*)A; a.dt_bias = (const float*)dt_bias; a.state = nullptr;
  a.L = seqlen; a.H = nheads; a.P = headdim; a.G = ngroups;
  a.nchunks = (seqlen + CQ - 1) / CQ;
  a.pre = (float*)workspace;
  bf16_t* sin16 = (bf16_t*)((unsigned char*)workspace + ((size_t)batch * nheads * a.nchunks * sizeof(float) + 255) / 256 * 256);
  a.state16 = sin16;
  {
    const size_t nb = (size_t)(nseg > 1 ? nseg - 1 : 0);
    unsigned char* after = (unsigned char*)sin16 + (nb * batch * nheads * headdim * CN * sizeof(bf16_t) + 255) / 256 * 256;
    a.items = (unsigned*)after;
    a.counters = (unsigned*)(after + nb * batch * nheads * MAXW * sizeof(unsigned));
  }
  a.tot = const_cast<float*>(chunk_tot);
  a.tot_stride = a.nchunks;
  a.ysb = ysb; a.ysl = ysl; a.dsb = dsb; a.dsl = dsl; a.dsh = dsh; a.csb = csb; a.csl = csl; a.csg = csg;
  a.softplus = dt_softplus; a.group_map = group_map; a.dt_min = dt_min; a.dt_max = dt_max;
  int nsegc = 0;                     // segments that exist (a short sequence may leave the last ones empty)
  for (int s = 1; s < nseg; ++s) if ((int64_t)s * seg_chunks < a.nchunks) ++nsegc;
  a.nsegc = nsegc > 0 ? nsegc : 1; a.first_seg = 1; a.seg_chunks = seg_chunks;
  const int64_t bh = (int64_t)batch * nheads, per_head = (int64_t)headdim * CN;
  const int64_t n4 = bh * per_head / 4;
  const int nb_chain = (int)((n4 + 255) / 256);
  const int npairs = nsegc > 0 ? nheads * batch * nsegc : 0;
  ssd_chain_prefix_kernel<<<dim3((unsigned)(nb_chain + (npairs + 3) / 4)), 256, 0, st>>>(seg_state, seg_decay, sin16, final_state, total_decay,
                                                                                      nseg, bh, per_head, nb_chain, a, npairs);
  if (nsegc > 0) {
    // walkers: one per CPW chunks of a (head, boundary)'s horizon (at most MAXW), listed by the prefix kernel, taken off the list by
    // persistent work-groups (4 per CU).  TV_CORR_SLOTS = n (dev tool): the round-3 grid of n work-groups per (head, boundary)
    // instead (whole scan at 163 940 / 32 868 tokens, bench-like dt, on that grid: 1: 2 560 / 754 us, 4: 2 454 / 694, 6: 2 458 / 681,
    // 8: 2 499 / 699, 16: 2 613 / 785, 32: 2 827 / 961).
    int slots = 0;
    if (const char* e = getenv("TV_CORR_SLOTS")) slots = atoi(e);
    // a list item packs the range (batch x boundary) into 12 bits and the head into 10: beyond that the grid form
    if (slots <= 0 && ((int64_t)batch * nsegc >= 4096 || nheads >= 1024)) slots = 6;
    if (slots > 0) {
      CorrArgs g = a;
      g.items = nullptr;
      launch_correct_pt(g, dim3(slots < seg_chunks ? slots : seg_chunks, nheads, batch * nsegc), headdim, st);
    } else {
      const int64_t maxitems = (int64_t)batch * nsegc * nheads * MAXW;
      launch_correct_list_pt(a, (int)(maxitems < 1024 ? maxitems : 1024), batch, headdim, st);
    }
  }
  TV_LAUNCH_CHECK();
}

extern "C" size_t tv_ssd_state_correction_workspace_bytes(int batch, int seqlen, int nheads) {
  if (batch <= 0 || seqlen <= 0 || nheads <= 0) return 0;
  return tv_ssd_correct_workspace_bytes(batch, seqlen, nheads);
}

extern "C" int tv_ssd_state_correction(void* y, const void* dt, const void* A, const void* Cm,
                                       const void* dt_bias, const void* state_in, int batch, int seqlen,
                                       int nheads, int headdim, int ngroups, int dstate,
                                       int64_t y_stride_b, int64_t y_stride_l, int64_t dt_stride_b,
                                       int64_t dt_stride_l, int64_t c_stride_b, int64_t c_stride_l,
                                       int64_t c_stride_g, int dtype, int dt_softplus, float dt_min,
                                       float dt_max, int group_map, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  TV_CHECK_ARG(batch > 0 && seqlen >= 0 && nheads > 0 && headdim > 0 && ngroups > 0 && dstate > 0 &&
                   nheads % ngroups == 0, "ssd_state_correction: bad sizes");
  if (seqlen == 0) return TV_OK;
  TV_CHECK_ARG(y && dt && A && Cm && state_in, "ssd_state_correction: null pointer");
  if (!tv_ssd_correct_supported(headdim, dstate, dtype, nheads))
    TV_UNSUPPORTED("ssd_state_correction: bf16, d_state 128, head_dim %% 8 == 0 and <= 128 only (got dtype %d, N %d, P %d)",
                   dtype, dstate, headdim);
  if (y_stride_l % 8 || y_stride_b % 8 || c_stride_l % 8 || c_stride_g % 8 || c_stride_b % 8 ||
      ((uintptr_t)y & 15) || ((uintptr_t)Cm & 15) || ((uintptr_t)state_in & 15))
    TV_UNSUPPORTED("ssd_state_correction: y / C rows must be 16-byte aligned");
  const size_t need = tv_ssd_correct_workspace_bytes(batch, seqlen, nheads);
  if (!workspace || workspace_bytes < need) {
    tv_set_error("ssd_state_correction: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return TV_ERR_WORKSPACE;
  }
  return tv_ssd_correct_launch(y, dt, A, Cm, dt_bias, state_in, batch, seqlen, nheads, headdim, ngroups,
                               y_stride_b, y_stride_l, dt_stride_b, dt_stride_l, c_stride_b, c_stride_l,
                               c_stride_g, dt_softplus, dt_min, dt_max, group_map, workspace, nullptr, 0,
                               (hipStream_t)stream);
}
