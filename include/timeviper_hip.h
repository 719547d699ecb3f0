/*
 * timeviper_hip.h — C ABI of libtimeviper_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary of the TimeViper long-video forward hot path.
 * The reference (xiaomi-research/timeviper) is pure Python; its accelerated
 * operators are calls into un-vendored wheels (mamba_ssm 2.2.5, causal_conv1d
 * 1.5.2, flash_attn 2.8.0.post2, timm).  Each entry point below replaces one of
 * those call sites.  A maintainer binds them with ctypes;
 * `timeviper_amd/kernels.py` is exactly that binding, exposing the reference's
 * Python operator names.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch storage) unless the
 *     name ends in `_host`;
 *   - innermost dimension is contiguous, outer strides are given in ELEMENTS;
 *   - `dtype` is a tv_dtype; fp32 statistics/accumulation inside every kernel;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never
 *     synchronised; no allocation happens inside any entry point (workspaces
 *     are caller-provided, sizes from the *_workspace_bytes query);
 *   - return value: 0 = TV_OK, negative = tv_status error (nothing launched).
 */
#ifndef TIMEVIPER_HIP_H
#define TIMEVIPER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { TV_F32 = 0, TV_BF16 = 1, TV_F16 = 2 } tv_dtype;

typedef enum {
  TV_OK = 0,
  TV_ERR_BAD_ARG = -1,      /* null pointer / non-positive size            */
  TV_ERR_UNSUPPORTED = -2,  /* shape or dtype outside the compiled kernels */
  TV_ERR_WORKSPACE = -3,    /* workspace too small                         */
  TV_ERR_LAUNCH = -4        /* hipGetLastError() != hipSuccess             */
} tv_status;

/* ABI version; bumped whenever a signature changes. */
int tv_abi_version(void);
/* Human readable string of the last error on this thread (never NULL). */
const char* tv_last_error(void);
/* Hash of the sources (csrc/, include/, compiler flags) this binary was built from; the Python
 * binding compares it with the hash of the tree it runs from, so that a kernel edit is never
 * tested or benchmarked through a stale library (timeviper_amd/build.py source_id()). */
const char* tv_build_id(void);

/* ------------------------------------------------------------------------
 * S2  causal depthwise conv1d (+bias, +SiLU), channels-last.
 * Replaces causal_conv1d_fn(x=(B,C,L) view of (B,L,C), weight=(C,K), bias,
 * activation="silu")        reference: modeling_nano.py:619-624 (CPU twin :705)
 *   y[b,t,c] = act(bias[c] + sum_{j<K} w[c,j] * xpad[b, t-(K-1)+j, c])
 * xpad rows t<0 come from `halo` ((B,K-1,C) contiguous, the K-1 rows that
 * precede this shard) or are zero when halo==NULL.
 * x,y: (B,L,C) with row strides in elements; weight (C,K) and bias (C) in
 * `dtype` too (what nn.Conv1d holds after .to(dtype)); bias may be NULL.
 * K in [2,4].  C % (16/sizeof(elem)) == 0.
 * --------------------------------------------------------------------- */
int tv_causal_conv1d_fwd(const void* x, const void* weight, const void* bias,
                         const void* halo, void* y, int batch, int seqlen,
                         int channels, int kernel, int64_t x_stride_b,
                         int64_t x_stride_l, int64_t y_stride_b,
                         int64_t y_stride_l, int dtype, int silu, void* stream);

/* Same convolution for the Mamba-2 mixer's xBC projection, with the three
 * channel segments [x: d_inner | B: G*N | C: G*N] (the split of
 * modeling_nano.py:628-636) written to three destinations: x as (B,L,d_inner)
 * rows, B and C GROUP-MAJOR as (B,G,L,N) — the layout the scan kernel streams
 * best.  Input x (B,L,d_inner+2GN) as in tv_causal_conv1d_fwd.               */
int tv_causal_conv1d_xbc_fwd(const void* x, const void* weight, const void* bias,
                             const void* halo, void* y_x, void* y_b, void* y_c,
                             int batch, int seqlen, int d_inner, int ngroups,
                             int dstate, int kernel, int64_t x_stride_b,
                             int64_t x_stride_l, int dtype, int silu,
                             void* stream);

/* The same with one more output: the causal part of C.B^T of every (64-token
 * chunk, group) — the G_{l,s} = C_l . B_s of the reference's intra-chunk term
 * (modeling_nano.py:800-811) — in the MFMA-fragment order tv_ssd_scan_cb_fwd
 * reads (6 fragments of 512 bf16 per chunk and group; tv_ssd_cb_bytes() bytes).
 * The B / C tiles are multiplied while they are still on the chip, so the scan
 * does not read B and C back for it.  bf16, d_state 128 only.                */
size_t tv_ssd_cb_bytes(int batch, int seqlen, int ngroups);
int tv_causal_conv1d_xbc_cb_fwd(const void* x, const void* weight, const void* bias,
                                const void* halo, void* y_x, void* y_b, void* y_c,
                                void* cb, int batch, int seqlen, int d_inner,
                                int ngroups, int dstate, int kernel,
                                int64_t x_stride_b, int64_t x_stride_l, int dtype,
                                int silu, void* stream);

/* Single-token decode step, replaces causal_conv1d_update (:495-501).
 * conv_state (B,C,K) contiguous, updated in place (shift left, append x).   */
int tv_causal_conv1d_update(const void* x, void* conv_state, const void* weight,
                            const void* bias, void* y, int batch, int channels,
                            int kernel, int dtype, int silu, void* stream);

/* ------------------------------------------------------------------------
 * L1  RMSNorm, optionally fused with the preceding residual add.
 * Replaces NemotronHRMSNorm.forward (modeling_nano.py:897-903) and the
 * `residual + hidden_states` of NemotronHBlock.forward (:966).
 *   s      = x (+ delta)            rounded to `dtype`  -> sum_out (if !NULL)
 *   y[r,:] = w * s * rsqrt(mean(s^2) + eps)             (fp32 math)
 * delta / sum_out may be NULL.  weight is `wdtype` (TV_F32 or == dtype).
 * --------------------------------------------------------------------- */
int tv_rmsnorm_fwd(const void* x, const void* delta, const void* weight,
                   void* sum_out, void* y, int64_t rows, int dim,
                   int64_t x_stride, int64_t delta_stride, int64_t sum_stride,
                   int64_t y_stride, float eps, int dtype, int wdtype,
                   void* stream);

/* ViT blocks (V1/V2): LayerNorm with the residual add of the previous sub-layer fused in
 * (timm Block: x = x + attn(norm1(x)); x = x + mlp(norm2(x))), and the exact (erf) GELU of
 * the MLP.  s = x (+ delta) rounded to dtype -> sum_out; y = (s-mean)*rsqrt(var+eps)*w + b.
 * row_bias (dim) fp32, optional: one row added to every s before the statistics (not to
 * sum_out) — the biases of the preceding projections when the caller lets the GEMM
 * accumulate straight into the residual stream (D = A W^T + C) and carries them beside it. */
int tv_layernorm_fwd(const void* x, const void* delta, const void* weight,
                     const void* bias, const void* row_bias, void* sum_out, void* y, int64_t rows, int dim,
                     int64_t x_stride, int64_t delta_stride, int64_t sum_stride,
                     int64_t y_stride, float eps, int dtype, void* stream);
int tv_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);
/* M1: y = relu(x)^2, the "relu2" activation of NemotronHMLP.forward (modeling_nano.py:993-994,
 * ACT2FN["relu2"]: square(relu(x)), one rounding to dtype); in place when y == x. */
int tv_relu2_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * S4  gated, grouped RMSNorm.  Replaces mamba_ssm rmsnorm_fn(x, weight,
 * bias=None, z=gate, eps, group_size, norm_before_gate=False)
 *                                        reference: modeling_nano.py:371-380
 *   u = x * silu(z);  y = w * u * rsqrt(mean_group(u^2) + eps)
 * z may be NULL (plain grouped RMSNorm).  dim % group_size == 0.
 * --------------------------------------------------------------------- */
int tv_rmsnorm_gated_fwd(const void* x, const void* z, const void* weight,
                         void* y, int64_t rows, int dim, int group_size,
                         int64_t x_stride, int64_t z_stride, int64_t y_stride,
                         float eps, int dtype, int wdtype, void* stream);

/* ------------------------------------------------------------------------
 * S3  Mamba-2 SSD selective scan (prefill).  Replaces
 * mamba_chunk_scan_combined(x, dt, A, B, C, chunk_size, D, z=None,
 *     seq_idx=None, return_final_states=True, dt_bias, dt_softplus=True,
 *     [dt_limit], [initial_states])
 *                 reference: modeling_nano.py:639-653 (CPU twin :775-851)
 *   dt_t  = clamp(softplus(dt_raw + dt_bias_h), dt_min, dt_max)
 *   S_t   = exp(dt_t A_h) S_{t-1} + dt_t x_t (outer) B_t ;  S_{-1} = init
 *   y_t   = S_t . C_t + D_h x_t ;   final = S_{L-1}
 * x (B,L,H,P), dt (B,L,H), Bm/Cm (B,L,G,N), y (B,L,H,P) in `dtype` with
 * per-tensor (batch,row) strides; Bm/Cm additionally carry a group stride, so
 * both the reference's token-major rows (stride_g = N) and the group-major
 * layout (G,L,N) written by tv_causal_conv1d_xbc_fwd (stride_l = N, stride_g =
 * L*N; consecutive tokens of one group are contiguous and spread over all L2
 * channels) are accepted; A, D, dt_bias (H) fp32; init_state /
 * final_state (B,H,P,N) fp32 contiguous (either may be NULL); D, dt_bias may
 * be NULL.  group_map: 0 = head h reads group h / (H/G) (GPU reference and
 * checkpoints), 1 = h % G (quirk of the reference's CPU torch_forward,
 * :781-782).  The chunk size is an implementation detail (results are
 * chunk-invariant up to rounding) and is not part of the ABI.
 * total_decay (B,H) fp32, optional: sum_t dt_t*A_h over the sequence, the
 * per-head log-decay a sequence-sharded caller needs to chain shard states.
 * workspace: tv_ssd_scan_workspace_bytes() bytes of device memory, 16-byte aligned (may be 0 /
 * NULL: the workspace-free kernel, the fp32 token recurrence, is used then — correct for every
 * shape and dtype, and two orders of magnitude slower than the marches); nothing persists in it
 * between calls.  seqlen == 0 is valid (the state passes through).
 * Approximations of the default bf16 kernel (head-per-wave march, csrc/ssd_head.hip), all far below
 * the bf16 rounding of y and bounded in tests/test_fullsize_gpu.py / test_ops_gpu.py:
 *  - a 64-token chunk that decays the state by more than 2^-64 starts the state anew: what the old
 *    state would have carried over (< 2^-64 of it) is dropped ("reset step");
 *  - inside such a chunk, token weights below 2^-126 (tokens whose true weight is < 2^-26 of the
 *    chunk's last token's) are flushed to zero;
 *  - x~ = w_s x_s is rounded to bf16 in a frame that depends on where the march started, so two
 *    marches over the same tokens from different starting points (sequence segments, shards) agree
 *    on y to a bf16 ulp and on the final state to 5e-3 relative, not to fp32; linearity in x holds
 *    up to the last bit of a sum;
 *  - sequence segments > 0 (the kernel cuts a long sequence into up to 16 segments marched
 *    concurrently from a zero state) are completed by tv_ssd_state_correction's operator, with its
 *    2^-32 cut-off (below).
 * --------------------------------------------------------------------- */
size_t tv_ssd_scan_workspace_bytes(int batch, int seqlen, int nheads,
                                   int headdim, int ngroups, int dstate,
                                   int dtype);
int tv_ssd_scan_fwd(const void* x, const void* dt, const void* A,
                    const void* Bm, const void* Cm, const void* D,
                    const void* dt_bias, const void* init_state, void* y,
                    void* final_state, void* total_decay, int batch,
                    int seqlen, int nheads, int headdim, int ngroups,
                    int dstate, int64_t x_stride_b, int64_t x_stride_l,
                    int64_t dt_stride_b, int64_t dt_stride_l,
                    int64_t b_stride_b, int64_t b_stride_l, int64_t b_stride_g,
                    int64_t c_stride_b, int64_t c_stride_l, int64_t c_stride_g,
                    int64_t y_stride_b, int64_t y_stride_l,
                    int dtype, int dt_softplus, float dt_min, float dt_max,
                    int group_map, void* workspace, size_t workspace_bytes,
                    void* stream);
/* tv_ssd_scan_fwd with the C.B^T fragments supplied by the caller (`cb`: the
 * output of tv_causal_conv1d_xbc_cb_fwd on the same Bm / Cm, or NULL: computed
 * here by a pre-pass over Bm and Cm as in tv_ssd_scan_fwd).  Kernels that do
 * not use the fragments (fp32, other d_state) ignore them.                   */
int tv_ssd_scan_cb_fwd(const void* x, const void* dt, const void* A,
                       const void* Bm, const void* Cm, const void* cb,
                       const void* D, const void* dt_bias,
                       const void* init_state, void* y, void* final_state,
                       void* total_decay, int batch, int seqlen, int nheads,
                       int headdim, int ngroups, int dstate, int64_t x_stride_b,
                       int64_t x_stride_l, int64_t dt_stride_b,
                       int64_t dt_stride_l, int64_t b_stride_b,
                       int64_t b_stride_l, int64_t b_stride_g, int64_t c_stride_b,
                       int64_t c_stride_l, int64_t c_stride_g, int64_t y_stride_b,
                       int64_t y_stride_l, int dtype, int dt_softplus, float dt_min,
                       float dt_max, int group_map, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Carried-in state correction ("sequence sharding"; the Y_off term of
 * modeling_nano.py:833-836 with the decay taken from the range start): a shard — another GPU's
 * tokens, or a later segment of one GPU's sequence — is scanned from a ZERO state by
 * tv_ssd_scan_fwd; once the state entering it is known its outputs are completed in place,
 *   y_t += exp(sum_{j<=t} dt_j A_h) * C_t . state_in[h]        (dt discretised as in tv_ssd_scan_fwd)
 * y (B,L,H,P) `dtype`, read-modify-write; dt (B,L,H) raw; Cm (B,L,G,N) with a group stride;
 * state_in (B,H,P,N) fp32 contiguous.  Past a head's decay horizon (factor < 2^-32: the term is
 * under 2.4e-10 of |C_t . state_in| — two orders of magnitude below the fp32 rounding of the sum it
 * would enter, seven below a bf16 ulp of y; csrc/ssd_correct.hip C_UNDERFLOW) the kernel stops at the
 * next chunk boundary, so the cost is that horizon, not L.  The dropped part is bounded by
 * 2^-32 |C_t . state_in| per element (tests/test_ops_gpu.py::test_ssd_state_correction_truncation_bound).
 * state_in enters the MFMA as bf16 (one rounding of the carried state, 2^-9 relative, like the
 * bf16 state operand of the scan's own Y_off product).
 * bf16, d_state 128, headdim % 8 == 0 (<= 128).  workspace:
 * tv_ssd_state_correction_workspace_bytes() bytes (per-chunk log-decays and their prefix). */
size_t tv_ssd_state_correction_workspace_bytes(int batch, int seqlen, int nheads);
int tv_ssd_state_correction(void* y, const void* dt, const void* A, const void* Cm,
                            const void* dt_bias, const void* state_in, int batch,
                            int seqlen, int nheads, int headdim, int ngroups,
                            int dstate, int64_t y_stride_b, int64_t y_stride_l,
                            int64_t dt_stride_b, int64_t dt_stride_l,
                            int64_t c_stride_b, int64_t c_stride_l,
                            int64_t c_stride_g, int dtype, int dt_softplus,
                            float dt_min, float dt_max, int group_map,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Force a particular implementation (testing/benchmarking; process-global):
 * 0 = auto (with a workspace: 6 where it applies, else 4 / 3; where no MFMA march applies — other dtypes, d_state != 128 —
 *     8 where it applies, else 1.  Without a workspace: 1),
 * 1 = generic fp32 token recurrence (any dtype / shape; the definition the others are tested against),
 * 3 = MFMA slice march, slices of <= 40 columns of a head per work-group (bf16, d_state 128, head_dim a multiple of 8
 *     that splits so; C.B^T pre-pass into `workspace`, tv_ssd_scan_workspace_bytes() bytes, 16-byte aligned),
 * 4 = whole-head slice march (head_dim 56 .. 80; 8 waves) x sequence segments,
 * 6 = head-per-wave march (csrc/ssd_head.hip; head_dim 32 / 64 / 80),
 * 8 = the generic path in chunk-parallel form (csrc/ssd_chunked.hip: any dtype, d_state <= 64, head_dim <= 128, 65 ..
 *     4 096 tokens; fp32 arithmetic, needs the workspace).
 * 3 / 4 / 6 fall back (6 -> 4 -> 3) where the shape does not fit and fail with TV_ERR_UNSUPPORTED where none does.
 * 2, 5 and 7 named kernels that measured slower and were removed (docs/history.md); they select 3, 4 and 6. */
void tv_ssd_scan_set_impl(int impl);
/* kernel family (same numbers) the most recent scan call of this process ran on; 0 before the first call */
int tv_ssd_scan_last_impl(void);
/* Implementation 6 at head_dim 80 with 4 heads per work-group (the Nano-9B shape): 1 = the 64-token step as the
 * generated instruction stream (csrc/ssd_head_step.inc; default), 0 = the C++ step, -1 = back to the default /
 * the environment's TV_HEAD_ASM.  Same results bit for bit (tests/test_ops_gpu.py); process-global, for A/B runs. */
void tv_ssd_head_set_asm(int on);

/* Single-token decode step, replaces selective_state_update (:528-539) with
 * the head-broadcast arguments the reference passes (A,D,dt_bias per head). */
int tv_selective_state_update(void* state, const void* x, const void* dt,
                              const void* A, const void* Bm, const void* Cm,
                              const void* D, const void* dt_bias, void* y,
                              int batch, int nheads, int headdim, int ngroups,
                              int dstate, int dtype, int dt_softplus,
                              void* stream);

/* ------------------------------------------------------------------------
 * ViT linears with fused epilogues: C = epilogue(A . W^T), A (M,K), W (N,K) =
 * nn.Linear.weight as stored, C (M,N), all bf16 row-major with element row strides lda / ldw / ldc.
 *   epilogue 0  C = acc + bias
 *            1  C = gelu(bf16(acc + bias)), exact (erf) GELU — timm Mlp fc1 + act
 *               (base_vision.py:274-278 via timm; InternVideo2 Mlp vit_scale_clean.py:296-320): the
 *               rounding points of a bf16 GEMM followed by tv_gelu_fwd, bit for bit
 *            2  C = C + acc (bias ignored): output projections accumulating into the residual stream
 * bias (N) fp32 or bf16 (bias_dtype = TV_F32 / TV_BF16) or NULL.  K % 128 == 0, N % 4 == 0, rows
 * 16-byte aligned; any M (row tails are masked).  fp32 accumulation on v_mfma_f32_16x16x32_bf16.
 * --------------------------------------------------------------------- */
int tv_gemm_bf16_fwd(const void* A, const void* W, const void* bias, void* C, int64_t M, int N,
                     int K, int64_t lda, int64_t ldw, int64_t ldc, int epilogue, int bias_dtype,
                     void* stream);
/* Big shapes (both extents >= 256, N % 8 == 0, >= 4 tiles per compute unit) with epilogue 0 or 1 run on a PERSISTENT
 * kernel whose tile epilogue is hidden inside the main loop (csrc/gemm_persist.hip); the rest — epilogue 2 always — on one
 * work-group per 256 x 256 tile (csrc/gemm.hip).  Same results bit for bit for epilogues 0 / 1.  mode -1: automatic
 * (default), 0: never, 1: wherever the shape allows (tests); grid: work-groups of the persistent kernels, 0 = one per
 * compute unit. */
void tv_gemm_set_persist(int mode, int grid);
/* Of the persistent kernels the first choice is csrc/gemm_drip.hip: 256 x 192 tiles (the ViT's widths 1 152 / 3 456 /
 * 4 352 are multiples of 192, not of 256) whose finished tile leaves during the NEXT tile's K loop — half of it parked in
 * LDS, half in accumulator registers, stored one 1 KiB piece per phase with the GELU applied on the way out.  Takes
 * K >= 640, K % 128 == 0, N % 8 == 0, N >= 192, M >= 256, an fp32 bias for epilogues 0 / 1.  mode -1: automatic (default;
 * TV_GEMM_DRIP=0 in the environment switches it off), 0: never, 1: wherever the shape allows (tests). */
void tv_gemm_set_drip(int mode);

/* ------------------------------------------------------------------------
 * A1 / T3 / ViT  fused softmax attention forward (flash-style, no S x S
 * matrix).  Replaces _flash_attention_forward (modeling_nano.py:1198-1209),
 * F.scaled_dot_product_attention (:1300, cross_attention.py:310) and
 * flash_attn_varlen_qkvpacked_func (flash_attention_class.py:59-66).
 * q (B,Lq,Hq,D), k/v (B,Lk,Hkv,D), o (B,Lq,Hq,D); strides per tensor for
 * (batch,row,head) in elements, D contiguous.  GQA: q head h uses kv head
 * h / (Hq/Hkv).  causal!=0: bottom-right aligned (query i sees keys
 * j <= i + Lk - Lq).  lse (B,Hq,Lq) fp32 optional (natural-log sum-exp of
 * the scaled scores), used to merge partial results across sequence shards.
 * D in {64,72,80,88,96,128}; dtype bf16/f16.
 * --------------------------------------------------------------------- */
int tv_flash_attn_fwd(const void* q, const void* k, const void* v, void* o,
                      void* lse, int batch, int seqlen_q, int seqlen_k,
                      int nheads_q, int nheads_kv, int headdim,
                      int64_t q_stride_b, int64_t q_stride_l,
                      int64_t q_stride_h, int64_t k_stride_b,
                      int64_t k_stride_l, int64_t k_stride_h,
                      int64_t v_stride_b, int64_t v_stride_l,
                      int64_t v_stride_h, int64_t o_stride_b,
                      int64_t o_stride_l, int64_t o_stride_h,
                      float softmax_scale, int causal, int dtype, void* stream);

/* Kernel variant for many short non-causal sequences of head_dim 65..80 (the SigLIP ViT frames; testing /
 * benchmarking, process-global): 0 = auto — bf16, >= 193 keys: the kernel with the key-tile loop as generated instruction
 * streams (csrc/attention_vit.hpp: one wave per SIMD, 64 query rows a wave, the softmax of tile i beside P.V of tile i - 1
 * and Q.K^T of tile i + 1, a lazy running maximum; round 6), otherwise the compiled streaming kernel (8 waves x 32 query
 * rows, two waves per SIMD); 4 = the same choice as 0; 5 = always the compiled streaming kernel (P's row sums out of the
 * P.V MFMAs through a ones column in the V ring); 3 = the compiled kernel with the row sums on the vector pipe (`l` summed
 * in fp32 before P is rounded) — 3 and 5 for A/B runs and tests.  The generated kernel's results equal the compiled one's
 * within the bf16 rounding of P (not bit for bit: P is rounded relative to another maximum).  (1 / 2 named two
 * measured-slower kernels removed in round 6: docs/history.md; the values are accepted and mean 0.)  Initial value: env
 * TV_FA_W64. */
void tv_flash_attn_set_variant(int variant);

/* The same operator with the QK^T and PV products on the FP8 matrix path of CDNA4
 * (v_mfma_f32_32x32x64_f8f6f4, OCP e4m3 operands, fp32 accumulation, fp32 softmax): BASELINE
 * config 5 ("fp8 MFMA attention path"; call sites modeling_qwen2.py:196-244,
 * modeling_nano.py:1198-1209).  The reference's arithmetic is bf16, so this variant is opt-in and
 * judged by tolerance: q, k, v are quantised per (batch, head) to max |x| = 440, P per row to
 * 2^8 p; inputs, output and lse keep the layout, dtype and meaning of tv_flash_attn_fwd.
 * headdim <= 128 (multiple of 8).  workspace: tv_flash_attn_fp8_workspace_bytes() bytes,
 * 256-byte aligned (the quantised copies: ~1 byte per q / k / v element, head_dim padded to
 * 128); nothing persists in it between calls. */
size_t tv_flash_attn_fp8_workspace_bytes(int batch, int seqlen_q, int seqlen_k,
                                         int nheads_q, int nheads_kv);
int tv_flash_attn_fp8_fwd(const void* q, const void* k, const void* v, void* o,
                          void* lse, int batch, int seqlen_q, int seqlen_k,
                          int nheads_q, int nheads_kv, int headdim,
                          int64_t q_stride_b, int64_t q_stride_l,
                          int64_t q_stride_h, int64_t k_stride_b,
                          int64_t k_stride_l, int64_t k_stride_h,
                          int64_t v_stride_b, int64_t v_stride_l,
                          int64_t v_stride_h, int64_t o_stride_b,
                          int64_t o_stride_l, int64_t o_stride_h,
                          float softmax_scale, int causal, int dtype,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Decode-step linear layers (torch.nn.Linear with 1..4 rows, modeling_nano.py:484-546 / :993-1000 at q_len 1):
 *   y[m][n] = sum_k f(x)[m][k] W[n][k] (+ bias[n]),   W (N, K) row-major with row stride ldw, bf16, fp32 accumulation.
 * f is the single-row operator the reference runs in front of the product, computed in the kernel's prologue with the
 * rounding points of the stand-alone kernels:
 *   prologue 0  f(x) = x
 *            1  RMSNorm with the residual add: s = bf16(x + delta) (delta NULL: s = x), written to sum_out if not NULL;
 *               f = bf16(norm_weight * (s * rsqrt(mean(s^2) + eps)))          (tv_rmsnorm_fwd; :897-903, :966)
 *            2  f = bf16(relu(x)^2)                                           (tv_relu2_fwd; :993-994)
 *            3  gated group RMSNorm: v = x * silu(gate) (gate NULL: v = x),
 *               f = bf16(norm_weight * (v * rsqrt(mean_group(v^2) + eps))), groups of group_size channels
 *                                                                             (tv_rmsnorm_gated_fwd; :371-380)
 * norm_weight_dtype TV_F32 or TV_BF16.  M <= 4, M * K * 2 <= 128 KiB, K % 8 == 0; x / W / delta / gate / sum_out rows
 * 16-byte aligned.  The arguments of the prologues not selected are ignored.
 * Epilogue (conv_state not NULL; K < 8192): the outputs [conv_row0, conv_row0 + conv_channels) — the xBC slice of the
 * mixer's in_proj — pass through tv_causal_conv1d_update (width 4, SiLU; same arithmetic) before they are stored:
 * conv_state (M, conv_channels, 4) is shifted and takes the bf16-rounded product, y holds the activation.
 * Replaces the in_proj -> causal_conv1d_update pair of modeling_nano.py:484-501.
 * --------------------------------------------------------------------- */
int tv_gemv_bf16_fwd(const void* x, const void* W, const void* bias, void* y, int M, int N, int K,
                     int64_t x_stride, int64_t ldw, int64_t y_stride, int prologue, const void* delta,
                     int64_t delta_stride, void* sum_out, int64_t sum_stride, const void* norm_weight,
                     int norm_weight_dtype, float eps, const void* gate, int64_t gate_stride,
                     int group_size, void* conv_state, const void* conv_weight, const void* conv_bias,
                     int conv_row0, int conv_channels, void* stream);

/* The decode step of the same operator: ONE query token per sequence against a K / V cache (q_len == 1 in
 * modeling_nano.py:1198-1209; cache layout of HybridMambaAttentionDynamicCache, :205-360).  q (B, Hq, D) and
 * o (B, Hq, D) by batch / head strides, k / v (B, Lk, Hkv, D) as in tv_flash_attn_fwd; lse (B, Hq) fp32 or NULL.
 * The q-heads of a kv-head share one pass over its K / V, the keys are split over the chip (split-KV) and merged by a
 * second launch.  seqlens_k: NULL, or DEVICE memory holding `batch` ints — the keys of each sequence actually in
 * use, clamped to [0, seqlen_k]; seqlen_k is then the capacity the launch is sized for, so a step captured in a
 * hipGraph replays against a growing cache.  headdim 128, bf16 (other shapes: tv_flash_attn_fwd with seqlen_q 1,
 * same results up to the summation order).  workspace: tv_attn_decode_workspace_bytes(), 16-byte aligned. */
size_t tv_attn_decode_workspace_bytes(int batch, int nheads_q, int nheads_kv, int seqlen_k);
int tv_attn_decode_fwd(const void* q, const void* k, const void* v, void* o, void* lse, int batch,
                       int seqlen_k, const int* seqlens_k, int nheads_q, int nheads_kv, int headdim,
                       int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_b,
                       int64_t k_stride_l, int64_t k_stride_h, int64_t v_stride_b,
                       int64_t v_stride_l, int64_t v_stride_h, int64_t o_stride_b,
                       int64_t o_stride_h, float softmax_scale, int dtype,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * T2  pdrop "attn" ranking.  Replaces modeling_nano.py:1822-1857,:1914-1939
 * without the (L,L) mask: one query row (the last prompt token) against all
 * keys <= that row, fp32 softmax per head, mean over heads, vision span only.
 *   scores[j] = mean_h softmax_k(q_h . K_{g(h),k} * scale)[vis_start + j]
 * q (Hq,D), k (L,Hkv,D) row/head strides; keys 0..n_keys-1 take part in the
 * softmax (n_keys = query_row+1); scores (n_vis) fp32.
 * workspace: tv_attn_rank_workspace_bytes.
 * --------------------------------------------------------------------- */
size_t tv_attn_rank_workspace_bytes(int n_keys, int nheads_q);
int tv_attn_rank_scores(const void* q, const void* k, void* scores,
                        int n_keys, int nheads_q, int nheads_kv, int headdim,
                        int64_t k_stride_l, int64_t k_stride_h,
                        int vis_start, int n_vis, float scale, int dtype,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The two halves of tv_attn_rank_scores for a sequence-sharded caller: every rank
 * computes the logits of ITS keys, logits (n_keys, Hq) fp32 key-major — per-shard pieces
 * concatenate along dim 0 — with the reference's roundings (q.K^T and the 1/sqrt(d) scaling
 * rounded to `dtype`, modeling_nano.py:1923-1927); after an all-gather every rank runs the
 * softmax statistics + head mean on the full (n_keys, Hq) array.  tv_attn_rank_scores is exactly
 * these two calls, so one GPU and N GPUs rank — and keep — the same tokens.
 * workspace of the second call: 2 * nheads_q floats. */
int tv_attn_rank_logits(const void* q, const void* k, void* logits, int n_keys,
                        int nheads_q, int nheads_kv, int headdim,
                        int64_t k_stride_l, int64_t k_stride_h, float scale,
                        int dtype, void* stream);
int tv_attn_rank_scores_from_logits(const void* logits, void* scores, int n_keys,
                                    int nheads_q, int vis_start, int n_vis,
                                    int dtype, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * T1  row gather (token keep / drop).  Replaces features[i][top_rank_index,:]
 * and the concat of modeling_nano.py:1982-1989.
 *   dst[r,:] = src[index[r],:]      index int64, values in [0, src_rows)
 * --------------------------------------------------------------------- */
int tv_gather_rows(const void* src, const int64_t* index, void* dst,
                   int64_t n_rows, int dim, int64_t src_stride,
                   int64_t dst_stride, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Qwen2 backbone ("next" row, BASELINE config 5).
 * tv_rope_fwd: rotary embedding in place on x (rows, heads, headdim) with element
 *   strides (x_stride_l, x_stride_h); cos / sin (rows, headdim) contiguous in x's
 *   dtype (the tables Qwen2RotaryEmbedding returns, modeling_qwen2.py:362-385):
 *   x' = x*cos + rotate_half(x)*sin  (apply_rotary_pos_emb :89-113, called :211-214).
 * tv_silu_mul_fwd: y = silu(gate) * up  (Qwen2MLP.forward :78-80), row strides in
 *   elements (gate and up may be the two halves of one fused projection).
 * --------------------------------------------------------------------- */
int tv_rope_fwd(void* x, const void* cos, const void* sin, int64_t rows,
                int heads, int headdim, int64_t x_stride_l, int64_t x_stride_h,
                int dtype, void* stream);
int tv_silu_mul_fwd(const void* gate, const void* up, void* y, int64_t rows,
                    int dim, int64_t gate_stride, int64_t up_stride,
                    int64_t y_stride, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * V3  ToMe projector: ONE round of bipartite soft matching + size-weighted merge,
 * all frames at once.  Replaces bipartite_soft_matching + merge_wavg
 * (timeviper/model/projector/tome.py:14-83) as called per round by
 * ToMe16_mlp_hd64.merge_tokens (:118-152).
 *   x (frames, tokens, dim) contiguous; size_in (frames, tokens) in x's dtype or
 *   NULL (all ones); r even tokens per frame are merged into their best odd match.
 *   x_out (frames, tokens - r, dim) = [kept evens, descending score | odds],
 *   size_out (frames, tokens - r).  metric = mean over `heads` head slices.
 *   workspace: tv_tome_workspace_bytes() bytes, 16-byte aligned.
 * --------------------------------------------------------------------- */
size_t tv_tome_workspace_bytes(int frames, int tokens, int dim, int heads);
int tv_tome_merge_round(const void* x, const void* size_in, void* x_out,
                        void* size_out, int frames, int tokens, int dim,
                        int heads, int r, int dtype, void* workspace,
                        size_t workspace_bytes, void* stream);

/* uniform keep indices, reference :1946-1953: torch.linspace(0, n-1, keep,
 * dtype=long) with CPU (double-step, symmetric halves) semantics; out int64
 * (keep) on the device.  Bit-exact with the CPU reference for any n<2^31.  */
int tv_uniform_keep_indices(int64_t* out, int64_t n_tokens, int64_t keep,
                            int64_t offset, void* stream);

/* complement of a sorted keep list inside [start, start+n): the dropped
 * vision tokens (reference :1966-1970, torch.isin + ~mask).  out_dropped
 * has n - n_keep entries, ascending.                                        */
int tv_dropped_indices(const int64_t* keep_sorted, int64_t n_keep,
                       int64_t start, int64_t n, int64_t* out_dropped,
                       void* stream);

/* ------------------------------------------------------------------------
 * V1/V2  ViT patch embedding as an im2col-free MFMA GEMM.  Replaces
 * timm PatchEmbed Conv2d(k=s=patch) (base_vision.py:146-170) and
 * InternVideo2 PatchEmbed Conv3d(k=s=(1,p,p)) (vit_scale_clean.py:445-461).
 *   out[f, py*gw+px, :] = W . patch(f,py,px) + bias (+ pos[py*gw+px,:])
 * pixels (F,Cin,H,W) contiguous `dtype`; weight (Dout, Cin*p*p) `dtype`
 * (the Conv weight flattened); bias (Dout) / pos (gh*gw, Dout) optional;
 * out (F, gh*gw, Dout).
 * workspace: tv_patch_embed_workspace_bytes(dout, cin, patch) bytes, 16-byte
 * aligned, caller-provided: the launcher packs the weight into it ((c,dy) rows
 * padded to the 16-wide MFMA k-step) before the GEMM; NULL selects the slower
 * path that re-pads the raw weight inside every work-group.
 * --------------------------------------------------------------------- */
size_t tv_patch_embed_workspace_bytes(int dout, int cin, int patch);
int tv_patch_embed_fwd(const void* pixels, const void* weight, const void* bias,
                       const void* pos, void* out, int frames, int cin,
                       int height, int width, int patch, int dout, int dtype,
                       void* workspace, void* stream);
/* Same GEMM with an explicit frame layout: the pixel offset of (frame f,
 * channel c) is (f / frames_per_group) * group_stride + (f % frames_per_group)
 * * frame_stride + c * chan_stride (elements).  Conv3d input (B,C,T,H,W) with
 * k=s=(1,p,p): frames_per_group=T, group_stride=C*T*H*W, frame_stride=H*W,
 * chan_stride=T*H*W; output token order (b, t, py, px) as the reference's
 * flatten(3).permute(0,2,3,1) (vit_scale_clean.py:455-460).               */
int tv_patch_embed_strided_fwd(const void* pixels, const void* weight,
                               const void* bias, const void* pos, void* out,
                               int frames, int cin, int height, int width,
                               int patch, int dout, int frames_per_group,
                               int64_t group_stride, int64_t frame_stride,
                               int64_t chan_stride, int dtype, void* workspace,
                               void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TIMEVIPER_HIP_H */
