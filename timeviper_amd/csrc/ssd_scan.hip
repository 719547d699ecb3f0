// S3 dispatcher: tv_ssd_scan_fwd picks an MFMA march kernel (bf16, d_state 128, MFMA-tileable head_dim, a
// workspace supplied: the head-per-wave march, else a slice march) or the generic fp32 path (chunk-parallel
// with a workspace and a small d_state, else the token recurrence).
#include <atomic>
#include "common.hpp"

int tv_ssd_generic_launch(const void* x, const void* dt, const void* A, const void* Bm,
                          const void* Cm, const void* D, const void* dt_bias,
                          const void* init_state, void* y, void* final_state, void* total_decay,
                          int batch, int seqlen, int nheads, int headdim, int ngroups, int dstate,
                          int64_t xsb, int64_t xsl, int64_t dsb, int64_t dsl, int64_t bsb,
                          int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg, int64_t ysb, int64_t ysl,
                          int dtype, int dt_softplus, float dt_min, float dt_max, int group_map,
                          hipStream_t st);

// ssd_slice.hip
bool tv_ssd_slice_supported(int seqlen, int nheads, int headdim, int ngroups, int dstate,
                            int dtype, int64_t xsl, int64_t bsl, int64_t bsg, int64_t csl, int64_t csg, int64_t ysl,
                            const void* x, const void* Bm, const void* Cm, const void* y, int wide);
size_t tv_ssd_slice_workspace_bytes(int batch, int seqlen, int nheads, int headdim, int ngroups,
                                    int dstate, int wide);
int tv_ssd_slice_launch(const void* x, const void* dt, const void* A, const void* Bm,
                        const void* Cm, const void* D, const void* dt_bias,
                        const void* init_state, void* y, void* final_state, void* total_decay,
                        int batch, int seqlen, int nheads, int headdim, int ngroups, int dstate,
                        int64_t xsb, int64_t xsl, int64_t dsb, int64_t dsl, int64_t bsb,
                        int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg, int64_t ysb, int64_t ysl,
                        int dtype, int dt_softplus, float dt_min, float dt_max, int group_map,
                        void* workspace, size_t workspace_bytes, int wide, const void* cb_pre, hipStream_t st);

// ssd_chunked.hip: the generic path's chunk-parallel form (small d_state, any dtype)
bool tv_ssd_chunked_supported(int seqlen, int headdim, int dstate);
size_t tv_ssd_chunked_workspace_bytes(int batch, int seqlen, int nheads, int headdim, int dstate);
int tv_ssd_chunked_launch(const void* x, const void* dt, const void* A, const void* Bm, const void* Cm, const void* D,
                          const void* dt_bias, const void* init_state, void* y, void* final_state, void* total_decay,
                          int batch, int seqlen, int nheads, int headdim, int ngroups, int dstate, int64_t xsb, int64_t xsl,
                          int64_t dsb, int64_t dsl, int64_t bsb, int64_t bsl, int64_t bsg, int64_t csb, int64_t csl,
                          int64_t csg, int64_t ysb, int64_t ysl, int dtype, int dt_softplus, float dt_min, float dt_max,
                          int group_map, void* workspace, hipStream_t st);

// ssd_head.hip
bool tv_ssd_head_supported(int seqlen, int nheads, int headdim, int ngroups, int dstate, int dtype,
                           int64_t xsl, int64_t bsl, int64_t bsg, int64_t csl, int64_t csg, int64_t ysl,
                           const void* x, const void* Bm, const void* Cm, const void* y);
size_t tv_ssd_head_workspace_bytes(int batch, int seqlen, int nheads, int headdim, int ngroups);
int tv_ssd_head_launch(const void* x, const void* dt, const void* A, const void* Bm, const void* Cm,
                       const void* D, const void* dt_bias, const void* init_state, void* y,
                       void* final_state, void* total_decay, int batch, int seqlen, int nheads, int headdim,
                       int ngroups, int64_t xsb, int64_t xsl, int64_t dsb, int64_t dsl, int64_t bsb,
                       int64_t bsl, int64_t bsg, int64_t csb, int64_t csl, int64_t csg, int64_t ysb,
                       int64_t ysl, int dt_softplus, float dt_min, float dt_max, int group_map,
                       void* workspace, size_t workspace_bytes, const void* cb_pre, hipStream_t st);

// 0 auto, 1 generic token recurrence (ssd_generic.hip: the definition), 3 slice march, two or more work-groups per head
// (ssd_slice.hip: the fallback for head dims that split into <= 40-column slices), 4 slice march, whole-head work-groups
// (head_dim 56 .. 80) x concurrent sequence segments + carried-in state correction (ssd_slice.hip + ssd_correct.hip),
// 6 head-per-wave march (ssd_head.hip: a wave owns a head, a work-group the heads of one B/C group, up to 16 sequence
// segments + correction; head_dim 32 / 64 / 80), 8 the generic path in chunk-parallel form (ssd_chunked.hip: small
// d_state, 2 .. 64 chunks) — what automatic selection takes where no march applies.
// Removed, docs/history.md: 2 (round 1's chunk march; the number now means 3), 5 (round 3's 12-wave whole-head layout; now
// means 4), 7 (round 4's two-waves-per-SIMD head march; now means 6).
// process-wide override for tests / dev tools; atomic so that concurrent callers never race on it
static std::atomic<int> g_ssd_impl{0};
static const int kAutoImpl = 6;   // head-per-wave march; falls back to 4, 3, the generic path

extern "C" void tv_ssd_scan_set_impl(int impl) { g_ssd_impl.store(impl, std::memory_order_relaxed); }
// which kernel family the most recent tv_ssd_scan_fwd / _cb_fwd of this process ran on (the numbers above; tests assert
// that a shape reached the kernel they mean to check)
static std::atomic<int> g_ssd_last{0};
extern "C" int tv_ssd_scan_last_impl(void) { return g_ssd_last.load(std::memory_order_relaxed); }

extern "C" size_t tv_ssd_scan_workspace_bytes(int batch, int seqlen, int nheads, int headdim,
                                              int ngroups, int dstate, int dtype) {
  if (dtype != TV_BF16 || dstate != 128)
    return (tv_ssd_chunked_workspace_bytes(batch, seqlen, nheads, headdim, dstate) + 255) / 256 * 256;
  const size_t narrow = tv_ssd_slice_workspace_bytes(batch, seqlen, nheads, headdim, ngroups, dstate, 0);
  const size_t wide = tv_ssd_slice_workspace_bytes(batch, seqlen, nheads, headdim, ngroups, dstate, 1);
  const size_t head = tv_ssd_head_workspace_bytes(batch, seqlen, nheads, headdim, ngroups);
  size_t m = narrow > wide ? narrow : wide;             // any variant may be selected (tv_ssd_scan_set_impl)
  return ((m > head ? m : head) + 255) / 256 * 256;
}

static int scan_impl(const void* x, const void* dt, const void* A, const void* Bm,
                     const void* Cm, const void* D, const void* dt_bias,
                     const void* init_state, void* y, void* final_state,
                     void* total_decay, int batch, int seqlen, int nheads, int headdim,
                     int ngroups, int dstate, int64_t x_stride_b, int64_t x_stride_l,
                     int64_t dt_stride_b, int64_t dt_stride_l, int64_t b_stride_b,
                     int64_t b_stride_l, int64_t b_stride_g, int64_t c_stride_b,
                     int64_t c_stride_l, int64_t c_stride_g, int64_t y_stride_b,
                     int64_t y_stride_l, int dtype, int dt_softplus,
                     float dt_min, float dt_max, int group_map, const void* cb, void* workspace,
                     size_t workspace_bytes, void* stream) {
  TV_CHECK_ARG(A && (seqlen == 0 || (x && dt && Bm && Cm && y)), "ssd_scan: null pointer");   // empty tensors have no storage
  TV_CHECK_ARG(batch > 0 && seqlen >= 0 && nheads > 0 && headdim > 0 && ngroups > 0 &&
                   dstate > 0 && nheads % ngroups == 0,
               "ssd_scan: bad sizes (B %d L %d H %d P %d G %d N %d)", batch, seqlen, nheads,
               headdim, ngroups, dstate);
  TV_CHECK_ARG(dtype == TV_F32 || dtype == TV_BF16 || dtype == TV_F16, "ssd_scan: dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (seqlen == 0) {
    // empty sequence: the state passes through unchanged
    if (final_state) {
      const size_t bytes = (size_t)batch * nheads * headdim * dstate * sizeof(float);
      if (init_state) (void)hipMemcpyAsync(final_state, init_state, bytes, hipMemcpyDeviceToDevice, st);
      else (void)hipMemsetAsync(final_state, 0, bytes, st);
    }
    if (total_decay) (void)hipMemsetAsync(total_decay, 0, (size_t)batch * nheads * sizeof(float), st);
    return TV_OK;
  }
  int forced = g_ssd_impl.load(std::memory_order_relaxed);   // one read per call
  if (forced == 2) forced = 3;          // removed kernels: the numbers select their successors
  if (forced == 5) forced = 4;
  if (forced == 7) forced = 6;
  const bool mfma_ok = forced != 1 && forced != 8 && workspace && dt_stride_l % 2 == 0 && dt_stride_b % 2 == 0 &&
                       (((uintptr_t)dt) & 3) == 0;
  const int impl = forced ? forced : kAutoImpl;
  if (mfma_ok && impl == 6 &&
      tv_ssd_head_supported(seqlen, nheads, headdim, ngroups, dstate, dtype, x_stride_l, b_stride_l, b_stride_g,
                            c_stride_l, c_stride_g, y_stride_l, x, Bm, Cm, y)) {
    g_ssd_last.store(6, std::memory_order_relaxed);
    // the head-per-wave march takes every chunk itself (floating, reset and standard steps in the one kernel)
    return tv_ssd_head_launch(x, dt, A, Bm, Cm, D, dt_bias, init_state, y, final_state, total_decay, batch, seqlen,
                              nheads, headdim, ngroups, x_stride_b, x_stride_l, dt_stride_b, dt_stride_l, b_stride_b,
                              b_stride_l, b_stride_g, c_stride_b, c_stride_l, c_stride_g, y_stride_b, y_stride_l,
                              dt_softplus, dt_min, dt_max, group_map, workspace, workspace_bytes, cb, st);
  }
  for (int wide = 1; wide >= 0; --wide) {       // impl 4 (and 6 where it does not apply): whole-head work-groups first, then slices of a head
    if (wide == 1 && impl != 4 && impl != 6) continue;
    if (mfma_ok && impl >= 3 &&
        tv_ssd_slice_supported(seqlen, nheads, headdim, ngroups, dstate, dtype, x_stride_l, b_stride_l,
                               b_stride_g, c_stride_l, c_stride_g, y_stride_l, x, Bm, Cm, y, wide)) {
      g_ssd_last.store(wide == 1 ? 4 : 3, std::memory_order_relaxed);
      return tv_ssd_slice_launch(x, dt, A, Bm, Cm, D, dt_bias, init_state, y, final_state,
                                 total_decay, batch, seqlen, nheads, headdim, ngroups, dstate,
                                 x_stride_b, x_stride_l, dt_stride_b, dt_stride_l, b_stride_b,
                                 b_stride_l, b_stride_g, c_stride_b, c_stride_l, c_stride_g, y_stride_b, y_stride_l, dtype,
                                 dt_softplus, dt_min, dt_max, group_map, workspace, workspace_bytes,
                                 wide, cb, st);
    }
  }
  if (forced >= 3 && forced != 8)
    TV_UNSUPPORTED("ssd_scan: MFMA march kernel forced but shape / dtype / alignment unsupported or no workspace given");
  if ((forced == 0 || forced == 8) && workspace && tv_ssd_chunked_supported(seqlen, headdim, dstate) &&
      workspace_bytes >= tv_ssd_chunked_workspace_bytes(batch, seqlen, nheads, headdim, dstate)) {
    g_ssd_last.store(8, std::memory_order_relaxed);
    return tv_ssd_chunked_launch(x, dt, A, Bm, Cm, D, dt_bias, init_state, y, final_state, total_decay, batch, seqlen,
                                 nheads, headdim, ngroups, dstate, x_stride_b, x_stride_l, dt_stride_b, dt_stride_l,
                                 b_stride_b, b_stride_l, b_stride_g, c_stride_b, c_stride_l, c_stride_g, y_stride_b,
                                 y_stride_l, dtype, dt_softplus, dt_min, dt_max, group_map, workspace, st);
  }
  if (forced == 8) TV_UNSUPPORTED("ssd_scan: chunk-parallel generic kernel forced but shape unsupported");
  g_ssd_last.store(1, std::memory_order_relaxed);
  return tv_ssd_generic_launch(x, dt, A, Bm, Cm, D, dt_bias, init_state, y, final_state,
                               total_decay, batch, seqlen, nheads, headdim, ngroups, dstate,
                               x_stride_b, x_stride_l, dt_stride_b, dt_stride_l, b_stride_b,
                               b_stride_l, b_stride_g, c_stride_b, c_stride_l, c_stride_g, y_stride_b, y_stride_l, dtype,
                               dt_softplus, dt_min, dt_max, group_map, st);
}

extern "C" int tv_ssd_scan_fwd(const void* x, const void* dt, const void* A, const void* Bm,
                               const void* Cm, const void* D, const void* dt_bias,
                               const void* init_state, void* y, void* final_state,
                               void* total_decay, int batch, int seqlen, int nheads, int headdim,
                               int ngroups, int dstate, int64_t x_stride_b, int64_t x_stride_l,
                               int64_t dt_stride_b, int64_t dt_stride_l, int64_t b_stride_b,
                               int64_t b_stride_l, int64_t b_stride_g, int64_t c_stride_b,
                               int64_t c_stride_l, int64_t c_stride_g, int64_t y_stride_b,
                               int64_t y_stride_l, int dtype, int dt_softplus,
                               float dt_min, float dt_max, int group_map, void* workspace,
                               size_t workspace_bytes, void* stream) {
  return scan_impl(x, dt, A, Bm, Cm, D, dt_bias, init_state, y, final_state, total_decay, batch, seqlen, nheads,
                   headdim, ngroups, dstate, x_stride_b, x_stride_l, dt_stride_b, dt_stride_l, b_stride_b, b_stride_l,
                   b_stride_g, c_stride_b, c_stride_l, c_stride_g, y_stride_b, y_stride_l, dtype, dt_softplus, dt_min,
                   dt_max, group_map, nullptr, workspace, workspace_bytes, stream);
}

// The same scan with the causal C.B^T fragments of every (chunk, group) supplied by the caller — the `cb` output of
// tv_causal_conv1d_xbc_cb_fwd on the same B / C (tv_ssd_cb_bytes() bytes) — instead of recomputed by a pre-pass
// that reads B and C back.  Kernels that do not use the fragments (fp32, other d_state) ignore them.
extern "C" int tv_ssd_scan_cb_fwd(const void* x, const void* dt, const void* A, const void* Bm,
                                  const void* Cm, const void* cb, const void* D, const void* dt_bias,
                                  const void* init_state, void* y, void* final_state,
                                  void* total_decay, int batch, int seqlen, int nheads, int headdim,
                                  int ngroups, int dstate, int64_t x_stride_b, int64_t x_stride_l,
                                  int64_t dt_stride_b, int64_t dt_stride_l, int64_t b_stride_b,
                                  int64_t b_stride_l, int64_t b_stride_g, int64_t c_stride_b,
                                  int64_t c_stride_l, int64_t c_stride_g, int64_t y_stride_b,
                                  int64_t y_stride_l, int dtype, int dt_softplus,
                                  float dt_min, float dt_max, int group_map, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  TV_CHECK_ARG(cb == nullptr || (((uintptr_t)cb) & 15) == 0, "ssd_scan_cb: C.B^T buffer must be 16-byte aligned");
  return scan_impl(x, dt, A, Bm, Cm, D, dt_bias, init_state, y, final_state, total_decay, batch, seqlen, nheads,
                   headdim, ngroups, dstate, x_stride_b, x_stride_l, dt_stride_b, dt_stride_l, b_stride_b, b_stride_l,
                   b_stride_g, c_stride_b, c_stride_l, c_stride_g, y_stride_b, y_stride_l, dtype, dt_softplus, dt_min,
                   dt_max, group_map, cb, workspace, workspace_bytes, stream);
}
