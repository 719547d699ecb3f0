"""Operator API of the TimeViper hot path, bound to the gfx950 kernels.

Each function keeps the NAME and ARGUMENT MEANING of the third-party operator
the reference calls (SURVEY.md §8b "inner boundary"), so `NemotronHMamba2Mixer`
and friends read like the reference's:

    causal_conv1d_fn / causal_conv1d_update      modeling_nano.py:619-624 / :495-501
    mamba_chunk_scan_combined                    modeling_nano.py:639-653
    selective_state_update                       modeling_nano.py:528-539
    rmsnorm_fn                                   modeling_nano.py:372-380
    flash_attn_func / _flash_attention_forward   modeling_nano.py:1198-1209
    flash_attn_varlen_qkvpacked_func             flash_attention_class.py:59-66
    scaled_dot_product_attention                 cross_attention.py:310-317

torch is plumbing only (device memory, current stream).  There is no CPU or
eager fallback: tensors must live on the GPU and the HIP library must be built.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from . import _capi
from ._capi import TV_BF16, TV_F16, TV_F32, TimeViperHipError, check

_DT = {torch.float32: TV_F32, torch.bfloat16: TV_BF16, torch.float16: TV_F16}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TimeViperHipError(f"unsupported dtype {t.dtype}") from None


def _gpu(*ts: Optional[torch.Tensor]) -> None:
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise TimeViperHipError(
                "timeviper_amd kernels run on the GPU only (got a CPU tensor); "
                "there is no CPU fallback in the product path"
            )
        # the launch goes to the CURRENT device's stream (`_stream`): a tensor of another device
        # would hand that stream foreign pointers.  One process drives one GPU here.
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise TimeViperHipError(
                f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: "
                "call torch.cuda.set_device() (one process per GPU)")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _rows2d(t: torch.Tensor) -> torch.Tensor:
    """View (..., D) as (rows, D) with one row stride, copying only if needed."""
    if t.stride(-1) != 1:
        t = t.contiguous()
    if t.dim() == 2:
        return t
    try:
        return t.view(-1, t.shape[-1])
    except RuntimeError:
        return t.reshape(-1, t.shape[-1])


# --------------------------------------------------------------------- conv1d
def causal_conv1d_fn(x, weight, bias=None, seq_idx=None, initial_states=None,
                     return_final_states=False, final_states_out=None, activation=None,
                     halo=None):
    """x: (B, C, L) — the reference passes the transposed view of a (B, L, C)
    tensor, which is exactly the channels-last layout the kernel wants.
    weight (C, K), bias (C).  Returns (B, C, L) (again a transposed view).
    `halo` (B, K-1, C): rows preceding this shard (sequence sharding)."""
    if seq_idx is not None or initial_states is not None or return_final_states:
        raise TimeViperHipError("causal_conv1d_fn: seq_idx/initial_states/final_states unsupported")
    if activation not in (None, "silu", "swish"):
        raise TimeViperHipError(f"causal_conv1d_fn: activation {activation!r}")
    _gpu(x, weight, bias, halo)
    B, Cc, L = x.shape
    xl = x.transpose(1, 2)  # (B, L, C)
    if xl.stride(2) != 1:
        xl = xl.contiguous()
    K = weight.shape[-1]
    w = weight.reshape(Cc, K).to(x.dtype).contiguous()
    b = None if bias is None else bias.to(x.dtype).contiguous()
    if halo is not None:
        halo = halo.to(x.dtype).contiguous()
        assert halo.shape == (B, K - 1, Cc)
    y = torch.empty((B, L, Cc), dtype=x.dtype, device=x.device)
    check(_capi.lib().tv_causal_conv1d_fwd(
        _p(xl), _p(w), _p(b), _p(halo), _p(y), B, L, Cc, K, xl.stride(0), xl.stride(1),
        y.stride(0), y.stride(1), _dt(x), int(activation in ("silu", "swish")), _stream()),
        "tv_causal_conv1d_fwd")
    return y.transpose(1, 2)


def causal_conv1d_xbc(xBC, weight, bias, d_inner: int, ngroups: int, dstate: int,
                      activation="silu", halo=None, return_cb: bool = False):
    """Mamba-2 mixer variant of causal_conv1d_fn + the [x | B | C] split of
    modeling_nano.py:628-636 in one pass.  xBC (B, L, d_inner + 2*G*N) (any row stride).
    Returns x (B, L, d_inner) and B, C as (B, L, G, N) VIEWS of group-major (B, G, L, N)
    storage — the layout the scan kernel streams best.
    `return_cb`: also return the causal C.B^T fragments of every (chunk, group) for
    `mamba_chunk_scan_combined(..., cb=...)` (an opaque bf16 tensor; None where the scan would
    not use them: other dtypes / d_state), computed while the B / C tiles are on the chip."""
    _gpu(xBC, weight, bias, halo)
    Bsz, L, Cc = xBC.shape
    assert Cc == d_inner + 2 * ngroups * dstate
    if xBC.stride(2) != 1:
        xBC = xBC.contiguous()
    K = weight.shape[-1]
    w = weight.reshape(Cc, K).to(xBC.dtype).contiguous()
    b = None if bias is None else bias.to(xBC.dtype).contiguous()
    if halo is not None:
        halo = halo.to(xBC.dtype).contiguous()
    yx = torch.empty((Bsz, L, d_inner), dtype=xBC.dtype, device=xBC.device)
    yb = torch.empty((Bsz, ngroups, L, dstate), dtype=xBC.dtype, device=xBC.device)
    yc = torch.empty((Bsz, ngroups, L, dstate), dtype=xBC.dtype, device=xBC.device)
    lib = _capi.lib()
    if return_cb and xBC.dtype == torch.bfloat16 and dstate == 128 and L > 0:
        cb = torch.empty(lib.tv_ssd_cb_bytes(Bsz, L, ngroups) // 2, dtype=torch.bfloat16, device=xBC.device)
        check(lib.tv_causal_conv1d_xbc_cb_fwd(
            _p(xBC), _p(w), _p(b), _p(halo), _p(yx), _p(yb), _p(yc), _p(cb), Bsz, L, d_inner, ngroups, dstate,
            K, xBC.stride(0), xBC.stride(1), _dt(xBC), int(activation in ("silu", "swish")), _stream()),
            "tv_causal_conv1d_xbc_cb_fwd")
        return yx, yb.transpose(1, 2), yc.transpose(1, 2), cb
    check(lib.tv_causal_conv1d_xbc_fwd(
        _p(xBC), _p(w), _p(b), _p(halo), _p(yx), _p(yb), _p(yc), Bsz, L, d_inner, ngroups, dstate,
        K, xBC.stride(0), xBC.stride(1), _dt(xBC), int(activation in ("silu", "swish")), _stream()),
        "tv_causal_conv1d_xbc_fwd")
    if return_cb:
        return yx, yb.transpose(1, 2), yc.transpose(1, 2), None
    return yx, yb.transpose(1, 2), yc.transpose(1, 2)


def causal_conv1d_update(x, conv_state, weight, bias=None, activation=None):
    """x (B, C); conv_state (B, C, K) updated in place; returns (B, C)."""
    _gpu(x, conv_state, weight, bias)
    B, Cc = x.shape
    K = weight.shape[-1]
    if not conv_state.is_contiguous() or conv_state.dtype != x.dtype:
        raise TimeViperHipError("causal_conv1d_update: conv_state must be contiguous, x.dtype")
    xc = x.contiguous()
    w = weight.reshape(Cc, K).to(x.dtype).contiguous()
    b = None if bias is None else bias.to(x.dtype).contiguous()
    y = torch.empty_like(xc)
    check(_capi.lib().tv_causal_conv1d_update(
        _p(xc), _p(conv_state), _p(w), _p(b), _p(y), B, Cc, K, _dt(x),
        int(activation in ("silu", "swish")), _stream()), "tv_causal_conv1d_update")
    return y


# ---------------------------------------------------------------------- norms
def rms_norm(x, weight, eps, residual=None, return_sum=False):
    """NemotronHRMSNorm (modeling_nano.py:897-903); with `residual`, normalises
    s = x + residual (rounded to x.dtype like the reference's bf16 add, :966) and
    can return s too."""
    _gpu(x, weight, residual)
    x2 = _rows2d(x)
    r2 = None if residual is None else _rows2d(residual)
    y = torch.empty(x2.shape, dtype=x.dtype, device=x.device)
    s = torch.empty_like(y) if (return_sum and residual is not None) else None
    w = weight if weight.dtype in (torch.float32, x.dtype) else weight.to(torch.float32)
    w = w.contiguous()
    check(_capi.lib().tv_rmsnorm_fwd(
        _p(x2), _p(r2), _p(w), _p(s), _p(y), x2.shape[0], x2.shape[1], x2.stride(0),
        0 if r2 is None else r2.stride(0), 0 if s is None else s.stride(0), y.stride(0),
        float(eps), _dt(x), _dt(w), _stream()), "tv_rmsnorm_fwd")
    y = y.view(x.shape)
    if return_sum:
        return y, (s.view(x.shape) if s is not None else x)
    return y


def layer_norm(x, weight, bias, eps, residual=None, return_sum=False, row_bias=None):
    """nn.LayerNorm over the last dim, optionally of s = x + residual (rounded to x.dtype),
    returning s too — the fused form of a pre-norm ViT block's residual add + norm.
    `row_bias` (D,) fp32: a constant row added to every s before the statistics (not to the
    returned sum): the projection biases a caller carries beside a stream its GEMMs accumulate into."""
    _gpu(x, weight, bias, residual, row_bias)
    x2 = _rows2d(x)
    r2 = None if residual is None else _rows2d(residual)
    y = torch.empty(x2.shape, dtype=x.dtype, device=x.device)
    s = torch.empty_like(y) if (return_sum and residual is not None) else None
    w = weight.to(x.dtype).contiguous()
    b = None if bias is None else bias.to(x.dtype).contiguous()
    rb = None if row_bias is None else row_bias.to(torch.float32).contiguous()
    check(_capi.lib().tv_layernorm_fwd(
        _p(x2), _p(r2), _p(w), _p(b), _p(rb), _p(s), _p(y), x2.shape[0], x2.shape[1], x2.stride(0),
        0 if r2 is None else r2.stride(0), 0 if s is None else s.stride(0), y.stride(0), float(eps),
        _dt(x), _stream()), "tv_layernorm_fwd")
    y = y.view(x.shape)
    if return_sum:
        return y, (s.view(x.shape) if s is not None else x)
    return y


def gelu(x, inplace=False):
    """exact (erf) GELU, elementwise."""
    _gpu(x)
    xc = x if x.is_contiguous() else x.contiguous()
    y = xc if inplace else torch.empty_like(xc)
    check(_capi.lib().tv_gelu_fwd(_p(xc), _p(y), xc.numel(), _dt(xc), _stream()), "tv_gelu_fwd")
    return y


def relu2(x, inplace=False):
    """square(relu(x)) — NemotronHMLP's activation (modeling_nano.py:993-994), elementwise."""
    _gpu(x)
    xc = x if x.is_contiguous() else x.contiguous()
    y = xc if inplace else torch.empty_like(xc)
    check(_capi.lib().tv_relu2_fwd(_p(xc), _p(y), xc.numel(), _dt(xc), _stream()), "tv_relu2_fwd")
    return y


def rmsnorm_fn(x, weight, bias=None, z=None, eps=1e-6, group_size=None,
               norm_before_gate=True, upcast=True):
    """mamba_ssm.ops.triton.layernorm_gated.rmsnorm_fn as the reference calls it
    (norm_before_gate=False): y = w * u * rsqrt(mean_group(u^2)+eps), u = x*silu(z)."""
    if bias is not None:
        raise TimeViperHipError("rmsnorm_fn: bias unsupported (reference passes None)")
    if z is not None and norm_before_gate:
        raise TimeViperHipError("rmsnorm_fn: only norm_before_gate=False is on the path")
    _gpu(x, weight, z)
    x2 = _rows2d(x)
    z2 = None if z is None else _rows2d(z)
    D = x2.shape[1]
    gs = D if group_size is None else int(group_size)
    y = torch.empty(x2.shape, dtype=x.dtype, device=x.device)
    w = weight if weight.dtype in (torch.float32, x.dtype) else weight.to(torch.float32)
    w = w.contiguous()
    check(_capi.lib().tv_rmsnorm_gated_fwd(
        _p(x2), _p(z2), _p(w), _p(y), x2.shape[0], D, gs, x2.stride(0),
        0 if z2 is None else z2.stride(0), y.stride(0), float(eps), _dt(x), _dt(w), _stream()),
        "tv_rmsnorm_gated_fwd")
    return y.view(x.shape)


# ------------------------------------------------------------------- linears with fused epilogues
GEMM_BIAS, GEMM_BIAS_GELU, GEMM_ACCUM = 0, 1, 2


def linear_fused(x, weight, bias=None, epilogue: int = GEMM_BIAS, out=None):
    """F.linear(x, weight, bias) on the hand-written bf16 GEMM (csrc/gemm.hip) with the epilogue fused:
    GEMM_BIAS (plain), GEMM_BIAS_GELU (exact GELU of the bf16-rounded result: the timm Mlp's fc1 + act) or
    GEMM_ACCUM (out += x @ weight.T, `out` required, bias ignored).  x (..., K) bf16 with dense rows,
    weight (N, K) bf16; K % 128 == 0, N % 4 == 0."""
    _gpu(x, weight, bias, out)
    if x.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16:
        raise TimeViperHipError("linear_fused: bf16 only")
    x2 = _rows2d(x)
    N, Kd = weight.shape
    if weight.stride(1) != 1:
        weight = weight.contiguous()
    if epilogue == GEMM_ACCUM:
        if out is None:
            raise TimeViperHipError("linear_fused: GEMM_ACCUM needs `out`")
        bias = None
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
    o2 = out.view(-1, N) if out.dim() != 2 else out
    if o2.stride(1) != 1 or o2.shape[0] != x2.shape[0]:
        raise TimeViperHipError("linear_fused: `out` must be (rows, N) with a contiguous last dim")
    bd = TV_F32
    if bias is not None:
        if bias.dtype not in (torch.float32, torch.bfloat16):
            bias = bias.float()
        bias = bias.contiguous()
        bd = _dt(bias)
    check(_capi.lib().tv_gemm_bf16_fwd(_p(x2), _p(weight), _p(bias), _p(o2), x2.shape[0], N, Kd, x2.stride(0),
                                       weight.stride(0), o2.stride(0), int(epilogue), bd, _stream()),
          "tv_gemm_bf16_fwd")
    return out


def param_key(params):
    """Cache key for tensors derived from parameters (stacked / padded / fp32 copies): storage, version counter, dtype
    and device of each.  Tensors created under `torch.inference_mode()` track no version ("Inference tensors do not
    track version counter"): they count as version 0 — such a model must not be written in place after its first forward.
    Note that writes through `.data` never bump the counter either: reload weights with `load_state_dict` / `copy_`."""
    key = []
    for p in params:
        try:
            v = p._version
        except RuntimeError:
            v = 0
        key.append((p.data_ptr(), v, p.dtype, p.device))
    return tuple(key)


def gemm_set_persist(mode: int = -1, grid: int = 0) -> None:
    """Which of the two GEMM kernels `linear_fused` runs on: -1 automatic (the persistent kernel from 4 tiles per
    compute unit on), 0 the per-tile kernel only, 1 the persistent kernel wherever the shape allows; `grid`:
    work-groups of the persistent kernel (0 = one per compute unit; tests use 8 so that small problems still give
    every work-group several tiles)."""
    _capi.lib().tv_gemm_set_persist(int(mode), int(grid))


# zero-padded inference copies of weights (model/vit/siglip.py): storage address -> (useful rows, useful columns), so that a
# FLOP count (bench.py) can leave the padding out whatever view of the copy a GEMM receives
PADDED_USEFUL: dict = {}


def gemm_set_drip(mode: int = -1) -> None:
    """The 256 x 192-tile persistent GEMM (csrc/gemm_drip.hip): -1 automatic (first choice wherever the shape allows and
    every compute unit gets a few tiles), 0 never, 1 wherever the shape allows (tests on small grids)."""
    _capi.lib().tv_gemm_set_drip(int(mode))


# ------------------------------------------------------------------- SSD scan
def _row_view(t: torch.Tensor, inner: int):
    """(B, L, ...) tensor whose trailing dims are contiguous with `inner`
    elements per row: return (tensor, stride_b, stride_l) without copying when
    the rows are dense (slices of a wider projection keep their row stride)."""
    ok = t.stride(-1) == 1
    exp = 1
    for d in range(t.dim() - 1, 1, -1):
        ok = ok and t.stride(d) == exp
        exp *= t.shape[d]
    if not ok:
        t = t.contiguous()
    return t, t.stride(0), t.stride(1)


def mamba_chunk_scan_combined(x, dt, A, B, C, chunk_size=None, D=None, z=None, dt_bias=None,
                              initial_states=None, seq_idx=None, cu_seqlens=None,
                              dt_softplus=False, dt_limit=(0.0, float("inf")),
                              return_final_states=False, return_varlen_states=False,
                              group_map="block", return_total_decay=False, cb=None):
    """x (B,L,H,P), dt (B,L,H), A (H), B/C (B,L,G,N), D (H), dt_bias (H),
    initial_states (B,H,P,N).  Returns y (B,L,H,P) [, final_states (B,H,P,N) fp32]
    [, total_decay (B,H) fp32].  `chunk_size` is accepted for signature parity
    and ignored (the result is chunk-invariant).  `group_map`: "block"
    (h // (H/G), GPU reference) or "tile" (h % G, reference CPU quirk).
    `cb`: the C.B^T fragments `causal_conv1d_xbc(..., return_cb=True)` returned for these B / C."""
    if z is not None or seq_idx is not None or cu_seqlens is not None or return_varlen_states:
        raise TimeViperHipError("mamba_chunk_scan_combined: z/seq_idx/cu_seqlens unsupported")
    if D is not None and D.dim() != 1:
        raise TimeViperHipError("mamba_chunk_scan_combined: D must be (nheads,)")
    _gpu(x, dt, A, B, C, D, dt_bias, initial_states)
    Bsz, L, H, P = x.shape
    G, N = B.shape[2], B.shape[3]
    if dt.dtype != x.dtype:
        dt = dt.to(x.dtype)
    if B.dtype != x.dtype:
        B = B.to(x.dtype)
    if C.dtype != x.dtype:
        C = C.to(x.dtype)
    x, xsb, xsl = _row_view(x, H * P)
    dt, dsb, dsl = _row_view(dt, H)
    # B/C: any (B, L, G, N) view with contiguous N (token-major rows or group-major storage)
    if B.stride(3) != 1:
        B = B.contiguous()
    if C.stride(3) != 1:
        C = C.contiguous()
    bsb, bsl, bsg = B.stride(0), B.stride(1), B.stride(2)
    csb, csl, csg = C.stride(0), C.stride(1), C.stride(2)
    f32 = lambda t: None if t is None else t.to(torch.float32).contiguous()
    A, D, dt_bias, initial_states = f32(A), f32(D), f32(dt_bias), f32(initial_states)
    y = torch.empty((Bsz, L, H, P), dtype=x.dtype, device=x.device)
    final = torch.empty((Bsz, H, P, N), dtype=torch.float32, device=x.device) \
        if return_final_states else None
    decay = torch.empty((Bsz, H), dtype=torch.float32, device=x.device) \
        if return_total_decay else None
    lib = _capi.lib()
    ws_bytes = lib.tv_ssd_scan_workspace_bytes(Bsz, L, H, P, G, N, _dt(x))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    if cb is not None:
        _gpu(cb)
        if cb.dtype != torch.bfloat16 or cb.numel() * 2 != lib.tv_ssd_cb_bytes(Bsz, L, G) or not cb.is_contiguous():
            raise TimeViperHipError("mamba_chunk_scan_combined: cb is not the C.B^T buffer of these shapes")
    check(lib.tv_ssd_scan_cb_fwd(
        _p(x), _p(dt), _p(A), _p(B), _p(C), _p(cb), _p(D), _p(dt_bias), _p(initial_states), _p(y),
        _p(final), _p(decay), Bsz, L, H, P, G, N, xsb, xsl, dsb, dsl, bsb, bsl, bsg, csb, csl, csg,
        y.stride(0), y.stride(1), _dt(x), int(bool(dt_softplus)), float(dt_limit[0]),
        float(min(dt_limit[1], 3.0e38)), {"block": 0, "tile": 1}[group_map], _p(ws), ws_bytes,
        _stream()), "tv_ssd_scan_cb_fwd")
    out = (y,)
    if return_final_states:
        out += (final,)
    if return_total_decay:
        out += (decay,)
    return out[0] if len(out) == 1 else out


def ssd_state_correction(y, dt, A, C, state_in, dt_bias=None, dt_softplus=False,
                         dt_limit=(0.0, float("inf")), group_map="block"):
    """In place: y_t += exp(sum_{j<=t} dt_j A_h) C_t . state_in[h] — completes the outputs of a scan
    that started from a zero state once the state entering it is known (sequence shards, SURVEY
    Appendix A).  y (B,L,H,P), dt (B,L,H) raw, C (B,L,G,N), state_in (B,H,P,N) fp32.  Returns y."""
    _gpu(y, dt, A, C, state_in, dt_bias)
    Bsz, L, H, P = y.shape
    G, N = C.shape[2], C.shape[3]
    assert y.stride(3) == 1 and y.stride(2) == P, "y rows must be dense (H*P elements)"
    if dt.dtype != y.dtype:
        dt = dt.to(y.dtype)
    dt, dsb, dsl = _row_view(dt, H)
    if C.dtype != y.dtype:
        C = C.to(y.dtype)
    if C.stride(3) != 1:
        C = C.contiguous()
    f32 = lambda t: None if t is None else t.to(torch.float32).contiguous()
    A, dt_bias, state_in = f32(A), f32(dt_bias), f32(state_in)
    lib = _capi.lib()
    ws_bytes = lib.tv_ssd_state_correction_workspace_bytes(Bsz, L, H)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=y.device)
    check(lib.tv_ssd_state_correction(
        _p(y), _p(dt), _p(A), _p(C), _p(dt_bias), _p(state_in), Bsz, L, H, P, G, N,
        y.stride(0), y.stride(1), dsb, dsl, C.stride(0), C.stride(1), C.stride(2), _dt(y),
        int(bool(dt_softplus)), float(dt_limit[0]), float(min(dt_limit[1], 3.0e38)),
        {"block": 0, "tile": 1}[group_map], _p(ws), ws_bytes, _stream()), "tv_ssd_state_correction")
    return y


def selective_state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None,
                           dt_softplus=False):
    """Single decode step.  state (B,H,P,N) fp32, in place.  The reference passes
    A/dt/dt_bias/D expanded over (P[,N]) from per-head values (:514-522); the
    kernel takes the per-head values, so expanded views are reduced here."""
    if z is not None:
        raise TimeViperHipError("selective_state_update: z unsupported (reference passes None)")
    _gpu(state, x, dt, A, B, C, D, dt_bias)
    Bsz, H, P = x.shape
    G, N = B.shape[1], B.shape[2]
    if state.dtype != torch.float32 or not state.is_contiguous():
        raise TimeViperHipError("selective_state_update: state must be contiguous fp32")
    head = lambda t, nd: None if t is None else \
        (t if t.dim() == nd else t[(..., *([0] * (t.dim() - nd)))]).to(torch.float32).contiguous()
    A1, D1, b1 = head(A, 1), head(D, 1), head(dt_bias, 1)
    dt1 = (dt if dt.dim() == 2 else dt[..., 0]).to(x.dtype).contiguous()
    xc, Bc, Cc = x.contiguous(), B.to(x.dtype).contiguous(), C.to(x.dtype).contiguous()
    y = torch.empty_like(xc)
    check(_capi.lib().tv_selective_state_update(
        _p(state), _p(xc), _p(dt1), _p(A1), _p(Bc), _p(Cc), _p(D1), _p(b1), _p(y), Bsz, H, P, G,
        N, _dt(x), int(bool(dt_softplus)), _stream()), "tv_selective_state_update")
    return y


def ssd_scan_set_impl(impl: int) -> None:
    """0 auto (= 6 where it applies, else 4, 3; other dtypes / d_state: 8, else 1), 1 generic fp32 token recurrence, 3 MFMA
    slice march on <= 40-column slices of a head, 4 whole-head slice march x sequence segments, 6 head-per-wave march
    (ssd_head.hip), 8 chunk-parallel fp32 form (ssd_chunked.hip); 2 / 5 / 7 (removed kernels) select 3 / 4 / 6.
    Process-global (dev tools and tests)."""
    _capi.lib().tv_ssd_scan_set_impl(int(impl))


def ssd_head_set_asm(on: int) -> None:
    """Head-per-wave march at head_dim 80 x 4 heads per work-group: 1 the generated step (default), 0 the C++ step, -1 default /
    TV_HEAD_ASM.  Process-global (A/B runs and the bit-identity test)."""
    _capi.lib().tv_ssd_head_set_asm(int(on))


def ssd_scan_last_impl() -> int:
    """Kernel family (the numbers of `ssd_scan_set_impl`) the most recent scan call of this process ran on."""
    return int(_capi.lib().tv_ssd_scan_last_impl())


# ------------------------------------------------------------------ attention
def flash_attn_set_variant(variant: int) -> None:
    """0 auto (bf16 ViT frames: the kernel with the generated tile loop, csrc/attention_vit.hpp; else the compiled streaming
    kernel), 5 always the compiled streaming kernel (P's row sums out of the P.V MFMAs), 3 the compiled kernel with the row
    sums on the vector pipe (A/B, tests); include/timeviper_hip.h.  Process-global (dev tools and tests)."""
    _capi.lib().tv_flash_attn_set_variant(int(variant))


_ATTN_FP8 = {"on": False, "min_keys": 4096, "min_queries": 64}


class fp8_attention:
    """Context manager / switch: inside it `flash_attn_func` (and everything built on it) runs the
    QK^T / PV products on the FP8 MFMA path (`tv_flash_attn_fp8_fwd`) — BASELINE config 5.  Off by
    default: the reference's attention arithmetic is bf16.  Key sequences shorter than `min_keys`
    stay on the bf16 kernel: the ViT towers (<= 1 025 keys per frame / tube; measured 2.3 ms fp8
    against 1.5 ms bf16 per 256 SigLIP frames — the quantisation pre-pass and head_dim 72 -> 128
    padding cost more than the MFMAs save); calls with fewer than `min_queries` queries stay there too: the
    decode step (one query against the whole cache: the quantisation pre-pass over the cache would cost
    more than the attention, with per-head scales that differ from the prefill's).  The long causal LM
    attention is where the fp8 matrix rate pays (131 172 tokens, 28/4 x 128: 83 ms against 122 ms).
    The switch is process-global state (like `ssd_scan_set_impl`): right for an evaluation process, not for
    two models with different settings in one process."""

    def __init__(self, on: bool = True, min_keys: int = 4096, min_queries: int = 64):
        self.new = {"on": bool(on), "min_keys": int(min_keys), "min_queries": int(min_queries)}

    def __enter__(self):
        self.old = dict(_ATTN_FP8)
        _ATTN_FP8.update(self.new)
        return self

    def __exit__(self, *exc):
        _ATTN_FP8.update(self.old)


def flash_attn_fp8_func(q, k, v, softmax_scale=None, causal=False, return_lse=False):
    """`flash_attn_func` with e4m3 MFMA operands (per-head scales, fp32 accumulation / softmax)."""
    _gpu(q, k, v)
    B, Lq, Hq, D = q.shape
    Lk, Hkv = k.shape[1], k.shape[2]
    fix = lambda t: t if t.stride(-1) == 1 else t.contiguous()
    q, k, v = fix(q), fix(k), fix(v)
    if D % 8 or D > 128:
        raise TimeViperHipError(f"flash_attn_fp8_func: head_dim {D} must be a multiple of 8, <= 128")
    for t in (q, k, v):
        if any(s % 8 for s in t.stride()[:3]) or t.data_ptr() % 16:
            raise TimeViperHipError("flash_attn_fp8_func: strides must be multiples of 8 elements")
    scale = 1.0 / math.sqrt(D) if softmax_scale is None else float(softmax_scale)
    o = torch.empty((B, Lq, Hq, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, Hq, Lq), dtype=torch.float32, device=q.device) if return_lse else None
    lib = _capi.lib()
    ws_bytes = lib.tv_flash_attn_fp8_workspace_bytes(B, Lq, Lk, Hq, Hkv)
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=q.device)
    check(lib.tv_flash_attn_fp8_fwd(
        _p(q), _p(k), _p(v), _p(o), _p(lse), B, Lq, Lk, Hq, Hkv, D,
        q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
        v.stride(0), v.stride(1), v.stride(2), o.stride(0), o.stride(1), o.stride(2),
        scale, int(bool(causal)), _dt(q), _p(ws), ws_bytes, _stream()), "tv_flash_attn_fp8_fwd")
    return (o, lse) if return_lse else o


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False,
                    return_lse=False):
    """q (B,Lq,Hq,D), k/v (B,Lk,Hkv,D) -> (B,Lq,Hq,D).  GQA without repeat_kv;
    causal mask bottom-right aligned (flash-attn >= 2.1 semantics)."""
    if dropout_p:
        raise TimeViperHipError("flash_attn_func: dropout is not on the inference path")
    _gpu(q, k, v)
    B, Lq, Hq, D = q.shape
    Lk, Hkv = k.shape[1], k.shape[2]
    # (Lq >= 64: a decode step against a long cache stays on the bf16 kernel — the fp8 path would re-quantise and
    # transpose the whole K / V cache for one query, with a scale that differs from the prefill's)
    if _ATTN_FP8["on"] and Lk >= _ATTN_FP8["min_keys"] and Lq >= _ATTN_FP8["min_queries"] and D <= 128 \
            and q.dtype in (torch.bfloat16, torch.float16):
        return flash_attn_fp8_func(q, k, v, softmax_scale, causal, return_lse)
    fix = lambda t: t if t.stride(-1) == 1 else t.contiguous()
    q, k, v = fix(q), fix(k), fix(v)
    if Lq == 1 and _decode_attn_takes(q, k, v):         # a decode step: split-KV kernel (causal or not: one query, last row)
        return flash_attn_decode(q, k, v, softmax_scale=softmax_scale, return_lse=return_lse)
    if D % 8:
        raise TimeViperHipError(f"flash_attn_func: head_dim {D} must be a multiple of 8")
    for t in (q, k, v):
        if any(s % 8 for s in t.stride()[:3]) or t.data_ptr() % 16:
            raise TimeViperHipError("flash_attn_func: strides must be multiples of 8 elements")
    scale = 1.0 / math.sqrt(D) if softmax_scale is None else float(softmax_scale)
    o = torch.empty((B, Lq, Hq, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, Hq, Lq), dtype=torch.float32, device=q.device) if return_lse else None
    check(_capi.lib().tv_flash_attn_fwd(
        _p(q), _p(k), _p(v), _p(o), _p(lse), B, Lq, Lk, Hq, Hkv, D,
        q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
        v.stride(0), v.stride(1), v.stride(2), o.stride(0), o.stride(1), o.stride(2),
        scale, int(bool(causal)), _dt(q), _stream()), "tv_flash_attn_fwd")
    return (o, lse) if return_lse else o


GEMV_NONE, GEMV_RMSNORM, GEMV_RELU2, GEMV_GATED = 0, 1, 2, 3


def gemv_takes(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Shapes tv_gemv_bf16_fwd is written for: 1..4 rows of bf16 against a bf16 (N, K) weight, f(x) within 128 KiB of LDS."""
    K = x.shape[-1]
    rows = x.numel() // max(K, 1)
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.dim() == 2
            and 1 <= rows <= 4 and K % 8 == 0 and rows * K * 2 <= 128 * 1024 and weight.stride(1) == 1
            and weight.stride(0) % 8 == 0)


def gemv_fused(x, weight, bias=None, prologue=GEMV_NONE, delta=None, sum_out=None, norm_weight=None, eps=0.0,
               gate=None, group_size=0, conv=None):
    """y = f(x) @ weight.T (+ bias) for 1..4 rows — the linear layers of a decode step (torch.nn.Linear at q_len 1) with
    the single-row operator in front of them computed in the kernel's prologue: GEMV_RMSNORM (NemotronHRMSNorm with the
    block's residual add: `delta` is added to x first, the sum goes to `sum_out`), GEMV_RELU2 (the MLP activation),
    GEMV_GATED (MambaRMSNormGated with `gate`), rounding where the stand-alone operators round.
    `conv=(conv_state, weight, bias, row0)`: the outputs [row0, row0 + C) go through `causal_conv1d_update` (width 4,
    SiLU) on conv_state (B, C, 4) before they are stored — the mixer's in_proj -> conv pair as one launch (K < 8192)."""
    _gpu(x, weight, bias, delta, sum_out, norm_weight, gate)
    K = x.shape[-1]
    x2 = _rows2d(x)
    M, N = x2.shape[0], weight.shape[0]
    if weight.shape[1] != K:
        raise TimeViperHipError(f"gemv_fused: weight {tuple(weight.shape)} against rows of {K}")
    # the C ABI carries no weight dtype: anything but dense bf16 rows would be re-interpreted silently
    if (x.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16 or weight.stride(1) != 1 or weight.stride(0) % 8
            or K % 8):
        raise TimeViperHipError("gemv_fused: bf16 x and weight with dense rows of a multiple of 8 elements only "
                                f"(x {x.dtype}, weight {weight.dtype}, strides {tuple(weight.stride())}, K {K})")
    y = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
    d2 = None if delta is None else _rows2d(delta)
    g2 = None if gate is None else _rows2d(gate)
    s2 = None if sum_out is None else sum_out.view(-1, K)
    for t in (d2, g2, s2):
        if t is not None and (t.shape != x2.shape or t.dtype != x.dtype):
            raise TimeViperHipError("gemv_fused: delta / gate / sum_out must have x's shape and dtype")
    if bias is not None and (bias.dtype != x.dtype or not bias.is_contiguous()):
        raise TimeViperHipError("gemv_fused: bias must be contiguous, of x's dtype")
    nw_dt = 0
    if norm_weight is not None:
        if not norm_weight.is_contiguous() or norm_weight.numel() != K:
            raise TimeViperHipError("gemv_fused: norm_weight must be a contiguous vector of K entries")
        nw_dt = _dt(norm_weight)
    cst = cw = cb = None
    row0 = nch = 0
    if conv is not None:
        cst, cw, cb, row0 = conv
        _gpu(cst, cw, cb)
        nch = cst.shape[1]
        if (cst.dtype != x.dtype or cw.dtype != x.dtype or not cst.is_contiguous() or not cw.is_contiguous()
                or cst.shape != (M, nch, 4) or cw.shape != (nch, 4) or (cb is not None and (cb.dtype != x.dtype or not cb.is_contiguous()))):
            raise TimeViperHipError("gemv_fused: conv needs a contiguous (rows, C, 4) state and (C, 4) weight of x's dtype")
    check(_capi.lib().tv_gemv_bf16_fwd(
        _p(x2), _p(weight), _p(bias), _p(y), M, N, K, x2.stride(0), weight.stride(0), N, int(prologue),
        _p(d2), 0 if d2 is None else d2.stride(0), _p(s2), 0 if s2 is None else s2.stride(0),
        _p(norm_weight), nw_dt, float(eps), _p(g2), 0 if g2 is None else g2.stride(0), int(group_size),
        _p(cst), _p(cw), _p(cb), int(row0), int(nch), _stream()), "tv_gemv_bf16_fwd")
    return y


def _decode_attn_takes(q, k, v) -> bool:
    """Shapes tv_attn_decode_fwd is written for (the rest stays on tv_flash_attn_fwd with one query row)."""
    return (q.dtype == torch.bfloat16 and q.shape[-1] == 128 and k.shape[1] >= 256
            and os.environ.get("TV_ATTN_DECODE", "1") != "0"
            and all(s % 8 == 0 for t in (q, k, v) for s in t.stride()[:3])
            and all(t.data_ptr() % 16 == 0 for t in (q, k, v)))


def flash_attn_decode(q, k, v, seqlens_k=None, softmax_scale=None, return_lse=False, out=None):
    """One query token per sequence against a K / V cache: q (B, 1, Hq, 128) bf16, k / v (B, Lk, Hkv, 128) ->
    (B, 1, Hq, 128) (the q_len == 1 call of modeling_nano.py:1198-1209).  `seqlens_k`: int32 tensor (B,) ON THE GPU with
    the keys in use per sequence (<= Lk, the capacity of the cache buffers) — the form a captured decode step replays
    with; None: all Lk keys."""
    _gpu(q, k, v, seqlens_k)
    B, Lq, Hq, D = q.shape
    Lk, Hkv = k.shape[1], k.shape[2]
    if Lq != 1:
        raise TimeViperHipError("flash_attn_decode: one query token per sequence")
    if seqlens_k is not None and (seqlens_k.dtype != torch.int32 or seqlens_k.numel() != B or not seqlens_k.is_contiguous()):
        raise TimeViperHipError("flash_attn_decode: seqlens_k must be a contiguous int32 tensor of `batch` entries")
    fix = lambda t: t if t.stride(-1) == 1 else t.contiguous()
    q, k, v = fix(q), fix(k), fix(v)
    scale = 1.0 / math.sqrt(D) if softmax_scale is None else float(softmax_scale)
    o = torch.empty((B, 1, Hq, D), dtype=q.dtype, device=q.device) if out is None else out
    if o.shape != (B, 1, Hq, D) or o.dtype != q.dtype or o.stride(-1) != 1:
        raise TimeViperHipError("flash_attn_decode: out must be (B, 1, Hq, D) of q's dtype with unit last stride")
    lse = torch.empty((B, Hq, 1), dtype=torch.float32, device=q.device) if return_lse else None
    lib = _capi.lib()
    ws = torch.empty((max(int(lib.tv_attn_decode_workspace_bytes(B, Hq, Hkv, Lk)), 16),), dtype=torch.uint8,
                     device=q.device)
    check(lib.tv_attn_decode_fwd(
        _p(q), _p(k), _p(v), _p(o), _p(lse), B, Lk, _p(seqlens_k), Hq, Hkv, D,
        q.stride(0), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
        v.stride(0), v.stride(1), v.stride(2), o.stride(0), o.stride(2),
        scale, _dt(q), _p(ws), ws.numel() * ws.element_size(), _stream()), "tv_attn_decode_fwd")
    return (o, lse) if return_lse else o


def _flash_attention_forward(query_states, key_states, value_states, attention_mask=None,
                             query_length=None, is_causal=True, dropout=0.0, position_ids=None,
                             softmax_scale=None, sliding_window=None, use_top_left_mask=False,
                             **kwargs):
    """transformers' helper as NemotronHFlashAttention2 calls it (:1198-1209)."""
    if attention_mask is not None or sliding_window is not None:
        raise TimeViperHipError("_flash_attention_forward: padding mask / sliding window unsupported")
    causal = is_causal and not (use_top_left_mask and query_states.shape[1] == 1)
    return flash_attn_func(query_states, key_states, value_states, dropout,
                           softmax_scale=softmax_scale, causal=causal)


def scaled_dot_product_attention(query, key, value, attn_mask=None, dropout_p=0.0,
                                 is_causal=False, scale=None):
    """F.scaled_dot_product_attention layout: (B, H, L, D) in and out."""
    if attn_mask is not None:
        raise TimeViperHipError("scaled_dot_product_attention: attn_mask unsupported")
    o = flash_attn_func(query.transpose(1, 2), key.transpose(1, 2), value.transpose(1, 2),
                        dropout_p, softmax_scale=scale, causal=is_causal)
    return o.transpose(1, 2)


def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0,
                                     softmax_scale=None, causal=False, **kwargs):
    """qkv (nnz, 3, H, D), sequences [cu_seqlens[i], cu_seqlens[i+1]) (flash_attention_class.py:59-66, :68-91).
    Equal lengths — what the InternVideo2 ViT produces: every clip has the same token count — are ONE launch over
    a (nseq, len) view; a ragged batch (the `key_padding_mask` / `unpad_input` path of the reference) runs one launch per
    run of consecutive equal-length sequences.  The boundaries are read on the host: one device-to-host copy of
    `cu_seqlens` when it lives on the GPU (the reference's unpad_input synchronises for max_seqlen as well)."""
    nnz, three, H, D = qkv.shape
    assert three == 3
    nseq = cu_seqlens.numel() - 1
    if nseq > 0 and nnz == nseq * max_seqlen:                # equal lengths: no host read
        x = qkv.view(nseq, max_seqlen, 3, H, D)
        o = flash_attn_func(x[:, :, 0], x[:, :, 1], x[:, :, 2], dropout_p, softmax_scale, causal)
        return o.reshape(nnz, H, D)
    cu = [int(v) for v in cu_seqlens.tolist()]
    if cu[0] != 0 or cu[-1] != nnz or any(b < a for a, b in zip(cu, cu[1:])):
        raise TimeViperHipError("flash_attn_varlen_qkvpacked_func: cu_seqlens must rise from 0 to nnz")
    out = torch.empty((nnz, H, D), dtype=qkv.dtype, device=qkv.device)
    i = 0
    while i < nseq:
        n = cu[i + 1] - cu[i]
        j = i + 1
        while j < nseq and cu[j + 1] - cu[j] == n:
            j += 1
        if n > 0:
            x = qkv[cu[i]:cu[j]].view(j - i, n, 3, H, D)
            out[cu[i]:cu[j]] = flash_attn_func(x[:, :, 0], x[:, :, 1], x[:, :, 2], dropout_p, softmax_scale,
                                               causal).reshape((j - i) * n, H, D)
        i = j
    return out


# --------------------------------------------------------------- token ops
def gather_rows(src: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """dst[r] = src[index[r]] for a (rows, D) tensor; index int64 on the GPU."""
    _gpu(src, index)
    s2 = _rows2d(src)
    idx = index.to(torch.int64).contiguous()
    out = torch.empty((idx.numel(), s2.shape[1]), dtype=src.dtype, device=src.device)
    check(_capi.lib().tv_gather_rows(_p(s2), _p(idx), _p(out), idx.numel(), s2.shape[1],
                                     s2.stride(0), out.stride(0), _dt(src), _stream()),
          "tv_gather_rows")
    return out


def uniform_keep_indices(n_tokens: int, keep: int, offset: int = 0, device="cuda"):
    """torch.linspace(0, n-1, keep, dtype=long) with CPU semantics, + offset
    (modeling_nano.py:1946-1957), produced on the GPU, bit-exact."""
    out = torch.empty((keep,), dtype=torch.int64, device=device)
    check(_capi.lib().tv_uniform_keep_indices(_p(out), int(n_tokens), int(keep), int(offset),
                                              _stream()), "tv_uniform_keep_indices")
    return out


def dropped_indices(keep_sorted: torch.Tensor, start: int, n: int) -> torch.Tensor:
    """[start, start+n) minus keep_sorted, ascending (modeling_nano.py:1966-1970)."""
    _gpu(keep_sorted)
    ks = keep_sorted.to(torch.int64).contiguous()
    out = torch.empty((n - ks.numel(),), dtype=torch.int64, device=ks.device)
    check(_capi.lib().tv_dropped_indices(_p(ks), ks.numel(), int(start), int(n), _p(out),
                                         _stream()), "tv_dropped_indices")
    return out


def attn_rank_scores(q_row: torch.Tensor, k: torch.Tensor, n_keys: int, vis_start: int,
                     n_vis: int, scale: Optional[float] = None) -> torch.Tensor:
    """Importance of each vision token for pdrop "attn" (modeling_nano.py:1914-1939):
    q_row (Hq, D) = last prompt token's queries, k (L, Hkv, D); softmax over keys
    [0, n_keys) per head in fp32, mean over heads, slice [vis_start, vis_start+n_vis)."""
    _gpu(q_row, k)
    Hq, D = q_row.shape
    L, Hkv, _ = k.shape
    q_row = q_row.contiguous()
    if k.stride(-1) != 1:
        k = k.contiguous()
    scale = 1.0 / math.sqrt(D) if scale is None else float(scale)
    lib = _capi.lib()
    ws_bytes = lib.tv_attn_rank_workspace_bytes(int(n_keys), Hq)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=k.device)
    scores = torch.empty((n_vis,), dtype=torch.float32, device=k.device)
    check(lib.tv_attn_rank_scores(_p(q_row), _p(k), _p(scores), int(n_keys), Hq, Hkv, D,
                                  k.stride(0), k.stride(1), int(vis_start), int(n_vis), scale,
                                  _dt(k), _p(ws), ws_bytes, _stream()), "tv_attn_rank_scores")
    return scores


def attn_rank_logits(q_row: torch.Tensor, k: torch.Tensor, scale: Optional[float] = None) -> torch.Tensor:
    """First half of `attn_rank_scores` for a sequence shard: logits (n_keys, Hq) fp32 of this
    rank's keys k (n_keys, Hkv, D) against the broadcast query row, with the reference's roundings."""
    _gpu(q_row, k)
    Hq, D = q_row.shape
    n, Hkv, _ = k.shape
    q_row = q_row.contiguous()
    if k.stride(-1) != 1:
        k = k.contiguous()
    scale = 1.0 / math.sqrt(D) if scale is None else float(scale)
    out = torch.empty((n, Hq), dtype=torch.float32, device=k.device)
    check(_capi.lib().tv_attn_rank_logits(_p(q_row), _p(k), _p(out), int(n), Hq, Hkv, D, k.stride(0),
                                          k.stride(1), scale, _dt(k), _stream()), "tv_attn_rank_logits")
    return out


def attn_rank_scores_from_logits(logits: torch.Tensor, vis_start: int, n_vis: int, dtype) -> torch.Tensor:
    """Second half: softmax over ALL keys per head, mean over heads (roundings in `dtype`, the
    activation dtype), slice [vis_start, vis_start + n_vis).  logits (n_keys, Hq) fp32."""
    _gpu(logits)
    logits = logits.contiguous()
    n, Hq = logits.shape
    ws = torch.empty(2 * Hq, dtype=torch.float32, device=logits.device)
    scores = torch.empty((n_vis,), dtype=torch.float32, device=logits.device)
    check(_capi.lib().tv_attn_rank_scores_from_logits(_p(logits), _p(scores), int(n), Hq, int(vis_start),
                                                      int(n_vis), _DT[dtype], _p(ws), ws.numel() * 4,
                                                      _stream()), "tv_attn_rank_scores_from_logits")
    return scores


# ---------------------------------------------------------------- patch embed
def apply_rotary_pos_emb_(q, k, cos, sin):
    """In-place rotary embedding of q (B, L, Hq, D) and k (B, L, Hkv, D) (views with a
    contiguous head dim) with cos / sin (B, L, D): modeling_qwen2.py:89-113."""
    _gpu(q, k, cos, sin)
    B, L, _, D = q.shape
    cs = cos.to(q.dtype).reshape(B * L, D).contiguous()
    sn = sin.to(q.dtype).reshape(B * L, D).contiguous()
    for t in (q, k):
        assert t.stride(-1) == 1 and (B == 1 or t.stride(0) == L * t.stride(1))
        check(_capi.lib().tv_rope_fwd(_p(t), _p(cs), _p(sn), B * L, t.shape[2], D, t.stride(1),
                                      t.stride(2), _dt(t), _stream()), "tv_rope_fwd")
    return q, k


def silu_mul(gate, up):
    """silu(gate) * up over the last dim (Qwen2MLP, modeling_qwen2.py:78-80)."""
    _gpu(gate, up)
    g2, u2 = _rows2d(gate), _rows2d(up)
    y = torch.empty(g2.shape, dtype=gate.dtype, device=gate.device)
    check(_capi.lib().tv_silu_mul_fwd(_p(g2), _p(u2), _p(y), g2.shape[0], g2.shape[1], g2.stride(0),
                                      u2.stride(0), y.stride(0), _dt(gate), _stream()), "tv_silu_mul_fwd")
    return y.view(gate.shape)


def tome_merge_round(x, size, r: int, heads: int = 16):
    """One ToMe round for every frame (tome.py:14-83): x (F, T, C), size (F, T, 1) or None ->
    (x' (F, T-r, C), size' (F, T-r, 1)); rows = [kept even tokens, descending match score |
    odd tokens with their merged partners], size-weighted average."""
    _gpu(x, size)
    F_, T, Cc = x.shape
    x = x.contiguous()
    sz = None if size is None else size.reshape(F_, T).to(x.dtype).contiguous()
    xo = torch.empty((F_, T - r, Cc), dtype=x.dtype, device=x.device)
    so = torch.empty((F_, T - r, 1), dtype=x.dtype, device=x.device)
    lib = _capi.lib()
    ws_bytes = lib.tv_tome_workspace_bytes(F_, T, Cc, heads)
    ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=x.device)
    check(lib.tv_tome_merge_round(_p(x), _p(sz), _p(xo), _p(so), F_, T, Cc, heads, int(r), _dt(x),
                                  _p(ws), int(ws_bytes), _stream()), "tv_tome_merge_round")
    return xo, so


def patch_embed(pixels, weight, bias=None, pos=None, patch: Optional[int] = None):
    """pixels (F, C, H, W) -> (F, gh*gw, Dout); weight is the Conv2d weight
    (Dout, C, p, p) (or Conv3d (Dout, C, 1, p, p)); pos (gh*gw, Dout) optional."""
    _gpu(pixels, weight, bias, pos)
    F_, Cin, H, W = pixels.shape
    Dout = weight.shape[0]
    p = int(weight.shape[-1]) if patch is None else int(patch)
    pixels = pixels.contiguous()
    w = weight.reshape(Dout, -1).to(pixels.dtype).contiguous()
    assert w.shape[1] == Cin * p * p
    b = None if bias is None else bias.to(pixels.dtype).contiguous()
    ps = None if pos is None else pos.reshape(-1, Dout).to(pixels.dtype).contiguous()
    out = torch.empty((F_, (H // p) * (W // p), Dout), dtype=pixels.dtype, device=pixels.device)
    ws = torch.empty((_capi.lib().tv_patch_embed_workspace_bytes(Dout, Cin, p),), dtype=torch.uint8,
                     device=pixels.device)
    check(_capi.lib().tv_patch_embed_fwd(_p(pixels), _p(w), _p(b), _p(ps), _p(out), F_, Cin, H,
                                         W, p, Dout, _dt(pixels), _p(ws), _stream()),
          "tv_patch_embed_fwd")
    return out


def patch_embed_video(pixels, weight, bias=None, pos=None):
    """Conv3d k=s=(1,p,p) on (B, C, T, H, W) -> (B, T*gh*gw, Dout), token order
    (t, py, px) as vit_scale_clean.py:455-460."""
    _gpu(pixels, weight, bias, pos)
    B, Cin, T, H, W = pixels.shape
    Dout, p = weight.shape[0], int(weight.shape[-1])
    pixels = pixels.contiguous()
    w = weight.reshape(Dout, -1).to(pixels.dtype).contiguous()
    b = None if bias is None else bias.to(pixels.dtype).contiguous()
    ps = None if pos is None else pos.reshape(-1, Dout).to(pixels.dtype).contiguous()
    npatch = (H // p) * (W // p)
    out = torch.empty((B, T * npatch, Dout), dtype=pixels.dtype, device=pixels.device)
    ws = torch.empty((_capi.lib().tv_patch_embed_workspace_bytes(Dout, Cin, p),), dtype=torch.uint8,
                     device=pixels.device)
    check(_capi.lib().tv_patch_embed_strided_fwd(
        _p(pixels), _p(w), _p(b), _p(ps), _p(out), B * T, Cin, H, W, p, Dout, T,
        Cin * T * H * W, H * W, T * H * W, _dt(pixels), _p(ws), _stream()), "tv_patch_embed_strided_fwd")
    return out
