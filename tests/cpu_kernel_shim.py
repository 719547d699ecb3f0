"""TEST INFRASTRUCTURE: oracle-backed stand-ins for `timeviper_amd.kernels`, so that the
host-side logic that sits ABOVE the C ABI (model mirror, sequence-parallel runner) can be
exercised on a CPU-only box (gloo, world_size 2).  Never imported by the product."""
import contextlib
import math

import torch

from oracle import ops as R
from oracle import vit as RV


def _conv_xbc(xBC, weight, bias, d_inner, ngroups, dstate, activation="silu", halo=None, return_cb=False):
    y = R.causal_conv1d_ref(xBC.float(), weight.float().reshape(xBC.shape[-1], -1),
                            None if bias is None else bias.float(), activation,
                            None if halo is None else halo.float()).to(xBC.dtype)
    B, L, _ = xBC.shape
    x, Bm, Cm = y.split([d_inner, ngroups * dstate, ngroups * dstate], dim=-1)
    out = (x.contiguous(), Bm.reshape(B, L, ngroups, dstate), Cm.reshape(B, L, ngroups, dstate))
    return out + (None,) if return_cb else out       # the C.B^T fragments are a device-side detail


def _conv_fn(x, weight, bias=None, activation=None, halo=None, **kw):
    y = R.causal_conv1d_ref(x.transpose(1, 2).float(), weight.float().reshape(x.shape[1], -1),
                            None if bias is None else bias.float(), activation,
                            None if halo is None else halo.float())
    return y.to(x.dtype).transpose(1, 2)


def _scan(x, dt, A, B, C, chunk_size=None, D=None, z=None, dt_bias=None, initial_states=None,
          dt_softplus=False, dt_limit=(0.0, float("inf")), return_final_states=False,
          group_map="block", return_total_decay=False, **kw):
    y, fin, dec = R.ssd_recurrence_ref(x, dt, A, B, C, D=D, dt_bias=dt_bias, dt_softplus=dt_softplus,
                                       dt_limit=dt_limit, initial_states=initial_states,
                                       group_map=group_map)
    out = (y.to(x.dtype),)
    if return_final_states:
        out += (fin.float(),)
    if return_total_decay:
        out += (dec.float(),)
    return out[0] if len(out) == 1 else out


def _state_corr(y, dt, A, C, state_in, dt_bias=None, dt_softplus=False, dt_limit=(0.0, float("inf")),
                group_map="block"):
    """y_t += exp(cs_t) C_t . S_in, from the definition"""
    Bsz, L, H, P = y.shape
    G = C.shape[2]
    d = dt.float() + (0.0 if dt_bias is None else dt_bias.float())
    if dt_softplus:
        d = torch.nn.functional.softplus(d)
    d = d.clamp(dt_limit[0], dt_limit[1])
    cs = torch.cumsum(d * A.float(), dim=1)                                   # (B, L, H)
    gidx = (torch.arange(H) % G) if group_map == "tile" else (torch.arange(H) // (H // G))
    Ch = C.float()[:, :, gidx]                                                # (B, L, H, N)
    corr = torch.einsum("blhn,bhpn->blhp", Ch, state_in.float()) * torch.exp(cs)[..., None]
    y.copy_((y.float() + corr).to(y.dtype))
    return y


def _rms(x, weight, eps, residual=None, return_sum=False):
    s = x if residual is None else (x + residual)
    y = R.rmsnorm_ref(s, weight, eps).to(x.dtype)
    return (y, s) if return_sum else y


def _ln(x, weight, bias, eps, residual=None, return_sum=False, row_bias=None):
    s = x if residual is None else (x + residual)
    sf = s.float() if row_bias is None else s.float() + row_bias.float()
    y = torch.nn.functional.layer_norm(sf, (s.shape[-1],), weight.float(),
                                       None if bias is None else bias.float(), eps).to(x.dtype)
    return (y, s) if return_sum else y


def _gated(x, weight, bias=None, z=None, eps=1e-6, group_size=None, norm_before_gate=True, **kw):
    return R.rmsnorm_gated_ref(x, weight, z, eps, group_size).to(x.dtype)


def _fa(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, return_lse=False):
    o, lse = R.attention_ref(q, k, v, causal, softmax_scale)
    return (o.to(q.dtype), lse) if return_lse else o.to(q.dtype)


def _fa_fwd(q, k, v, attention_mask=None, query_length=None, is_causal=True, **kw):
    return _fa(q, k, v, causal=is_causal)


def _rank_logits(q_row, k, scale=None):
    """(keys, heads) logits with the reference's roundings (modeling_nano.py:1923-1927)"""
    H, D = q_row.shape
    rep = H // k.shape[1]
    dt = k.dtype
    logit = torch.einsum("hd,khd->kh", q_row.float(), k.float().repeat_interleave(rep, 1)).to(dt)
    return (logit.float() / math.sqrt(D)).to(dt).float()


def _rank_from_logits(logits, vis_start, n_vis, dtype):
    p = torch.softmax(logits, dim=0).to(dtype).float()
    return p.mean(1).to(dtype).float()[vis_start:vis_start + n_vis]


def _rank(q_row, k, n_keys, vis_start, n_vis, scale=None):
    return _rank_from_logits(_rank_logits(q_row, k[:n_keys], scale), vis_start, n_vis, k.dtype)


def _patch_video(pix, w, b=None, pos=None):
    y = torch.nn.functional.conv3d(pix.float(), w.float(), None if b is None else b.float(),
                                   stride=w.shape[2:])
    y = y.flatten(3).permute(0, 2, 3, 1).flatten(1, 2)
    if pos is not None:
        y = y + pos.float().reshape(-1, y.shape[-1])
    return y.to(pix.dtype)


def _rope_(q, k, cos, sin):
    def rot(x):
        h = x.shape[-1] // 2
        return torch.cat((-x[..., h:], x[..., :h]), dim=-1)
    c, s_ = cos[:, :, None, :].to(q.dtype), sin[:, :, None, :].to(q.dtype)
    q.copy_(q * c + rot(q) * s_)
    k.copy_(k * c + rot(k) * s_)
    return q, k


def _dropped(keep_sorted, start, n):
    allidx = torch.arange(start, start + n)
    return allidx[~torch.isin(allidx, keep_sorted)]


def _conv_update(x, conv_state, weight, bias=None, activation=None):
    y, new = R.causal_conv1d_update_ref(x, conv_state, weight.reshape(x.shape[1], -1), bias, activation)
    conv_state.copy_(new.to(conv_state.dtype))
    return y.to(x.dtype)


def _state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """one token of the recurrence (modeling_nano.py:528-539 with per-head A / dt / D)"""
    Bsz, H, P = x.shape
    G = B.shape[1]
    head = lambda t, nd: None if t is None else (t if t.dim() == nd else t[(..., *([0] * (t.dim() - nd)))]).float()
    A1, D1, b1 = head(A, 1), head(D, 1), head(dt_bias, 1)
    d = (dt if dt.dim() == 2 else dt[..., 0]).float()
    if b1 is not None:
        d = d + b1
    if dt_softplus:
        d = torch.nn.functional.softplus(d)
    Bh = B.float().repeat_interleave(H // G, dim=1)
    Ch = C.float().repeat_interleave(H // G, dim=1)
    state.mul_(torch.exp(d * A1)[..., None, None]).add_((d[..., None] * x.float())[..., None] * Bh[:, :, None, :])
    y = torch.einsum("bhpn,bhn->bhp", state, Ch)
    if D1 is not None:
        y = y + D1[None, :, None] * x.float()
    return y.to(x.dtype)


@contextlib.contextmanager
def cpu_kernels():
    from timeviper_amd import kernels as K
    patches = {
        "causal_conv1d_xbc": _conv_xbc, "causal_conv1d_fn": _conv_fn,
        "mamba_chunk_scan_combined": _scan, "rms_norm": _rms, "rmsnorm_fn": _gated,
        "layer_norm": _ln, "gelu": lambda x, inplace=False: torch.nn.functional.gelu(x),
        "flash_attn_func": _fa, "_flash_attention_forward": _fa_fwd,
        "gather_rows": lambda src, idx: src.reshape(-1, src.shape[-1])[idx],
        "uniform_keep_indices": lambda n, keep, offset=0, device="cpu":
            R.uniform_keep_indices_ref(n, keep) + offset,
        "dropped_indices": _dropped, "attn_rank_scores": _rank,
        "attn_rank_logits": _rank_logits, "attn_rank_scores_from_logits": _rank_from_logits,
        "patch_embed": lambda pix, w, b=None, pos=None, patch=None:
            R.patch_embed_ref(pix, w, b, pos).to(pix.dtype),
        "patch_embed_video": _patch_video,
        "apply_rotary_pos_emb_": _rope_,
        "silu_mul": lambda g, u: torch.nn.functional.silu(g) * u,
        "tome_merge_round": RV.tome_merge_round_ref,
        "relu2": lambda x, inplace=False: torch.square(torch.relu(x)),
        "causal_conv1d_update": _conv_update, "selective_state_update": _state_update,
        "ssd_state_correction": _state_corr,
    }
    saved = {k: getattr(K, k) for k in patches}
    for k, v in patches.items():
        setattr(K, k, v)
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(K, k, v)
