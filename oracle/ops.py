"""CPU oracle for the TimeViper forward hot path — TEST INFRASTRUCTURE ONLY.

Eager-PyTorch (CPU, fp32) restatements of the operators on the path, each citing
the reference file:line it follows (reference = xiaomi-research/timeviper,
`timeviper/model/llm/llm_repo/nano/modeling_nano.py` unless another file is named).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this package; the product path (`timeviper_amd/`) never does.

Pinning: every function here is checked in `tests/test_oracle_golden.py` against
fixtures under `tests/golden/` that were produced by importing and running the
reference's own Python in the build container (`oracle/make_golden.py`).  The one
exception is `rmsnorm_gated_ref`: its arithmetic lives in the un-vendored wheel
mamba_ssm==2.2.5 (`mamba_ssm/ops/triton/layernorm_gated.py`), absent from the
reference tree, so that operator is "parity unpinned" (restated from the
published algorithm; anchored on the call site :371-380).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- S3
def ssd_recurrence_ref(x, dt, A, B, C, D=None, dt_bias=None, dt_softplus=True,
                       dt_limit=(0.0, float("inf")), initial_states=None, group_map="block"):
    """Token-by-token definition of the selective scan, independent of any
    chunking (first-principles oracle; pins the head->group convention):
        S_t = exp(dt_t A_h) S_{t-1} + dt_t x_t (outer) B_t ;  y_t = S_t C_t + D_h x_t
    x (B,L,H,P) dt (B,L,H) A (H) B,C (B,L,G,N).  Returns y (B,L,H,P), final (B,H,P,N),
    total_decay (B,H).  fp64 internally."""
    x, dt, A, B, C = (t.double() for t in (x, dt, A, B, C))
    Bsz, L, H, P = x.shape
    G, N = B.shape[2], B.shape[3]
    if dt_bias is not None:
        dt = dt + dt_bias.double()
    if dt_softplus:
        dt = F.softplus(dt)
    dt = torch.clamp(dt, dt_limit[0], dt_limit[1])
    hidx = torch.arange(H)
    gidx = hidx // (H // G) if group_map == "block" else hidx % G
    Bh, Ch = B[:, :, gidx], C[:, :, gidx]                     # (B,L,H,N)
    S = torch.zeros(Bsz, H, P, N, dtype=torch.float64) if initial_states is None \
        else initial_states.double().clone()
    y = torch.empty(Bsz, L, H, P, dtype=torch.float64)
    for t in range(L):
        dA = torch.exp(dt[:, t] * A)                          # (B,H)
        S = S * dA[..., None, None] + (dt[:, t, :, None] * x[:, t])[..., None] * Bh[:, t, :, None, :]
        y[:, t] = (S * Ch[:, t, :, None, :]).sum(-1)
    if D is not None:
        y = y + D.double()[None, None, :, None] * x
    return y, S, (dt * A).sum(1)


def ssd_chunk_scan_ref(x, dt, A, B, C, chunk_size, D=None, dt_bias=None, dt_softplus=True,
                       dt_limit=(0.0, float("inf")), initial_states=None, group_map="block"):
    """Chunked SSD exactly as the reference's CPU path states it
    (NemotronHMamba2Mixer.torch_forward :775-851; helpers segment_sum :159-186,
    reshape_into_chunks :133-156), in fp32, with the 6-D broadcasts replaced by
    einsums of the same contractions.  group_map="tile" reproduces the CPU path's
    `B.repeat(1,1,H//G,1)` (:781-782, head h -> group h % G); "block" is what the
    GPU kernels and checkpoints use (h // (H/G))."""
    dtype = torch.float32
    x, dt, A, B, C = (t.to(dtype) for t in (x, dt, A, B, C))
    Bsz, L, H, P = x.shape
    G, N = B.shape[2], B.shape[3]
    if dt_bias is not None:
        dt = dt + dt_bias.to(dtype)
    if dt_softplus:
        dt = F.softplus(dt)                                                  # :776
    dt = torch.clamp(dt, dt_limit[0], dt_limit[1])                           # :777
    hidx = torch.arange(H)
    gidx = hidx // (H // G) if group_map == "block" else hidx % G            # :781-782
    Bh, Ch = B[:, :, gidx], C[:, :, gidx]
    Q = chunk_size
    pad = (Q - L % Q) % Q                                                    # :783
    D_res = None if D is None else D.to(dtype)[None, None, :, None] * x      # :785
    xd = x * dt[..., None]                                                   # :788
    a = A[None, None, :] * dt                                                # :789
    padl = lambda t: F.pad(t, (0, 0) * (t.dim() - 2) + (0, pad))
    xd, Bh, Ch = (padl(t).reshape(Bsz, -1, Q, *t.shape[2:]) for t in (xd, Bh, Ch))
    a = F.pad(a, (0, 0, 0, pad)).reshape(Bsz, -1, Q, H).permute(0, 3, 1, 2)  # (B,H,c,Q) :795
    a_cs = torch.cumsum(a, dim=-1)                                           # :796
    # 1. intra-chunk (diagonal blocks) :800-811
    seg = a_cs[..., :, None] - a_cs[..., None, :]                            # cs_l - cs_s
    mask = torch.tril(torch.ones(Q, Q, dtype=torch.bool))
    Lm = torch.exp(seg.masked_fill(~mask, -torch.inf))                       # (B,H,c,l,s)
    Gm = torch.einsum("bclhn,bcshn->bclsh", Ch, Bh)                          # :803-804
    M = Gm * Lm.permute(0, 2, 3, 4, 1)                                       # :807-808
    Y_diag = torch.einsum("bclsh,bcshp->bclhp", M, xd)                       # :811
    # 2. per-chunk states :815-817
    decay_states = torch.exp(a_cs[..., -1:] - a_cs)                          # (B,H,c,Q)
    states = torch.einsum("bclhn,bhcl,bclhp->bchpn", Bh, decay_states, xd)
    # 3. inter-chunk recurrence :821-829
    prev = torch.zeros_like(states[:, :1]) if initial_states is None \
        else initial_states.to(dtype)[:, None]
    states = torch.cat([prev, states], dim=1)
    chunk_decay = a_cs[..., -1]                                              # (B,H,c)
    nch = chunk_decay.shape[-1]
    new_states = [states[:, 0]]
    for c in range(nch):
        new_states.append(new_states[-1] * torch.exp(chunk_decay[:, :, c])[..., None, None]
                          + states[:, c + 1])
    new_states = torch.stack(new_states, dim=1)
    states_in, final = new_states[:, :-1], new_states[:, -1]
    # 4. state -> output :833-836
    Y_off = torch.einsum("bclhn,bchpn,bhcl->bclhp", Ch, states_in, torch.exp(a_cs))
    y = (Y_diag + Y_off).reshape(Bsz, -1, H, P)[:, :L]                       # :839-846
    if D_res is not None:
        y = y + D_res
    return y, final, (dt * A).sum(1)


# --------------------------------------------------------------------------- S2
def causal_conv1d_ref(x_blc, weight, bias=None, activation="silu", halo=None):
    """x (B,L,C): `act(conv1d(x.T)[..., :L].T)` with padding K-1 (:705; module :414-421).
    weight (C,K).  `halo` (B,K-1,C) replaces the zero left padding."""
    Bsz, L, Cc = x_blc.shape
    K = weight.shape[-1]
    if L == 0:
        return x_blc.float()
    xt = x_blc.transpose(1, 2).float()
    left = torch.zeros(Bsz, Cc, K - 1) if halo is None else halo.transpose(1, 2).float()
    y = F.conv1d(torch.cat([left, xt], dim=-1), weight.float().reshape(Cc, 1, K),
                 None if bias is None else bias.float(), groups=Cc)
    if activation in ("silu", "swish"):
        y = F.silu(y)
    return y.transpose(1, 2)


def causal_conv1d_update_ref(x_bc, conv_state, weight, bias=None, activation="silu"):
    """decode step (:685-695): roll state left, append x, dot with the taps."""
    conv_state = torch.cat([conv_state[..., 1:], x_bc[..., None]], dim=-1)
    y = (conv_state.float() * weight.float()[None]).sum(-1)
    if bias is not None:
        y = y + bias.float()
    if activation in ("silu", "swish"):
        y = F.silu(y)
    return y, conv_state


# ---------------------------------------------------------------------- S4 / L1
def rmsnorm_gated_ref(x, weight, z=None, eps=1e-5, group_size=None):
    """mamba_ssm rmsnorm_fn(..., norm_before_gate=False) as called at :371-380.
    PARITY UNPINNED (third-party mamba_ssm==2.2.5, not in the reference tree):
    u = x*silu(z); per group: u * rsqrt(mean(u^2)+eps) * w, fp32.  The only in-tree statement of
    the semantics is the docstring of visualize/nano/my_ssd_combined.py:1975, :2032 ("If False, we
    do RMSNorm(x * F.silu(z))") and the group_size = dim // ngroups convention of its call :1678-1688."""
    u = x.float()
    if z is not None:
        u = u * F.silu(z.float())
    D = u.shape[-1]
    gs = D if group_size is None else group_size
    ug = u.reshape(*u.shape[:-1], D // gs, gs)
    ug = ug * torch.rsqrt(ug.pow(2).mean(-1, keepdim=True) + eps)
    return ug.reshape(u.shape) * weight.float()


def rmsnorm_ref(x, weight, eps):
    """NemotronHRMSNorm.forward :897-903 (fp32 statistics, fp32 weight)."""
    h = x.float()
    var = h.pow(2).mean(-1, keepdim=True)
    return weight.float() * (h * torch.rsqrt(var + eps))


# --------------------------------------------------------------------- A1 / T3
def attention_ref(q, k, v, causal, scale=None):
    """q (B,Lq,Hq,D), k/v (B,Lk,Hkv,D): repeat_kv (:998-1009) + softmax(QK^T/sqrt(d))V
    (:1300-1307), causal mask bottom-right aligned, fp32.  Returns (o, lse)."""
    B, Lq, Hq, D = q.shape
    Lk, Hkv = k.shape[1], k.shape[2]
    rep = Hq // Hkv
    qf = q.float().permute(0, 2, 1, 3)
    kf = k.float().permute(0, 2, 1, 3).repeat_interleave(rep, dim=1)
    vf = v.float().permute(0, 2, 1, 3).repeat_interleave(rep, dim=1)
    scale = 1.0 / math.sqrt(D) if scale is None else scale
    s = qf @ kf.transpose(-1, -2) * scale
    if causal:
        i = torch.arange(Lq)[:, None]
        j = torch.arange(Lk)[None, :]
        s = s.masked_fill(j > i + (Lk - Lq), -torch.inf)
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None])
    p = torch.nan_to_num(p, nan=0.0)
    return (p @ vf).permute(0, 2, 1, 3), lse


def fp8_quantise_ref(x, qmax: float = 440.0):
    """What the fp8 attention variant (BASELINE config 5; tv_flash_attn_fp8_fwd) does to q, k and
    v before the MFMAs: one scale per (batch, head) so that max |x| -> qmax, rounding to OCP
    e4m3 (torch.float8_e4m3fn, round to nearest even), and the matching dequantisation.
    x (B, L, H, D) -> fp32 tensor of the values the kernel multiplies.  The reference has no fp8
    path (its attention is bf16, modeling_qwen2.py:196-244): this restates the QUANTISER of the
    opt-in variant so that its MFMA / softmax arithmetic can be checked separately from the
    quantisation error."""
    xf = x.float()
    amax = xf.abs().amax(dim=(1, 3), keepdim=True)
    qs = torch.where(amax > 0, qmax / amax, torch.ones_like(amax))
    q8 = (xf * qs).to(torch.float8_e4m3fn).float()
    return q8 / qs


# --------------------------------------------------------------------------- T1
def uniform_keep_indices_ref(n_tokens: int, keep: int) -> torch.Tensor:
    """`torch.linspace(0, image_tokens-1, keep_length, dtype=torch.long)` (:1946-1953)
    restated from ATen's CPU kernel: double step, first half counts up from start,
    second half down from end, truncation toward zero."""
    if keep <= 0:
        return torch.empty(0, dtype=torch.long)
    if keep == 1:
        return torch.zeros(1, dtype=torch.long)
    import numpy as np
    start, end = 0.0, float(n_tokens - 1)
    step = (end - start) / (keep - 1)
    i = np.arange(keep, dtype=np.int64)
    half = keep // 2
    lo = start + step * i.astype(np.float64)
    hi = end - step * (keep - i - 1).astype(np.float64)
    return torch.from_numpy(np.trunc(np.where(i < half, lo, hi)).astype(np.int64))


def attn_rank_scores_ref(hidden, q_w, k_w, num_heads, num_kv_heads, head_dim, query_row,
                         vis_start, n_vis, q_b=None, k_b=None):
    """pdrop "attn" importance (:1822-1857, :1914-1939) for batch 1 / eval:
    q,k = q_proj/k_proj of the UN-NORMED hidden states; one query row; causal row of
    the mask; softmax in fp32 cast back to the activation dtype; mean over heads;
    vision span.  hidden (L, Dm).  Never builds the (L,L) mask (only row
    `query_row` of it is used by the reference)."""
    dt = hidden.dtype
    q = F.linear(hidden[query_row:query_row + 1], q_w, q_b).view(1, num_heads, head_dim).transpose(0, 1)
    k = F.linear(hidden, k_w, k_b).view(-1, num_kv_heads, head_dim).transpose(0, 1)
    k = k.repeat_interleave(num_heads // num_kv_heads, dim=0)           # repeat_kv :1845
    w = torch.matmul(q, k.transpose(1, 2)) / math.sqrt(head_dim)         # (H,1,L) :1923-1927
    mask_row = torch.zeros(hidden.shape[0], dtype=dt)
    mask_row[query_row + 1:] = float("-inf")                             # :1848-1857 row
    w = w + mask_row
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(dt)                 # :1929-1933
    avg = torch.mean(w, dim=0)[:, vis_start:vis_start + n_vis]           # :1935-1938
    return torch.mean(avg, dim=0)                                        # :1939


def topk_keep_ref(scores: torch.Tensor, keep: int) -> torch.Tensor:
    """`scores.topk(keep).indices` (:1942) with a DEFINED tie-break (lower index
    first); the reference's tie order is device dependent."""
    order = torch.sort(scores.float(), descending=True, stable=True).indices
    return order[:keep]


# ----------------------------------------------------------------------- V1/V2
def patch_embed_ref(pixels, weight, bias=None, pos=None):
    """Conv2d(k=s=p) + flatten(2).transpose(1,2) (+pos): timm PatchEmbed as used by
    TimmViTBackbone (timeviper/model/vit/base_vision.py:146-170)."""
    y = F.conv2d(pixels.float(), weight.float(), None if bias is None else bias.float(),
                 stride=weight.shape[-1])
    y = y.flatten(2).transpose(1, 2)
    if pos is not None:
        y = y + pos.float().reshape(1, -1, y.shape[-1])
    return y


def patch_embed_video_ref(pixels, weight, bias=None):
    """InternVideo2 PatchEmbed (timeviper/model/vit/internvideo2/vit_scale_clean.py:445-461):
    Conv3d k=s=(1,p,p) on (B,C,T,H,W) -> flatten(3).permute(0,2,3,1) -> (B, T*L, D)."""
    p = weight.shape[-1]
    y = F.conv3d(pixels.float(), weight.float(), None if bias is None else bias.float(),
                 stride=(1, p, p))
    y = y.flatten(3).permute(0, 2, 3, 1)
    return y.reshape(y.shape[0], -1, y.shape[-1])
