"""SigLIP / DINOv2-style ViT image backbone with timm-compatible parameter names.

The reference wraps `timm.create_model(...)` (third-party, not vendored; contract in
timeviper/model/vit/base_vision.py:126-278): forward = `get_intermediate_layers(
n={depth-2})` — patch-embed, + learned pos-embed, blocks 0..depth-2, no final norm,
prefix tokens stripped — returning (B, num_patches, embed_dim).  This module
re-states the public timm VisionTransformer forward for that path with:
  * the patch embedding as the im2col-free MFMA GEMM (`kernels.patch_embed`, bias and
    pos-embed fused in the epilogue),
  * attention on the flash kernel (head_dim 72 for so400m, non-causal),
  * Linear layers on torch (hipBLASLt), LayerNorm / GELU on torch.
Parameter names follow timm (`patch_embed.proj`, `pos_embed`, `blocks.{i}.{norm1,attn.
qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}`, `norm`, `attn_pool.*`) so a timm checkpoint
loads with strict=True; the pooling head is kept for that reason only and never run.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import kernels as K

TIMM_VIT_CONFIGS = {
    # public timm model definitions (not in the reference tree)
    "vit_so400m_patch14_siglip_384": dict(img_size=384, patch_size=14, embed_dim=1152, depth=27,
                                          num_heads=16, mlp_hidden=4304, class_token=False,
                                          reg_tokens=0, pool="map"),
    "vit_so400m_patch14_siglip_224": dict(img_size=224, patch_size=14, embed_dim=1152, depth=27,
                                          num_heads=16, mlp_hidden=4304, class_token=False,
                                          reg_tokens=0, pool="map"),
    "vit_base_patch16_siglip_224": dict(img_size=224, patch_size=16, embed_dim=768, depth=12,
                                        num_heads=12, mlp_hidden=3072, class_token=False,
                                        reg_tokens=0, pool="map"),
    "vit_base_patch16_siglip_256": dict(img_size=256, patch_size=16, embed_dim=768, depth=12,
                                        num_heads=12, mlp_hidden=3072, class_token=False,
                                        reg_tokens=0, pool="map"),
    "vit_base_patch16_siglip_384": dict(img_size=384, patch_size=16, embed_dim=768, depth=12,
                                        num_heads=12, mlp_hidden=3072, class_token=False,
                                        reg_tokens=0, pool="map"),
    "vit_large_patch14_reg4_dinov2.lvd142m": dict(img_size=224, patch_size=14, embed_dim=1024,
                                                  depth=24, num_heads=16, mlp_hidden=4096,
                                                  class_token=True, reg_tokens=4, pool="token",
                                                  layer_scale=1e-5, no_embed_class=True),
}


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x, pos: Optional[torch.Tensor] = None):
        return K.patch_embed(x, self.proj.weight, self.proj.bias, pos)


GEMM_ALIGN = 256     # hipBLASLt's bf16 macro-tile on gfx950


def _aligned(n: int) -> int:
    return -(-n // GEMM_ALIGN) * GEMM_ALIGN


class ZeroPaddedLinears:
    """Inference-time copies of Linear weights, zero-padded so that the GEMM's N (and the next
    GEMM's K) is a multiple of the library's 256-wide macro-tile: so400m's 4304 and 3456 are not,
    and hipBLASLt runs those shapes 6-9 % slower than the padded ones (devtools/bench_gemm_pad.py).
    The padding is exact: padded outputs are bias 0 + 0, GELU(0) = 0, and padded K columns meet zero
    weights.  Parameters keep their checkpoint shapes; the copies are rebuilt whenever a parameter
    is replaced or written in place."""

    def __init__(self):
        self._key, self._val = None, None

    @staticmethod
    def wanted(x: torch.Tensor, n: int) -> bool:
        return x.is_cuda and not torch.is_grad_enabled() and n >= 1024 and n % GEMM_ALIGN != 0

    def get(self, params, build):
        key = K.param_key(params)
        if key != self._key:
            self._key, self._val = key, build()
        return self._val


def _pad_rows(w: torch.Tensor, n: int) -> torch.Tensor:
    out = w.new_zeros((n,) + tuple(w.shape[1:]))
    out[: w.shape[0]] = w
    out._tv_useful_rows = w.shape[0]        # (bench.py counts useful flops)
    if w.dim() == 2:
        K.PADDED_USEFUL[out.data_ptr()] = (w.shape[0], w.shape[1])
    return out


def _pad_cols(w: torch.Tensor, k: int) -> torch.Tensor:
    out = w.new_zeros((w.shape[0], k))
    out[:, : w.shape[1]] = w
    out._tv_useful_cols = w.shape[1]
    K.PADDED_USEFUL[out.data_ptr()] = (w.shape[0], w.shape[1])
    return out


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)
        self._padded = ZeroPaddedLinears()

    def heads(self, x):
        """qkv projection + attention, without the output projection"""
        B, N, C = x.shape
        if self._own_qkv(x):
            # the persistent GEMM (csrc/gemm_persist.hip) on the weights as stored: 3 456 columns are 13 tiles and one
            # shifted back — no padded copies, and the bias is added in the kernel's epilogue
            qkv = K.linear_fused(x, self.qkv.weight, self.qkv.bias, epilogue=K.GEMM_BIAS)
            qkv = qkv.view(B, N, 3, self.num_heads, self.head_dim)
        elif ZeroPaddedLinears.wanted(x, 3 * C):
            w, b = self._padded.get((self.qkv.weight, self.qkv.bias), lambda: (
                _pad_rows(self.qkv.weight.detach(), _aligned(3 * C)),
                _pad_rows(self.qkv.bias.detach(), _aligned(3 * C))))
            qkv = F.linear(x, w, b)[..., : 3 * C].unflatten(-1, (3, self.num_heads, self.head_dim))
        else:
            qkv = self.qkv(x).view(B, N, 3, self.num_heads, self.head_dim)
        o = K.flash_attn_func(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], softmax_scale=self.scale,
                              causal=False)
        return o.reshape(B, N, C)

    def _own_qkv(self, x) -> bool:
        """qkv projection + bias on the hand-written persistent GEMM instead of hipBLASLt — opt-in (TV_VIT_OWN_QKV=1).
        Stand-alone on 2 048 SigLIP frames it wins (9.5 ms against 10.2 ms for the library on the weights as stored and
        9.7 ms on the copies zero-padded to 3 584 columns, same box); inside the 10 240-frame forward the library on the
        padded copies is 0.5 % of the step ahead (1 225.6 against 1 219.9 frames/s, same box, round 5), so that stays
        the default.  bf16 on the GPU, inference, K a multiple of 128, N of 8, enough rows for every compute unit to
        walk a few tiles."""
        w = self.qkv.weight
        return (os.environ.get("TV_VIT_OWN_QKV", "0") == "1" and x.is_cuda and x.dtype == torch.bfloat16
                and w.dtype == torch.bfloat16 and self.qkv.bias is not None and not torch.is_grad_enabled()
                and w.shape[1] % 128 == 0 and w.shape[0] % 8 == 0 and x.numel() // x.shape[-1] >= 65536)

    def forward(self, x):
        return self.proj(self.heads(x))


class Mlp(nn.Module):
    def __init__(self, dim, hidden, act="gelu"):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU(approximate="tanh" if act == "gelu_tanh" else "none")
        self.exact_gelu = act != "gelu_tanh"
        self.fc2 = nn.Linear(hidden, dim)
        self._padded = ZeroPaddedLinears()

    def hidden(self, x):
        """act(fc1(x)) and the fc2 weight that goes with it (zero-padded to the GEMM tile when that pays)"""
        Hd = self.fc1.out_features
        if ZeroPaddedLinears.wanted(x, Hd):
            w1, b1, w2 = self._padded.get((self.fc1.weight, self.fc1.bias, self.fc2.weight), lambda: (
                _pad_rows(self.fc1.weight.detach(), _aligned(Hd)), _pad_rows(self.fc1.bias.detach(), _aligned(Hd)),
                _pad_cols(self.fc2.weight.detach(), _aligned(Hd))))
            if self._fused_fc1(x, w1):
                return K.linear_fused(x, w1, b1, epilogue=K.GEMM_BIAS_GELU), w2
            h = F.linear(x, w1, b1)
        else:
            w2 = self.fc2.weight
            if self._fused_fc1(x, self.fc1.weight):
                return K.linear_fused(x, self.fc1.weight, self.fc1.bias, epilogue=K.GEMM_BIAS_GELU), w2
            h = self.fc1(x)
        h = K.gelu(h, inplace=True) if self.exact_gelu else self.act(h)
        return h, w2

    def _fused_fc1(self, x, w1) -> bool:
        """fc1 + bias + exact GELU in ONE kernel (csrc/gemm_persist.hip from 4 tiles per compute unit on, csrc/gemm.hip
        below: the activation is applied to the accumulators, same rounding points as GEMM-then-GELU): 14.0 ms
        (persistent) / 14.8 ms (per tile) against 16.7 ms for hipBLASLt + tv_gelu_fwd per 2 048 SigLIP frames.  bf16 on the GPU, inference, K a multiple of 128, enough rows to fill the chip; TV_VIT_FUSED_FC1=0
        switches it off."""
        return (self.exact_gelu and x.is_cuda and x.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16
                and not torch.is_grad_enabled() and w1.shape[1] % 128 == 0 and w1.shape[0] % 4 == 0
                and x.numel() // x.shape[-1] >= 4096 and os.environ.get("TV_VIT_FUSED_FC1", "1") != "0")

    def forward(self, x):
        h, w2 = self.hidden(x)
        return F.linear(h, w2, self.fc2.bias)


def _accumulate(x2: torch.Tensor, h2: torch.Tensor, w: torch.Tensor, role: str) -> None:
    """x2 += h2 @ w.T (one rounding of the sum to bf16), the residual stream's output projections.  Default: hipBLASLt
    (`torch.addmm`, beta = 1).  The hand-written GEMM with the accumulating epilogue (csrc/gemm_drip.hip: 256 x 192 tiles —
    1 152 columns are 6 of them, 4.5 of the library's 256-wide ones) takes the roles named in TV_VIT_OWN_ACCUM (comma
    separated: proj, fc2) when the shape allows: bf16, K a multiple of 128 and >= 1 152, enough rows for every compute
    unit to walk a few tiles."""
    own = _OWN_ACCUM
    if (role in own and x2.is_cuda and x2.dtype == torch.bfloat16 and h2.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
            and not torch.is_grad_enabled() and w.shape[1] % 128 == 0 and w.shape[1] >= 1152 and w.shape[0] % 8 == 0
            and x2.shape[0] >= 65536 and h2.stride(1) == 1 and w.stride(1) == 1):
        K.linear_fused(h2, w, None, epilogue=K.GEMM_ACCUM, out=x2)
    else:
        torch.addmm(x2, h2, w.t(), out=x2)


_OWN_ACCUM = tuple(r for r in os.environ.get("TV_VIT_OWN_ACCUM", "").split(",") if r)


class LayerScale(nn.Module):
    def __init__(self, dim, init):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_hidden, act="gelu", layer_scale=None, eps=1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads)
        self.ls1 = LayerScale(dim, layer_scale) if layer_scale else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, mlp_hidden, act)
        self.ls2 = LayerScale(dim, layer_scale) if layer_scale else nn.Identity()

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))

    def forward_fused(self, x, delta):
        """Same block on the (stream, pending sub-layer output) pair: each residual add is
        fused into the LayerNorm that follows it.  Returns (x', delta') with the block's
        output = x' + delta'."""
        n1, n2 = self.norm1, self.norm2
        if delta is None:
            h = K.layer_norm(x, n1.weight, n1.bias, n1.eps)
        else:
            h, x = K.layer_norm(x, n1.weight, n1.bias, n1.eps, residual=delta, return_sum=True)
        a = self.ls1(self.attn(h))
        h, x = K.layer_norm(x, n2.weight, n2.bias, n2.eps, residual=a, return_sum=True)
        return x, self.ls2(self.mlp(h))

    def forward_stream(self, x, pend, pend_mid=None, pend_out=None):
        """Same block (no LayerScale) on a residual stream the projections accumulate INTO: the output
        GEMMs of both sub-layers run as x += h W^T (one pass over x inside the GEMM's epilogue instead
        of a separate read-add-write), their biases are carried beside the stream in `pend` (fp32, one
        row) and enter through the LayerNorm kernel, which then reads x once and writes the normalised
        rows once — 2 passes over the stream per LayerNorm instead of 4.  x (B, N, C), contiguous, is
        updated in place; the block's output is x + pend.  Arithmetic: x + h W^T is rounded to bf16
        once (the two-step form rounds h W^T + b and the sum separately).  `pend_mid` / `pend_out`: the
        bias rows after the attention / after the block where the caller has them pre-summed (they are a pure
        function of the parameters: VisionTransformer._pending_rows)."""
        n1, n2 = self.norm1, self.norm2
        x2 = x.view(-1, x.shape[-1])
        h = K.layer_norm(x, n1.weight, n1.bias, n1.eps, row_bias=pend)
        o = self.attn.heads(h)
        _accumulate(x2, o.view(x2.shape), self.attn.proj.weight, "proj")
        if pend_mid is None:          # (callers that do not carry the pre-summed rows)
            pend_mid = self.attn.proj.bias.float() if pend is None else pend + self.attn.proj.bias.float()
        h = K.layer_norm(x, n2.weight, n2.bias, n2.eps, row_bias=pend_mid)
        hid, w2 = self.mlp.hidden(h)
        _accumulate(x2, hid.view(x2.shape[0], -1), w2, "fc2")
        return x, (pend_mid + self.mlp.fc2.bias.float() if pend_out is None else pend_out)


class AttentionPoolLatent(nn.Module):
    """SigLIP 'map' head — parameters only (state-dict compatibility); the reference
    never runs it (num_classes=0 and forward is get_intermediate_layers)."""

    def __init__(self, dim, num_heads, mlp_hidden):
        super().__init__()
        self.latent = nn.Parameter(torch.zeros(1, 1, dim))
        self.q = nn.Linear(dim, dim)
        self.kv = nn.Linear(dim, dim * 2)
        self.proj = nn.Linear(dim, dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, mlp_hidden)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=384, patch_size=14, embed_dim=1152, depth=27, num_heads=16,
                 mlp_hidden=4304, class_token=False, reg_tokens=0, pool="map", act="gelu",
                 layer_scale=None, no_embed_class=False, in_chans=3):
        super().__init__()
        self.embed_dim, self.depth = embed_dim, depth
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.num_prefix_tokens = (1 if class_token else 0) + reg_tokens
        self.no_embed_class = no_embed_class
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if class_token else None
        self.reg_token = nn.Parameter(torch.zeros(1, reg_tokens, embed_dim)) if reg_tokens else None
        n_pos = self.patch_embed.num_patches + (0 if no_embed_class else self.num_prefix_tokens)
        self.pos_embed = nn.Parameter(torch.randn(1, n_pos, embed_dim) * 0.02)
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_hidden, act, layer_scale)
                                      for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.attn_pool = AttentionPoolLatent(embed_dim, num_heads, mlp_hidden) if pool == "map" else None

    def _pending_rows(self, blocks):
        """Per block: the fp32 bias row carried beside the stream after the attention projection and after
        the MLP (running sums of proj.bias and fc2.bias in block order, the same left-to-right fp32 adds
        `Block.forward_stream` would make) — built once per parameter version instead of two adds and two
        casts per block and call."""
        ps = [p for b in blocks for p in (b.attn.proj.bias, b.mlp.fc2.bias)]
        key = K.param_key(ps)
        if key != getattr(self, "_pend_key", None):
            rows, pend = [], None
            with torch.no_grad():
                for b in blocks:
                    pb = b.attn.proj.bias.detach().float()
                    mid = pb if pend is None else pend + pb
                    pend = mid + b.mlp.fc2.bias.detach().float()
                    rows.append((mid, pend))
            self._pend_key, self._pend_rows = key, rows
        return self._pend_rows

    def get_intermediate_layers(self, x, n=None):
        """timm semantics for n={k}: run blocks 0..k, return block k's output with
        the prefix tokens removed, no final norm."""
        last = max(n) if n is not None else self.depth - 1
        if self.num_prefix_tokens == 0:
            x = self.patch_embed(x, self.pos_embed[0])          # bias + pos fused in the GEMM epilogue
        else:
            npatch = self.patch_embed.num_patches
            if self.no_embed_class:
                x = self.patch_embed(x, self.pos_embed[0])
                prefix = [t for t in (self.cls_token, self.reg_token) if t is not None]
                x = torch.cat([t.expand(x.shape[0], -1, -1) for t in prefix] + [x], dim=1)
            else:
                x = self.patch_embed(x, None)
                prefix = [t for t in (self.cls_token, self.reg_token) if t is not None]
                x = torch.cat([t.expand(x.shape[0], -1, -1) for t in prefix] + [x], dim=1)
                x = x + self.pos_embed
        blocks = [self.blocks[i] for i in range(last + 1)]
        if x.is_cuda and all(isinstance(b.ls1, nn.Identity) and b.mlp.exact_gelu for b in blocks) \
                and os.environ.get("TV_VIT_STREAM", "1") != "0":
            # output projections accumulate into the stream, biases ride beside it (Block.forward_stream)
            x = x if x.is_contiguous() else x.contiguous()
            rows = self._pending_rows(blocks)
            pend = None
            for blk, (mid, out) in zip(blocks, rows):
                x, pend = blk.forward_stream(x, pend, mid, out)
            x = x + pend.to(x.dtype)
            return (x[:, self.num_prefix_tokens:],)
        delta = None
        for blk in blocks:
            x, delta = blk.forward_fused(x, delta)
        x = x + delta
        return (x[:, self.num_prefix_tokens:],)

    def forward(self, x):
        return self.get_intermediate_layers(x, n={self.depth - 2})[0]
