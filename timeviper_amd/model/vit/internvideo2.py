"""InternVideo2-1B video ViT tower (reference timeviper/model/vit/internvideo2/).

Mirror of the eval forward of `PretrainVisionTransformer_clean`
(vit_scale_clean.py:464-729) as configured by `InternVideo2VisionTower._build_model`
(model.py:143-171): tubelet-1 Conv3d patch embed, cls token, learned (sincos-initialised)
video / image position tables, `depth + x_vis_return_idx + 1` pre-norm blocks with
RMSNorm(eps 1e-6), bias-free qkv, q/k RMS-normalised over the full embedding dim,
fp32 LayerScale, GELU MLP of width int(dim*48/11); `x_vis_only` so no pooling head.

Device work goes to the HIP operators: `kernels.patch_embed_video` (Conv3d as the
im2col-free MFMA GEMM), `kernels.rms_norm` (block norms with the residual add fused,
and the q/k norms on strided views of the qkv projection), `kernels.flash_attn_func`
(head_dim 88, non-causal), `kernels.gelu`.  Linear layers stay on hipBLASLt.

Parameter names equal the reference's (`patch_embed.proj`, `cls_token`, `pos_embed`,
`img_pos_embed`, `blocks.{i}.{norm1,attn.{qkv,proj,q_norm,k_norm},ls1,norm2,
mlp.{fc1,fc2},ls2}`), so `InternVideo2-1B_f4_vision.pt` loads the way
backbone.py:63-91 loads it.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import kernels as K
from .siglip import ZeroPaddedLinears, _aligned, _pad_rows
from . import VisionBackbone


# ------------------------------------------------------------------ position tables
def _sincos_1d(dim: int, pos: np.ndarray) -> np.ndarray:
    """(M,) positions -> (M, dim): [sin | cos] halves with 10000^(-2i/dim) frequencies
    (pos_embed.py:104-122; frequencies in fp32 as there)."""
    freq = np.arange(dim // 2, dtype=np.float32)
    freq /= dim / 2.0
    freq = 1.0 / 10000 ** freq
    ang = np.outer(pos.reshape(-1), freq)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def sincos_pos_embed_3d(dim: int, grid: int, t_size: int, cls_token: bool = True) -> np.ndarray:
    """MAE-ST table used by `init_pos_embed` (vit_scale_clean.py:604-626, pos_embed.py:14-53):
    the first dim/4 channels encode the frame index, the remaining 3*dim/4 the (x, y)
    patch position; rows ordered (t, y, x), an all-zero row first for the cls token."""
    assert dim % 4 == 0
    d_sp, d_t = dim // 4 * 3, dim // 4
    ys, xs = np.meshgrid(np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32),
                         indexing="ij")
    spatial = np.concatenate([_sincos_1d(d_sp // 2, xs), _sincos_1d(d_sp // 2, ys)], axis=1)
    temporal = _sincos_1d(d_t, np.arange(t_size, dtype=np.float32))
    table = np.concatenate([np.repeat(temporal[:, None, :], grid * grid, axis=1),
                            np.repeat(spatial[None], t_size, axis=0)], axis=-1).reshape(-1, dim)
    if cls_token:
        table = np.concatenate([np.zeros((1, dim)), table], axis=0)
    return table


# ------------------------------------------------------------------ layers
class RMSNorm(nn.Module):
    """vit_scale_clean.py:152-163."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x, residual=None, return_sum=False):
        return K.rms_norm(x, self.weight, self.variance_epsilon, residual=residual,
                          return_sum=return_sum)


class LayerScale(nn.Module):
    """vit_scale_clean.py:166-185 with force_fp32 (the tower's setting)."""

    def __init__(self, dim, init_values=1e-5):
        super().__init__()
        self.weight = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return (x.float() * self.weight.float()).to(x.dtype)


class Attention(nn.Module):
    """vit_scale_clean.py:188-297: both of its branches compute softmax(q k^T / sqrt(d)) v
    with q, k RMS-normalised across all heads at once."""

    def __init__(self, dim, num_heads, qkv_bias=False, qk_normalization=True):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.qk_normalization = qk_normalization
        self.q_norm = RMSNorm(dim) if qk_normalization else nn.Identity()
        self.k_norm = RMSNorm(dim) if qk_normalization else nn.Identity()
        self._padded = ZeroPaddedLinears()

    def forward(self, x):
        B, N, C = x.shape
        if ZeroPaddedLinears.wanted(x, 3 * C):      # 3 x 1408 = 4224 -> 4352 (GEMM tile multiple)
            qb = self.qkv.bias
            w, b = self._padded.get((self.qkv.weight,) + ((qb,) if qb is not None else ()), lambda: (
                _pad_rows(self.qkv.weight.detach(), _aligned(3 * C)),
                None if qb is None else _pad_rows(qb.detach(), _aligned(3 * C))))
            qkv = F.linear(x, w, b)
        else:
            qkv = self.qkv(x)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:3 * C]
        if self.qk_normalization:
            q, k = self.q_norm(q), self.k_norm(k)
        hd = (B, N, self.num_heads, self.head_dim)
        o = K.flash_attn_func(q.view(hd), k.view(hd), v.view(hd), softmax_scale=self.scale,
                              causal=False)
        return self.proj(o.reshape(B, N, C))


class FlashAttention(nn.Module):
    """flash_attention_class.py:16-110: softmax attention on packed qkv.  qkv (B, S, 3, H, D) with an optional
    `key_padding_mask` (B, S) bool (True = token present: the padded positions are removed before the kernel and come
    back as zeros, `unpad_input` / `pad_input`), or already unpadded (nnz, 3, H, D) with `cu_seqlens` / `max_s`.
    The tower itself calls the kernel on equal-length clips; this module keeps the reference's entry point whole."""

    def __init__(self, softmax_scale=None, attention_dropout=0.0, device=None, dtype=None):
        super().__init__()
        self.softmax_scale, self.dropout_p = softmax_scale, attention_dropout

    def forward(self, qkv, key_padding_mask=None, causal=False, cu_seqlens=None, max_s=None, need_weights=False):
        assert not need_weights
        assert qkv.dtype in (torch.float16, torch.bfloat16)
        if self.training and self.dropout_p:
            raise NotImplementedError("attention dropout is a training feature")
        if cu_seqlens is not None:
            assert max_s is not None
            return K.flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_s, 0.0, softmax_scale=self.softmax_scale,
                                                      causal=causal), None
        B, S = qkv.shape[:2]
        if key_padding_mask is None:
            o = K.flash_attn_func(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 0.0, self.softmax_scale, causal)
            return o, None
        nheads, hd = qkv.shape[-2], qkv.shape[-1]
        lens = key_padding_mask.sum(dim=-1, dtype=torch.int32)
        idx = torch.nonzero(key_padding_mask.flatten(), as_tuple=False).flatten()       # unpad_input
        cu = torch.nn.functional.pad(torch.cumsum(lens, 0, dtype=torch.int32), (1, 0))
        x = qkv.reshape(B * S, 3, nheads, hd)[idx]
        o_un = K.flash_attn_varlen_qkvpacked_func(x, cu, int(lens.max()), 0.0, softmax_scale=self.softmax_scale,
                                                  causal=causal)
        out = torch.zeros((B * S, nheads, hd), dtype=qkv.dtype, device=qkv.device)     # pad_input
        out[idx] = o_un
        return out.view(B, S, nheads, hd), None


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        w1 = self.fc1.weight
        # fc1 + bias + exact GELU in one kernel where the hand-written GEMM applies (csrc/gemm.hip, see
        # model/vit/siglip.py::Mlp._fused_fc1); vit_scale_clean.py:296-320
        if (x.is_cuda and x.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16 and not torch.is_grad_enabled()
                and w1.shape[1] % 128 == 0 and w1.shape[0] % 4 == 0 and x.numel() // x.shape[-1] >= 4096
                and os.environ.get("TV_VIT_FUSED_FC1", "1") != "0"):
            return self.fc2(K.linear_fused(x, w1, self.fc1.bias, epilogue=K.GEMM_BIAS_GELU))
        return self.fc2(K.gelu(self.fc1(x), inplace=True))


class Block(nn.Module):
    """vit_scale_clean.py:322-416 (non-fused-norm branch, drop_path inactive in eval)."""

    def __init__(self, dim, num_heads, mlp_ratio=48 / 11, qkv_bias=False, init_values=1e-5,
                 qk_normalization=True):
        super().__init__()
        self.norm1 = RMSNorm(dim)
        self.attn = Attention(dim, num_heads, qkv_bias, qk_normalization)
        self.ls1 = LayerScale(dim, init_values) if init_values else nn.Identity()
        self.norm2 = RMSNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.ls2 = LayerScale(dim, init_values) if init_values else nn.Identity()

    def forward_fused(self, x, delta):
        """(stream, pending sub-layer output) -> same pair after this block; the residual
        adds ride inside the RMSNorm kernels."""
        if delta is None:
            h = self.norm1(x)
        else:
            h, x = self.norm1(x, residual=delta, return_sum=True)
        a = self.ls1(self.attn(h))
        h, x = self.norm2(x, residual=a, return_sum=True)
        return x, self.ls2(self.mlp(h))

    def forward(self, x, residual=None):
        assert residual is None
        x, d = self.forward_fused(x, None)
        return x + d


class PatchEmbed(nn.Module):
    """vit_scale_clean.py:419-461."""

    def __init__(self, img_size=224, patch_size=14, in_chans=3, embed_dim=1408, num_frames=4,
                 tubelet_size=1):
        super().__init__()
        if tubelet_size != 1:
            raise NotImplementedError("the tower is built with tubelet_size=1 (model.py:161)")
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.grid_size = (num_frames, img_size // patch_size, img_size // patch_size)
        self.num_img_patches = self.grid_size[1] * self.grid_size[2]
        self.num_patches = self.grid_size[0] * self.num_img_patches
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=(1, patch_size, patch_size),
                              stride=(1, patch_size, patch_size))

    def forward(self, x):
        """(B, C, T, H, W) -> (B, T, HW/p^2, D)."""
        B, _, T = x.shape[:3]
        y = K.patch_embed_video(x, self.proj.weight, self.proj.bias)
        return y.view(B, T, -1, y.shape[-1])


class PretrainVisionTransformer_clean(nn.Module):
    def __init__(self, in_chans=3, patch_size=14, img_size=224, qkv_bias=False, embed_dim=1408,
                 num_heads=16, mlp_ratio=48 / 11, init_values=1e-5, qk_normalization=True,
                 depth=40, num_frames=4, tubelet_size=1, sep_image_video_pos_embed=True,
                 x_vis_return_idx=-2, x_vis_only=True, **unused):
        super().__init__()
        if not x_vis_only:
            raise NotImplementedError("clip_projector head is never built by the tower")
        self.num_frames, self.embed_dim = num_frames, embed_dim
        self.depth = depth + x_vis_return_idx + 1
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, num_frames,
                                      tubelet_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.sep_image_video_pos_embed = sep_image_video_pos_embed
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        if sep_image_video_pos_embed:
            self.img_pos_embed = nn.Parameter(
                torch.zeros(1, self.patch_embed.num_img_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, init_values, qk_normalization)
            for _ in range(self.depth)])
        self.init_pos_embed()
        nn.init.trunc_normal_(self.cls_token, std=0.02)

    def init_pos_embed(self):
        g, t = self.patch_embed.grid_size[1], self.patch_embed.grid_size[0]
        self.pos_embed.data.copy_(
            torch.from_numpy(sincos_pos_embed_3d(self.embed_dim, g, t)).float().unsqueeze(0))
        if self.sep_image_video_pos_embed:
            self.img_pos_embed.data.copy_(
                torch.from_numpy(sincos_pos_embed_3d(self.embed_dim, g, 1)).float().unsqueeze(0))

    @property
    def dtype(self):
        return self.patch_embed.proj.weight.dtype

    def _pos_table(self, use_image: bool):
        """vit_scale_clean.py:679-707."""
        if not use_image:
            return self.pos_embed
        if self.sep_image_video_pos_embed:
            return self.img_pos_embed
        per_frame = self.pos_embed[:, 1:].view(1, self.num_frames, -1, self.embed_dim).mean(dim=1)
        return torch.cat([self.pos_embed[:, :1], per_frame], dim=1)

    def forward(self, x, mask=None, use_image=False):
        if mask is not None:
            raise NotImplementedError("token masking is a pre-training feature")
        x = self.patch_embed(x.type(self.dtype))
        B, T, L, C = x.shape
        x = torch.cat((self.cls_token.expand(B, -1, -1).to(x.dtype), x.view(B, T * L, C)), dim=1)
        x = x + self._pos_table(use_image).to(x.dtype)
        delta = None
        for blk in self.blocks:
            x, delta = blk.forward_fused(x, delta)
        return x + delta


# ------------------------------------------------------------------ tower + backbone
@dataclass
class InternVideo2VisionConfig:
    """model.py:35-80 (fields the tower reads)."""
    num_frames: int = 4
    hidden_size: int = 1408
    num_hidden_layers: int = 40
    num_attention_heads: int = 16
    num_channels: int = 3
    image_size: int = 224
    patch_size: int = 14
    x_vis_return_idx: int = -2
    sep_image_video_pos_embed: bool = True
    use_checkpoint: bool = False
    checkpoint_num: int = 0
    vision_tower_path: Optional[str] = None
    model_type: str = "internvideo2_vision_model"


class InternVideo2VisionTower(nn.Module):
    """model.py:136-198."""

    def __init__(self, config: InternVideo2VisionConfig):
        super().__init__()
        self.config = config
        self.vision_tower = PretrainVisionTransformer_clean(
            in_chans=config.num_channels, img_size=config.image_size,
            patch_size=config.patch_size, embed_dim=config.hidden_size,
            depth=config.num_hidden_layers, num_heads=config.num_attention_heads,
            mlp_ratio=48 / 11, qkv_bias=False, init_values=0.00001, qk_normalization=True,
            num_frames=config.num_frames, tubelet_size=1,
            sep_image_video_pos_embed=config.sep_image_video_pos_embed,
            x_vis_return_idx=config.x_vis_return_idx, x_vis_only=True)
        self.vision_tower.requires_grad_(False)

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, is_video: Optional[bool] = None,
                clip_frames: Optional[int] = None):
        """Video input arrives (T, B, C, H, W); it is regrouped into 4-frame clips with the
        reference's permute + reshape (model.py:178-182) — for B == 1 and T > 4 that
        reshape, not a permute, decides which frames share a clip, and it is kept as is.
        `clip_frames`: the caller's clip length (256 in `generic_vlm.py:274`); a longer input is
        regrouped clip by clip exactly as separate calls would be and the tubes of all clips run
        through the tower in ONE batch (tubes are independent)."""
        if is_video is None:
            is_video = pixel_values.shape[1] > 1
        if is_video:
            C, H, W = pixel_values.shape[2:]
            step = clip_frames or pixel_values.shape[0]
            px = torch.cat([c.permute(1, 2, 0, 3, 4).reshape(c.shape[1] * (c.shape[0] // 4), C, 4, H, W)
                            for c in pixel_values.split(step)])
        else:
            px = pixel_values.permute(0, 2, 1, 3, 4)
        return self.vision_tower(px, use_image=not is_video)[:, 1:, :]

    @property
    def dtype(self):
        return next(self.vision_tower.parameters()).dtype

    @property
    def device(self):
        return next(self.vision_tower.parameters()).device


class InternVideo2ViTBackbone(VisionBackbone):
    """backbone.py:27-143 without the checkpoint download: weights come from
    `load_model_weights(path)` or stay at their synthetic initialisation."""

    def __init__(self, vision_backbone_id: str = "internvideo2-1b-16-224px",
                 image_resize_strategy: str = "resize-naive", default_image_size: int = 224,
                 num_frames: int = 4, vision_config: Optional[InternVideo2VisionConfig] = None):
        super().__init__(vision_backbone_id, image_resize_strategy, default_image_size)
        self.vision_config = vision_config or InternVideo2VisionConfig(
            image_size=default_image_size, num_frames=num_frames)
        self.featurizer = InternVideo2VisionTower(self.vision_config)
        self.featurizer.eval()
        self.dtype = torch.bfloat16

    def load_model_weights(self, ckpt_path: str):
        state = torch.load(ckpt_path, map_location="cpu")
        return self.featurizer.vision_tower.load_state_dict(state, strict=False)

    batched_clips = True     # forward(..., clip_frames=n) == per-clip calls concatenated

    def forward(self, pixel_values: torch.Tensor, is_video: Optional[bool] = None,
                clip_frames: Optional[int] = None, **kwargs):
        return self.featurizer(pixel_values, is_video=is_video, clip_frames=clip_frames)

    @property
    def get_identifier(self) -> str:
        return "internvideo2"

    @property
    def default_image_resolution(self) -> Tuple[int, int, int]:
        return (3, self.default_image_size, self.default_image_size)

    @property
    def embed_dim(self) -> int:
        return self.vision_config.hidden_size

    @property
    def num_patches(self) -> int:
        return self.featurizer.vision_tower.patch_embed.num_img_patches

    @property
    def half_precision_dtype(self) -> torch.dtype:
        return torch.bfloat16
