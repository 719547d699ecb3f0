"""`MultiViTBackbone`: several vision encoders over the same frames (BASELINE config 4: DINOv2-L +
InternVideo2-1B).

The reference imports this class from `timeviper/model/vit/multivit_backbones.py`
(vit/__init__.py:3, constructed at model/__init__.py:54-57), but that file is not part of the
reference tree.  It is therefore specified here from its callers:

* `backbone_ids` — list of member ids, `backbones` — `nn.ModuleDict` keyed by
  `bid.replace("-", "_")`, each member exposing `embed_dim` (generic_vlm.py:180-186);
* `get_identifier == "multivit"` (generic_vlm.py:415);
* `forward(pixel_values, is_video=...)` receives ONE tensor (the caller splits it into 256-frame
  clips with `Tensor.split`, generic_vlm.py:274-279) and returns `{bid: patch_features}`, the dict
  `MultiToMe16_mlp_hd64` / `MultiMLPProjector` index by `bid` (projector/tome.py:200-213);
* ids: `"a+b"` (registry.py:87-99, `default_image_size` = the largest member's) or the named
  variant `dinosiglip-vit-so-384px` (registry.py:74-82).

Two things the callers do not determine are fixed here and stated as this package's behaviour:
frames are resized (bilinear, no antialias) to a member's own `default_image_size` when they arrive
at a different size, and the frame tensor may be (T, C, H, W) or (T, 1, C, H, W) — image encoders
get the 4-D form, InternVideo2 the 5-D form it expects (model.py:178-182)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import (InternVideo2ViTBackbone, TimmViTBackbone, VisionBackbone, get_vision_backbone_config)


class MultiViTBackbone(VisionBackbone):
    def __init__(self, vision_backbone_id: str, image_resize_strategy: str = "resize-naive",
                 default_image_size: Optional[int] = None, member_kwargs: Optional[Dict[str, dict]] = None):
        cfg = get_vision_backbone_config(vision_backbone_id)
        if cfg["type"] != "multi":
            raise ValueError(f"`{vision_backbone_id}` is not a multi-encoder id")
        super().__init__(vision_backbone_id, image_resize_strategy,
                         default_image_size or cfg["default_image_size"])
        self.backbone_ids = list(cfg["backbones"])
        if len(set(self.backbone_ids)) != len(self.backbone_ids):
            raise ValueError("a member id may appear only once")
        member_kwargs = member_kwargs or {}
        self.backbones = nn.ModuleDict()
        for bid in self.backbone_ids:
            kind = get_vision_backbone_config(bid)["type"]
            kw = dict(member_kwargs.get(bid, {}))
            if kind == "timm":
                member = TimmViTBackbone(bid, image_resize_strategy, **kw)
            elif kind == "internvideo2":
                member = InternVideo2ViTBackbone(bid, image_resize_strategy, **kw)
            else:
                raise ValueError(f"`{bid}` cannot be a member of a multi-encoder backbone")
            self.backbones[bid.replace("-", "_")] = member
        self.dtype = torch.bfloat16

    def _frames_for(self, member: VisionBackbone, pixel_values: torch.Tensor) -> torch.Tensor:
        px = pixel_values
        if px.dim() == 5:
            if px.shape[1] != 1:
                raise ValueError("multi-encoder input is (T, C, H, W) or (T, 1, C, H, W)")
            px = px[:, 0]
        elif px.dim() != 4:
            raise ValueError(f"expected 4-D or 5-D frames, got {tuple(px.shape)}")
        size = member.default_image_size
        if px.shape[-2:] != (size, size):
            px = F.interpolate(px, size=(size, size), mode="bilinear", align_corners=False)
        return px.unsqueeze(1) if isinstance(member, InternVideo2ViTBackbone) else px

    batched_clips = True     # members are frame-independent or take `clip_frames` themselves

    def forward(self, pixel_values: torch.Tensor, is_video: Optional[bool] = None,
                clip_frames: Optional[int] = None, **kwargs):
        out = {}
        for bid in self.backbone_ids:
            member = self.backbones[bid.replace("-", "_")]
            px = self._frames_for(member, pixel_values)
            if isinstance(member, InternVideo2ViTBackbone):
                out[bid] = member(px, is_video=bool(is_video), clip_frames=clip_frames)
            else:
                out[bid] = member(px)
        return out

    @property
    def get_identifier(self) -> str:
        return "multivit"

    @property
    def default_image_resolution(self):
        return (3, self.default_image_size, self.default_image_size)

    @property
    def embed_dim(self) -> int:
        # read by `_initialize_projector` before it branches on `backbone_ids` (generic_vlm.py:178)
        return sum(m.embed_dim for m in self.backbones.values())

    @property
    def num_patches(self) -> int:
        return sum(m.num_patches for m in self.backbones.values())

    @property
    def half_precision_dtype(self) -> torch.dtype:
        return torch.bfloat16
