"""Vision backbones (reference timeviper/model/vit/): the `VisionBackbone` contract of
base_vision.py:77-124 and the registry ids of registry.py:23-113."""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Tuple

import torch
import torch.nn as nn

from .siglip import TIMM_VIT_CONFIGS, VisionTransformer

VISION_MODEL_REGISTRY = {
    "siglip-vit-b16-224px": ("siglip", "vit_base_patch16_siglip_224", 224),
    "siglip-vit-b16-256px": ("siglip", "vit_base_patch16_siglip_256", 256),
    "siglip-vit-b16-384px": ("siglip", "vit_base_patch16_siglip_384", 384),
    "siglip-vit-so400m": ("siglip", "vit_so400m_patch14_siglip_224", 224),
    "siglip-vit-so400m-384px": ("siglip", "vit_so400m_patch14_siglip_384", 384),
    "dinov2-vit-l": ("dinov2", "vit_large_patch14_reg4_dinov2.lvd142m", 224),
}

INTERNVIDEO2_REGISTRY = {      # registry.py:64-73
    "internvideo2-1b-16-224px": {"default_image_size": 224, "num_frames": 4,
                                 "vision_tower_path": "./ckpts/InternVideo2-1B_f4_vision.pt"},
}


MULTI_REGISTRY = {              # registry.py:74-82
    "dinosiglip-vit-so-384px": {"backbones": ["dinov2-vit-l", "siglip-vit-so400m-384px"],
                                "default_image_size": 384},
}


def get_vision_backbone_config(vision_backbone_id: str) -> Dict[str, Any]:
    if "+" in vision_backbone_id:          # registry.py:87-99
        members = vision_backbone_id.split("+")
        size = max(get_vision_backbone_config(b).get("default_image_size", 224) for b in members)
        return {"type": "multi", "vision_family": "multi", "identifier": vision_backbone_id,
                "backbones": members, "default_image_size": size}
    if vision_backbone_id in MULTI_REGISTRY:
        return {"type": "multi", "vision_family": "multi", "identifier": "multi",
                **MULTI_REGISTRY[vision_backbone_id]}
    if vision_backbone_id in VISION_MODEL_REGISTRY:
        fam, timm_id, size = VISION_MODEL_REGISTRY[vision_backbone_id]
        return {"type": "timm", "timm_id": timm_id, "default_image_size": size,
                "vision_family": fam, "identifier": fam}
    if vision_backbone_id in INTERNVIDEO2_REGISTRY:
        return {"type": "internvideo2", "vision_family": "internvideo2", "identifier": "internvideo2",
                **INTERNVIDEO2_REGISTRY[vision_backbone_id]}
    raise ValueError(f"Vision Backbone `{vision_backbone_id}` is not supported!")


class VisionBackbone(nn.Module):
    def __init__(self, vision_backbone_id: str, image_resize_strategy: str,
                 default_image_size: int = 224) -> None:
        super().__init__()
        self.identifier = vision_backbone_id
        self.image_resize_strategy = image_resize_strategy
        self.default_image_size = default_image_size
        self.featurizer: nn.Module = None
        self.image_transform = None

    def get_image_transform(self):
        return self.image_transform

    @property
    def get_identifier(self) -> str:
        return "vanilla-vision-backbone"


class TimmViTBackbone(VisionBackbone):
    """`featurizer` = ViT restated with timm parameter names; forward returns the
    second-to-last block's patch features (base_vision.py:165-170, :274-278)."""

    def __init__(self, vision_backbone_id: str, image_resize_strategy: str = "resize-naive",
                 default_image_size: Optional[int] = None, depth_override: Optional[int] = None):
        cfg = get_vision_backbone_config(vision_backbone_id)
        super().__init__(vision_backbone_id, image_resize_strategy,
                         default_image_size or cfg["default_image_size"])
        self.cfg = cfg
        kw = dict(TIMM_VIT_CONFIGS[cfg["timm_id"]])
        kw["img_size"] = self.default_image_size
        if depth_override is not None:
            kw["depth"] = depth_override
        self.featurizer = VisionTransformer(**kw)
        self.featurizer.eval()
        self.dtype = torch.bfloat16

    frame_independent = True     # every frame is encoded on its own: any clip size gives the same rows

    def forward(self, pixel_values: torch.Tensor, **kwargs) -> torch.Tensor:
        return self.featurizer(pixel_values)

    @property
    def get_identifier(self) -> str:
        return self.cfg.get("identifier", self.identifier)

    @property
    def default_image_resolution(self) -> Tuple[int, int, int]:
        return (3, self.default_image_size, self.default_image_size)

    @property
    def embed_dim(self) -> int:
        return self.featurizer.embed_dim

    @property
    def num_patches(self) -> int:
        return self.featurizer.patch_embed.num_patches

    @property
    def half_precision_dtype(self) -> torch.dtype:
        return torch.bfloat16


TimmCheckpointBackbone = TimmViTBackbone


from .internvideo2 import (InternVideo2ViTBackbone, InternVideo2VisionConfig,  # noqa: E402
                           InternVideo2VisionTower)
from .multivit import MultiViTBackbone  # noqa: E402
