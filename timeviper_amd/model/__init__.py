"""Factories mirroring timeviper/model/__init__.py:40-133 (`get_vision_backbone_and_transform`,
`get_llm_backbone_and_tokenizer`, `get_vlm`), for the backbones on the hot path."""
from __future__ import annotations

from typing import Optional

import torch

from .generic_vlm import GenericTimeViperVLM, HybridTimeViperVLM
from .llm import GenericLLMBackbone, NemotronHConfig
from .vit import (InternVideo2ViTBackbone, MultiViTBackbone, TimmViTBackbone, VisionBackbone,
                  get_vision_backbone_config)


def get_vision_backbone_and_transform(vision_backbone_id: str, image_resize_strategy: str = "resize-naive",
                                      use_zero3: bool = False, **kw):
    cfg = get_vision_backbone_config(vision_backbone_id)
    if cfg["type"] == "internvideo2":
        vb = InternVideo2ViTBackbone(vision_backbone_id, image_resize_strategy, **kw)
    elif cfg["type"] == "multi":
        vb = MultiViTBackbone(vision_backbone_id, image_resize_strategy, **kw)
    else:
        vb = TimmViTBackbone(vision_backbone_id, image_resize_strategy, **kw)
    return vb, vb.get_image_transform()


def get_llm_backbone_and_tokenizer(llm_backbone_id: str, llm_max_length: Optional[int] = None,
                                   hf_token: Optional[str] = None, inference_mode: bool = False,
                                   attn_implementation: str = "flash_attention_2",
                                   continue_pretrain_ckpt=None, merge_module: str = "no_merge",
                                   use_pdrop: bool = False, pdrop_type: Optional[str] = None,
                                   config: Optional[NemotronHConfig] = None):
    llm = GenericLLMBackbone(llm_backbone_id, config=config, llm_max_length=llm_max_length,
                             inference_mode=inference_mode, attn_implementation=attn_implementation,
                             merge_module=merge_module, use_pdrop=use_pdrop, pdrop_type=pdrop_type)
    return llm, llm.tokenizer


def get_vlm(model_id: str, vision_backbone: VisionBackbone, llm_backbone: GenericLLMBackbone,
            arch_specifier: str = "tome_mlp-16", visual_token_order: str = "raw", **kw):
    return HybridTimeViperVLM(model_id, vision_backbone, llm_backbone,
                              arch_specifier=arch_specifier, visual_token_order=visual_token_order, **kw)


def build_synthetic_timeviper(llm_config=None,
                              vision_backbone_id: str = "siglip-vit-so400m-384px",
                              pdrop_type: Optional[str] = None, merge_module: str = "no_merge",
                              device="cuda", dtype=torch.bfloat16, seed: int = 0,
                              vit_depth: Optional[int] = None, image_size: Optional[int] = None,
                              vision_config=None, llm_backbone_id: str = "nanov2-9b",
                              member_kwargs=None):
    """Random-init TimeViper (there are no checkpoints offline): weights N(0, 0.02),
    A_log = log U[1,16], dt_bias = softplus^-1(U[1e-3,1e-1]), D = 1."""
    torch.manual_seed(seed)
    with torch.device("meta"):
        vtype = get_vision_backbone_config(vision_backbone_id)["type"]
        if vtype == "multi":
            vb = MultiViTBackbone(vision_backbone_id, default_image_size=image_size, member_kwargs=member_kwargs)
        elif vtype == "internvideo2":
            vb = InternVideo2ViTBackbone(vision_backbone_id, default_image_size=image_size or 224,
                                         vision_config=vision_config)
        else:
            vb = TimmViTBackbone(vision_backbone_id, depth_override=vit_depth, default_image_size=image_size)
        llm = GenericLLMBackbone(llm_backbone_id, config=llm_config, merge_module=merge_module,
                                 use_pdrop=pdrop_type is not None, pdrop_type=pdrop_type)
        vlm = HybridTimeViperVLM("timeviper-synthetic", vb, llm, arch_specifier="tome_mlp-16")
    vlm = vlm.to_empty(device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in vlm.named_parameters():
            if name.endswith("A_log"):
                p.copy_(torch.log(torch.rand(p.shape, device=device, generator=g) * 15 + 1))
            elif name.endswith("dt_bias"):
                lo, hi = 1e-3, 1e-1
                dt = torch.exp(torch.rand(p.shape, device=device, generator=g)
                               * (torch.log(torch.tensor(hi)) - torch.log(torch.tensor(lo)))
                               + torch.log(torch.tensor(lo)))
                p.copy_(dt + torch.log(-torch.expm1(-dt)))
            elif name.endswith(".D") or "norm" in name.split(".")[-2] and name.endswith("weight"):
                p.fill_(1.0)
            elif name.endswith("gamma") or name.endswith(("ls1.weight", "ls2.weight")):
                p.fill_(1.0)
            elif name.endswith("alpha"):
                p.fill_(0.5)
            elif name.endswith("bias"):
                p.zero_()
            else:
                p.normal_(0.0, 0.02, generator=g)
    vlm = vlm.to(dtype).eval()
    return vlm


__all__ = ["GenericTimeViperVLM", "HybridTimeViperVLM", "get_vision_backbone_and_transform",
           "get_llm_backbone_and_tokenizer", "get_vlm", "build_synthetic_timeviper"]
