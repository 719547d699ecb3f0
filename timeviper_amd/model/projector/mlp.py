"""Linear-GELU-Linear projectors (reference timeviper/model/projector/mlp.py:13-68).
Parameter names `projector.{0,2}.{weight,bias}` are part of the checkpoint contract."""
from typing import Dict

import torch
import torch.nn as nn


def _interleave(outputs):
    """Two-encoder fusion rule shared by the Multi* projectors (mlp.py:52-68,
    tome.py:214-231): equalise shapes when element counts agree, then interleave
    token-wise if the token counts match, else concatenate along tokens."""
    if len(outputs) == 2:
        a, b = outputs
        if a.shape != b.shape and a.numel() == b.numel():
            if a.shape[0] > b.shape[0]:
                outputs[1] = b.reshape(a.shape)
            else:
                outputs[0] = a.reshape(b.shape)
    if outputs[0].shape[1] != outputs[1].shape[1]:
        return torch.cat(outputs, dim=1)
    return torch.stack(outputs, dim=2).flatten(1, 2)


class MLPProjector(nn.Module):
    def __init__(self, vision_dim: int, llm_dim: int, mlp_type: str = "gelu_mlp") -> None:
        super().__init__()
        if mlp_type != "gelu_mlp":
            raise ValueError(f"Projector with `{mlp_type}` is not supported!")
        self.projector = nn.Sequential(nn.Linear(vision_dim, llm_dim, bias=True), nn.GELU(),
                                       nn.Linear(llm_dim, llm_dim, bias=True))

    def forward(self, img_patches: torch.Tensor) -> torch.Tensor:
        return self.projector(img_patches)


class MultiMLPProjector(nn.Module):
    def __init__(self, vision_dims: Dict[str, int], llm_dim: int, mlp_type: str = "gelu_mlp"):
        super().__init__()
        self.keys = list(vision_dims.keys())
        self.projectors = nn.ModuleDict({k: MLPProjector(d, llm_dim, mlp_type)
                                         for k, d in vision_dims.items()})

    def forward(self, img_patches: Dict[str, torch.Tensor]) -> torch.Tensor:
        return _interleave([self.projectors[k](img_patches[k]) for k in self.keys])
