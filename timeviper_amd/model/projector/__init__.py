from .mlp import MLPProjector, MultiMLPProjector
from .tome import MultiToMe16_mlp_hd64, ToMe16_mlp_hd64

__all__ = ["MLPProjector", "MultiMLPProjector", "ToMe16_mlp_hd64", "MultiToMe16_mlp_hd64",
           ]
