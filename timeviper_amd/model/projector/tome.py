"""ToMe token merging 729 -> 16 tokens/frame + MLP (reference
timeviper/model/projector/tome.py:14-231).  Every round is one call of the HIP operator
`kernels.tome_merge_round` (`bipartite_soft_matching` :14-67 + `merge_wavg` :70-83 in four
kernels: metric, matching, sort, size-weighted merge — csrc/tome.hip).  There is no host-side
path: CPU tensors raise, like every other operator (the step-by-step restatement that the
tests pin against the reference golden vectors is test infrastructure, not part of this package)."""
from typing import Dict, Union

import torch
import torch.nn as nn

from ... import kernels as K
from .mlp import _interleave


def merge_schedule(p: int, target: int):
    """r per round: halve until the remainder fits (tome.py:126-136); 729->16 gives
    [364, 182, 91, 46, 23, 7]."""
    assert p > target, f"{p} should greater than {target}"
    rs = []
    while p != target:
        if p - target <= p // 2:
            rs.append(p - target)
            break
        rs.append(p // 2)
        p -= p // 2
    return rs


class ToMe16_mlp_hd64(nn.Module):
    def __init__(self, vision_dim: int, llm_dim: int, mlp_type: str = "tome_mlp",
                 num_compressed_tokens: int = 16, token_order: str = "raw") -> None:
        super().__init__()
        self.num_attention_heads = 16
        self.num_compressed_tokens = num_compressed_tokens
        if mlp_type == "tome_mlp":
            self.projector = nn.Sequential(nn.Linear(vision_dim, llm_dim, bias=True), nn.GELU(),
                                           nn.Linear(llm_dim, llm_dim, bias=True))
        elif mlp_type == "fused_tome_mlp":
            d4 = vision_dim * 4
            self.initial_projection_dim = d4
            self.projector = nn.Sequential(nn.Linear(vision_dim, d4, bias=True), nn.GELU(),
                                           nn.Linear(d4, llm_dim, bias=True), nn.GELU(),
                                           nn.Linear(llm_dim, llm_dim, bias=True))
        else:
            raise ValueError(f"Fused Projector with `{mlp_type}` is not supported!")
        self.token_order = token_order

    def merge_tokens(self, x, target_num_token, token_order):
        size = None
        b, p, c = x.shape
        head = self.num_attention_heads
        for r in merge_schedule(p, target_num_token):
            x, size = K.tome_merge_round(x, size, min(r, p // 2), head)
            p = x.shape[1]
        if token_order in ("ascending", "descending"):
            idx = size.squeeze(-1).argsort(dim=1, descending=token_order == "descending")
            x = x.gather(dim=1, index=idx.unsqueeze(-1).expand(-1, -1, c))
        return x

    def forward(self, x, compress=False, local_num_frames=-1):
        if local_num_frames not in (-1, 1):
            assert compress is True
        if compress:
            if local_num_frames != -1:
                num_frames = local_num_frames
                x = x.reshape(x.shape[0], -1, x.shape[-1])
            else:
                num_frames = x.shape[0]
                x = x.reshape(1, -1, x.shape[-1])
            n_tok = self.num_compressed_tokens * num_frames
        else:
            n_tok = self.num_compressed_tokens * local_num_frames
        x = self.merge_tokens(x, target_num_token=n_tok, token_order=self.token_order)
        return self.projector(x)


class MultiToMe16_mlp_hd64(nn.Module):
    def __init__(self, vision_dims: Dict[str, int], llm_dim: int, mlp_type: str = "tome_mlp",
                 num_compressed_tokens: int = 16, token_order: str = "raw") -> None:
        super().__init__()
        if "tome_mlp" not in mlp_type:
            raise ValueError(f"Projector with `{mlp_type}` is not supported!")
        self.keys = list(vision_dims.keys())
        self.projectors = nn.ModuleDict({
            k: ToMe16_mlp_hd64(d, llm_dim, mlp_type, num_compressed_tokens, token_order)
            for k, d in vision_dims.items()})

    def forward(self, img_patches: Dict[str, torch.Tensor], compress=False,
                local_num_frames: Union[int, Dict[str, int]] = -1) -> torch.Tensor:
        outs = []
        for k in self.keys:
            lnf = local_num_frames.get(k, -1) if isinstance(local_num_frames, dict) else local_num_frames
            outs.append(self.projectors[k](img_patches[k], compress=compress, local_num_frames=lnf))
        return _interleave(outs)
