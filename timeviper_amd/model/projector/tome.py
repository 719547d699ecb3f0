"""ToMe token merging 729 -> 16 tokens/frame + MLP (reference
timeviper/model/projector/tome.py:14-231).  On the GPU every round is one call of the HIP
operator `kernels.tome_merge_round` (metric, bipartite matching, sort, size-weighted merge:
csrc/tome.hip); the torch functions below restate the reference step by step and are what
the CPU tests check against the reference golden vectors."""
from typing import Callable, Dict, Tuple, Union

import torch
import torch.nn as nn

from ... import kernels as K
from .mlp import _interleave


def bipartite_soft_matching(metric: torch.Tensor, r: int) -> Tuple[Callable, Callable]:
    """Balanced (even/odd) bipartite matching; returns merge(x, mode) (tome.py:14-67).
    metric (B, T, C); r tokens of the even set are merged into their best odd match."""
    t = metric.shape[1]
    r = min(r, t // 2)
    assert r > 0, r
    with torch.no_grad():
        unit = metric / metric.norm(dim=-1, keepdim=True)
        even, odd = unit[..., ::2, :], unit[..., 1::2, :]
        sim = even @ odd.transpose(-1, -2)
        best_val, best_odd = sim.max(dim=-1)
        order = best_val.argsort(dim=-1, descending=True)[..., None]
        keep_idx, src_idx = order[..., r:, :], order[..., :r, :]
        dst_idx = best_odd[..., None].gather(dim=-2, index=src_idx)

    def merge(x: torch.Tensor, mode="mean") -> torch.Tensor:
        ev, od = x[..., ::2, :], x[..., 1::2, :]
        n, t1, c = ev.shape
        kept = ev.gather(dim=-2, index=keep_idx.expand(n, t1 - r, c))
        moved = ev.gather(dim=-2, index=src_idx.expand(n, r, c))
        od = od.scatter_add(-2, dst_idx.expand(n, r, c), moved)
        return torch.cat([kept, od], dim=1)

    def unmerge(x: torch.Tensor) -> torch.Tensor:
        nk = keep_idx.shape[1]
        kept, od = x[..., :nk, :], x[..., nk:, :]
        n, _, c = kept.shape
        moved = od.gather(dim=-2, index=dst_idx.expand(n, r, c))
        out = torch.zeros(n, metric.shape[1], c, device=x.device, dtype=x.dtype)
        out[..., 1::2, :] = od
        out.scatter_(dim=-2, index=(2 * keep_idx).expand(n, nk, c), src=kept)
        out.scatter_(dim=-2, index=(2 * src_idx).expand(n, r, c), src=moved)
        return out

    return merge, unmerge


def merge_wavg(merge: Callable, x: torch.Tensor, size: torch.Tensor = None):
    """Size-weighted average merge (tome.py:70-83)."""
    if size is None:
        size = torch.ones_like(x[..., 0, None])
    x = merge(x * size, mode="sum")
    size = merge(size, mode="sum")
    return x / size, size


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
            if x.is_cuda:       # HIP kernels: metric, matching, sort and weighted merge of a round
                x, size = K.tome_merge_round(x, size, min(r, p // 2), head)
            else:               # host-side restatement (CPU tests against the reference golden)
                metric = x.reshape(b, p, head, c // head).mean(2)
                merge, _ = bipartite_soft_matching(metric, r)
                x, size = merge_wavg(merge, x, size)
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
