"""CPU oracle for the SigLIP/DINOv2-style ViT path — TEST INFRASTRUCTURE ONLY.

The reference delegates this arithmetic to **timm** (third-party, un-vendored, version
not pinned; call sites timeviper/model/vit/base_vision.py:146-170, :274-278), so there
is no in-repo twin to run: PARITY UNPINNED.  This restates timm's public
`VisionTransformer` forward for `get_intermediate_layers(n={depth-2})`: Conv2d patch
embedding, learned position embedding, pre-norm blocks (LayerNorm eps 1e-6, MHA with
scale head_dim^-0.5, GELU MLP), output of block depth-2 without the final norm."""
import torch
import torch.nn.functional as F

from . import ops


def vit_intermediate_ref(sd, pixels, depth, num_heads, patch, take=None, eps=1e-6):
    take = depth - 2 if take is None else take
    x = ops.patch_embed_ref(pixels, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"])
    x = x + sd["pos_embed"].float()
    D = x.shape[-1]
    hd = D // num_heads
    for i in range(take + 1):
        p = f"blocks.{i}."
        h = F.layer_norm(x, (D,), sd[p + "norm1.weight"].float(), sd[p + "norm1.bias"].float(), eps)
        qkv = F.linear(h, sd[p + "attn.qkv.weight"].float(), sd[p + "attn.qkv.bias"].float())
        B, N, _ = qkv.shape
        qkv = qkv.view(B, N, 3, num_heads, hd)
        o, _ = ops.attention_ref(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=False,
                                 scale=hd ** -0.5)
        x = x + F.linear(o.reshape(B, N, D), sd[p + "attn.proj.weight"].float(),
                         sd[p + "attn.proj.bias"].float())
        h = F.layer_norm(x, (D,), sd[p + "norm2.weight"].float(), sd[p + "norm2.bias"].float(), eps)
        h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"].float(), sd[p + "mlp.fc1.bias"].float()))
        x = x + F.linear(h, sd[p + "mlp.fc2.weight"].float(), sd[p + "mlp.fc2.bias"].float())
    return x


# ---------------------------------------------------------------- InternVideo2 tower
def internvideo2_rmsnorm_ref(x, w, eps=1e-6):
    """vit_scale_clean.py:152-163: fp32 statistics, cast back, then the weight."""
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(x.dtype)


def internvideo2_tower_ref(sd, pixel_values, num_heads, is_video=None, eps=1e-6):
    """`InternVideo2VisionTower.forward` (model.py:173-190) over
    `PretrainVisionTransformer_clean.forward` (vit_scale_clean.py:664-722), PINNED by
    tests/golden/internvideo2.npz (generated from the reference itself).
    sd: vision_tower state dict (the number of `blocks.*` entries is the depth run).
    pixel_values: (T, B, C, H, W) for video, (B, 1, C, H, W) for images."""
    if is_video is None:
        is_video = pixel_values.shape[1] > 1
    if is_video:
        T, B, C, H, W = pixel_values.shape
        px = pixel_values.permute(1, 2, 0, 3, 4).reshape(B * (T // 4), C, 4, H, W)   # model.py:180-182
    else:
        px = pixel_values.permute(0, 2, 1, 3, 4)
    w = sd["patch_embed.proj.weight"]
    x = F.conv3d(px.to(w.dtype), w, sd["patch_embed.proj.bias"], stride=w.shape[2:])
    x = x.flatten(3).permute(0, 2, 3, 1)                       # (B, T, L, D)  :455-460
    Bc, Tc, L, D = x.shape
    x = torch.cat([sd["cls_token"].expand(Bc, -1, -1), x.reshape(Bc, Tc * L, D)], dim=1)
    if is_video:
        pos = sd["pos_embed"]
    elif "img_pos_embed" in sd:
        pos = sd["img_pos_embed"]
    else:                                                      # :688-703 joint table, frame mean
        nf = (sd["pos_embed"].shape[1] - 1) // L
        pos = torch.cat([sd["pos_embed"][:, :1],
                         sd["pos_embed"][:, 1:].view(1, nf, L, D).mean(dim=1)], dim=1)
    x = x + pos
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    hd = D // num_heads
    for i in range(depth):
        p = f"blocks.{i}."
        h = internvideo2_rmsnorm_ref(x, sd[p + "norm1.weight"], eps)
        qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd.get(p + "attn.qkv.bias"))
        q, k, v = qkv.split(D, dim=-1)
        q = internvideo2_rmsnorm_ref(q, sd[p + "attn.q_norm.weight"], eps)     # over all heads, :238-249
        k = internvideo2_rmsnorm_ref(k, sd[p + "attn.k_norm.weight"], eps)
        N = q.shape[1]
        o, _ = ops.attention_ref(q.view(Bc, N, num_heads, hd), k.view(Bc, N, num_heads, hd),
                                 v.reshape(Bc, N, num_heads, hd), causal=False, scale=hd ** -0.5)
        a = F.linear(o.reshape(Bc, N, D).to(x.dtype), sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        x = x + (a.float() * sd[p + "ls1.weight"].float()).to(x.dtype)         # LayerScale fp32 :166-185
        h = internvideo2_rmsnorm_ref(x, sd[p + "norm2.weight"], eps)
        h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        m = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        x = x + (m.float() * sd[p + "ls2.weight"].float()).to(x.dtype)
    return x[:, 1:, :]


# ---------------------------------------------------------------------------------------------
# ToMe (reference timeviper/model/projector/tome.py) — pinned by tests/golden/tome.npz
def tome_bipartite_matching_ref(metric: torch.Tensor, r: int):
    """tome.py:14-67 (the `merge` closure only; `unmerge` is never called on the path).
    metric (B, T, C); the r even tokens most similar to an odd token are merged into it."""
    t = metric.shape[1]
    r = min(r, t // 2)
    assert r > 0, r
    unit = metric / metric.norm(dim=-1, keepdim=True)
    even, odd = unit[..., ::2, :], unit[..., 1::2, :]
    sim = even @ odd.transpose(-1, -2)
    best_val, best_odd = sim.max(dim=-1)
    order = best_val.argsort(dim=-1, descending=True)[..., None]
    keep_idx, src_idx = order[..., r:, :], order[..., :r, :]
    dst_idx = best_odd[..., None].gather(dim=-2, index=src_idx)

    def merge(x: torch.Tensor) -> torch.Tensor:
        ev, od = x[..., ::2, :], x[..., 1::2, :]
        n, t1, c = ev.shape
        kept = ev.gather(dim=-2, index=keep_idx.expand(n, t1 - r, c))
        moved = ev.gather(dim=-2, index=src_idx.expand(n, r, c))
        od = od.scatter_add(-2, dst_idx.expand(n, r, c), moved)
        return torch.cat([kept, od], dim=1)

    return merge


def tome_merge_round_ref(x: torch.Tensor, size, r: int, heads: int):
    """One round of `merge_tokens` (tome.py:137-146): metric = mean over `heads` channel chunks,
    bipartite matching, size-weighted average (`merge_wavg`, tome.py:70-83)."""
    b, p, c = x.shape
    metric = x.reshape(b, p, heads, c // heads).mean(2)
    merge = tome_bipartite_matching_ref(metric, r)
    if size is None:
        size = torch.ones_like(x[..., 0, None])
    xs = merge(x * size)
    size = merge(size)
    return xs / size, size


def tome_schedule_ref(p: int, target: int):
    """tome.py:126-136: halve until the remainder fits; 729 -> 16 gives [364, 182, 91, 46, 23, 7]."""
    rs = []
    while p != target:
        if p - target <= p // 2:
            rs.append(p - target)
            break
        rs.append(p // 2)
        p -= p // 2
    return rs


def tome_merge_tokens_ref(x: torch.Tensor, target: int, heads: int = 16):
    """`ToMe16_mlp_hd64.merge_tokens` with token_order "raw" (tome.py:118-152)."""
    size = None
    for r in tome_schedule_ref(x.shape[1], target):
        x, size = tome_merge_round_ref(x, size, min(r, x.shape[1] // 2), heads)
    return x

