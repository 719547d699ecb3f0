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
