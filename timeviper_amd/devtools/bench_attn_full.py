"""Dev tool: the attention kernel at the sizes of the default run — causal d=128 at 131 k keys (after
the first pdrop stage) and the ViT shape in a 2 048-frame launch with padded qkv rows (stride 3 584)."""
import sys
import torch
sys.path.insert(0, ".")
from timeviper_amd import kernels as K  # noqa: E402
from timeviper_amd.devtools.bench_ops import timeit  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
La = 131172
q, k, v = rn(1, La, 40, 128), rn(1, La, 8, 128), rn(1, La, 8, 128)
ms = timeit(lambda: K.flash_attn_func(q, k, v, causal=True), iters=2, warmup=1)
print(f"causal L={La}: {ms:8.2f} ms  {2 * La * La * 40 * 128 / ms / 1e9:7.1f} TFLOP/s")
del q, k, v
F_ = 2048
qkv = rn(F_, 729, 3584)[..., :3456].unflatten(-1, (3, 16, 72))
ms = timeit(lambda: K.flash_attn_func(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=False), iters=3, warmup=1)
print(f"ViT {F_}x729 d72: {ms:8.2f} ms  {4 * F_ * 16 * 729 * 729 * 72 / ms / 1e9:7.1f} TFLOP/s")
