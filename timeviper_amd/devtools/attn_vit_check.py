"""Dev tool: ViT-shaped attention (frames x 729 tokens x 16 heads x 72) through the kernel variant that
TV_FA_STREAM selects (0: one work-group per query block, 1: streaming, the default), checked
against an fp32 reference on a few frames, and timed.   python timeviper_amd/devtools/attn_vit_check.py [frames=256] [L=729]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from timeviper_amd import kernels as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 729
H, D = 16, 72
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, L, 3, H, D, device="cuda", dtype=torch.bfloat16, generator=g)
qkv[:, :, 0] *= 3.0                                    # sharper softmax: the running maximum moves
q, k, v = qkv.unbind(2)
o, lse = K.flash_attn_func(q, k, v, return_lse=True)
torch.cuda.synchronize()
worst = 0.0
for b in sorted({0, 1, B // 2, B - 2, B - 1}):
    if b < 0 or b >= B:
        continue
    qf, kf, vf = (t[b].float().transpose(0, 1) for t in (q, k, v))          # (H, L, D)
    s = qf @ kf.transpose(1, 2) / D ** 0.5
    ref = torch.softmax(s, -1) @ vf
    err = ((o[b].float().transpose(0, 1) - ref).norm() / ref.norm()).item()
    lerr = (lse[b] - torch.logsumexp(s, -1)).abs().max().item()
    worst = max(worst, err)
    print(f"frame {b}: rel L2 {err:.3e}  max |lse err| {lerr:.3e}")
assert worst < 1e-2 and torch.isfinite(o.float()).all(), worst
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    K.flash_attn_func(q, k, v)
e0.record()
for _ in range(20):
    K.flash_attn_func(q, k, v)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"TV_FA_STREAM={os.environ.get('TV_FA_STREAM', '(default)')}: {ms:.3f} ms  {4 * B * H * L * L * D / ms / 1e9:.0f} TFLOP/s")
