"""Dev tool: SigLIP-so400m tower on one 256-frame clip, run in sub-clips of s frames (activations of
a sub-clip may stay in the 256 MB Infinity Cache between the kernels of a block)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from timeviper_amd.model.vit import TimmViTBackbone  # noqa: E402

dev = torch.device("cuda", 0)
with torch.device("meta"):
    vb = TimmViTBackbone("siglip-vit-so400m-384px")
vb = vb.to_empty(device=dev)
with torch.no_grad():
    for n, p in vb.named_parameters():
        if p.dim() > 1:
            p.normal_(0, 0.02)
        elif n.endswith("bias"):
            p.zero_()
        else:
            p.fill_(1.0)
vb = vb.bfloat16().eval()
NF = 2048
pix = torch.randn(NF, 3, 384, 384, device=dev, dtype=torch.bfloat16)
for s in [int(a) for a in sys.argv[1:]] or [256, 128, 64, 32]:
    with torch.inference_mode():
        f = lambda: torch.cat([vb(c) for c in pix.split(s)])
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
    print(f"sub-clip {s:4d} frames: {dt * 1e3:7.1f} ms per {NF} frames = {NF / dt:6.0f} frames/s")
