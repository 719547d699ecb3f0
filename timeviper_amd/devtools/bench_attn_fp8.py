"""Dev tool: fp8 vs bf16 attention kernels — errors against the fp32 definition on sampled rows and
kernel times.   python timeviper_amd/devtools/bench_attn_fp8.py [L=32868] [Hq=28] [Hkv=4] [D=128] [causal=1]"""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from timeviper_amd import kernels as K  # noqa: E402


def main():
    a = [int(x) for x in sys.argv[1:]]
    L, Hq, Hkv, D, causal = (a + [32868, 28, 4, 128, 1][len(a):])[:5]
    B = 1 if L > 2048 else 256
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(B, L, Hq, D, device=dev, generator=g).bfloat16()
    k = torch.randn(B, L, Hkv, D, device=dev, generator=g).bfloat16()
    v = torch.randn(B, L, Hkv, D, device=dev, generator=g).bfloat16()
    flops = 4.0 * B * L * L * D * Hq / (2 if causal else 1)
    outs = {}
    for name, fn in (("bf16", K.flash_attn_func), ("fp8", K.flash_attn_fp8_func)):
        for _ in range(2):
            o = fn(q, k, v, causal=bool(causal))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            o = fn(q, k, v, causal=bool(causal))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        outs[name] = o
        print(f"{name}: {ms:.3f} ms  {flops / ms / 1e9:.0f} TFLOP/s (incl. pre-pass for fp8)")
    rows = [0, 1, 63, 64, L // 3, L - 1]
    rep = Hq // Hkv
    for i in rows:
        hi = i + 1 if causal else L
        qi = q[0, i].float().view(Hkv, rep, D)
        sc = torch.einsum("ghd,lgd->ghl", qi, k[0, :hi].float()) / math.sqrt(D)
        ref = torch.einsum("ghl,lgd->ghd", torch.softmax(sc, -1), v[0, :hi].float()).reshape(Hq, D)
        e = {n: ((o[0, i].float() - ref).norm() / ref.norm()).item() for n, o in outs.items()}
        print(f"row {i}: rel err bf16 {e['bf16']:.3e}  fp8 {e['fp8']:.3e}")
    d = outs["fp8"].float() - outs["bf16"].float()
    print(f"fp8 vs bf16: rel L2 {(d.norm() / outs['bf16'].float().norm()).item():.3e}  max abs {d.abs().max().item():.3e}")


if __name__ == "__main__":
    main()
