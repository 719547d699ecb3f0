"""Dev tool: do hipBLASLt's ViT GEMMs run faster with tile-aligned N / K (zero-padded weights)?"""
import torch
import torch.nn.functional as F
from bench_gemm import timeit

import sys
M = int(sys.argv[1]) * 729 if len(sys.argv) > 1 else 2048 * 729
for name, K, N in [("fc1 N=4304", 1152, 4304), ("fc1 N=4352", 1152, 4352), ("fc2 K=4304", 4304, 1152),
                   ("fc2 K=4352", 4352, 1152), ("qkv N=3456", 1152, 3456), ("qkv N=3584", 1152, 3584)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: F.linear(x, w, b), iters=5, warmup=2)
    print(f"{name:20s} {ms:8.3f} ms  {2 * M * K * N / ms / 1e9:7.1f} TF/s (nominal shape)")
    del x, w
