"""Dev tool: do hipBLASLt's ViT GEMMs run faster with tile-aligned N / K (zero-padded weights)?"""
import torch
import torch.nn.functional as F
from bench_gemm import timeit

M = 163940
for name, K, N in [("in_proj N=22656", 4480, 22656), ("in_proj N=22784", 4480, 22784),
                   ("mlp_up N=15680", 4480, 15680), ("mlp_up N=15872", 4480, 15872),
                   ("mlp_down K=15680", 15680, 4480), ("mlp_down K=15872", 15872, 4480),
                   ("out_proj N=4480", 10240, 4480), ("out_proj N=4608", 10240, 4608),
                   ("K=4480 N=5120", 4480, 5120), ("K=4608 N=5120", 4608, 5120)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: F.linear(x, w, b), iters=5, warmup=2)
    print(f"{name:20s} {ms:8.3f} ms  {2 * M * K * N / ms / 1e9:7.1f} TF/s (nominal shape)")
    del x, w
