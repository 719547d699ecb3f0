"""Dev tool: does TunableOp find faster hipBLASLt / rocBLAS solutions than the default heuristic for the
shapes of the default run (2 048-frame ViT clips with padded N / K, LLM at 163 940 tokens)?"""
import os
import sys
import torch
import torch.nn.functional as F
from bench_gemm import timeit

shapes = [("vit qkv", 2048 * 729, 1152, 3584, True), ("vit proj", 2048 * 729, 1152, 1152, True),
          ("vit fc1", 2048 * 729, 1152, 4352, True), ("vit fc2", 2048 * 729, 4352, 1152, True),
          ("in_proj", 163940, 4480, 22656, False), ("mlp_up", 163940, 4480, 15680, False),
          ("mlp_down", 163940, 15680, 4480, False), ("out_proj", 163940, 10240, 4480, False)]
base = {}
for name, M, K, N, hb in shapes:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16) if hb else None
    base[name] = timeit(lambda: F.linear(x, w, b), iters=5, warmup=2)
    del x, w
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(30)
torch.cuda.tunable.set_max_tuning_iterations(5)
torch.cuda.tunable.set_filename(sys.argv[1] if len(sys.argv) > 1 else "/tmp/tunableop.csv")
for name, M, K, N, hb in shapes:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16) if hb else None
    F.linear(x, w, b)                       # tunes
    t = timeit(lambda: F.linear(x, w, b), iters=5, warmup=2)
    print(f"{name:9s} default {base[name]:8.3f} ms   tuned {t:8.3f} ms   ({100 * (base[name] - t) / base[name]:+.1f} %)", flush=True)
    del x, w
getattr(torch.cuda.tunable, "write_file", lambda: None)()
