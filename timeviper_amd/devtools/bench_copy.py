"""Dev tool: practical HBM roof of the box — torch's device copy / in-place scale on GELU-sized buffers."""
import torch
from bench_gemm import timeit

for gb in (1.6, 6.4):
    n = int(gb * 2**30 / 2)
    x = torch.randn(n, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x), iters=10)
    t2 = timeit(lambda: x.mul_(1.0001), iters=10)
    print(f"{gb} GiB: copy {2 * n * 2 / t / 1e6:.0f} GB/s   in-place mul {2 * n * 2 / t2 / 1e6:.0f} GB/s")
