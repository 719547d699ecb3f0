"""Dev tool: does folding the residual add into the GEMM (D = A.B^T + C, beta = 1) cost GEMM time?"""
import torch
import torch.nn.functional as F
from bench_gemm import timeit

M = 2048 * 729
for name, K, N in [("proj", 1152, 1152), ("fc2", 4352, 1152)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    out = torch.empty_like(x)
    t0 = timeit(lambda: F.linear(a, w, b), iters=5, warmup=2)
    t1 = timeit(lambda: torch.addmm(x, a, w.t()), iters=5, warmup=2)
    t2 = timeit(lambda: torch.addmm(x, a, w.t(), out=out), iters=5, warmup=2)
    t3 = timeit(lambda: x.addmm_(a, w.t()), iters=5, warmup=2)
    print(f"{name}: linear+bias {t0:.3f} ms | addmm(C=x) {t1:.3f} | addmm out= {t2:.3f} | addmm_ in place {t3:.3f}")
