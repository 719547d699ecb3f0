"""Dev tool: the four SigLIP-so400m GEMM shapes (256-frame clip) on the BLAS back ends torch offers."""
import sys
import torch
import torch.nn.functional as F


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


M = 256 * 729
shapes = [("qkv", 1152, 3456), ("proj", 1152, 1152), ("fc1", 1152, 4304), ("fc2", 4304, 1152)]
for lib in (sys.argv[1:] or ["hipblaslt", "cublas"]):
    torch.backends.cuda.preferred_blas_library(lib)
    for name, K, N in shapes:
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
        b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
        wt = w.t().contiguous()
        ms = timeit(lambda: F.linear(x, w, b))
        ms2 = timeit(lambda: torch.addmm(b, x, wt))
        ms3 = timeit(lambda: F.linear(x, w))
        fl = 2 * M * K * N
        print(f"{lib:10s} {name:5s} linear+bias {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s | addmm(NN) {ms2:7.3f} ms {fl/ms2/1e9:7.1f} | no-bias {ms3:7.3f} ms {fl/ms3/1e9:7.1f}")
