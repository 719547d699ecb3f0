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


def main():
    VIT = [("qkv", 256 * 729, 1152, 3456, True), ("proj", 256 * 729, 1152, 1152, True),
           ("fc1", 256 * 729, 1152, 4304, True), ("fc2", 256 * 729, 4304, 1152, True)]
    LLM = [("in_proj", 32768, 4480, 22656, False), ("out_proj", 32768, 10240, 4480, False),
           ("mlp_up", 32768, 4480, 15680, False), ("mlp_down", 32768, 15680, 4480, False),
           ("attn_q", 32768, 4480, 5120, False), ("attn_o", 32768, 5120, 4480, False)]
    for lib in (sys.argv[1:] or ["hipblaslt", "cublas"]):
        torch.backends.cuda.preferred_blas_library(lib)
        for name, M, K, N, has_bias in VIT + LLM:
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
            b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
            wt = w.t().contiguous()
            ms = timeit(lambda: F.linear(x, w, b if has_bias else None))
            ms2 = timeit(lambda: torch.addmm(b, x, wt))
            ms3 = timeit(lambda: F.linear(x, w))
            fl = 2 * M * K * N
            print(f"{lib:10s} {name:5s} linear+bias {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s | addmm(NN) {ms2:7.3f} ms {fl/ms2/1e9:7.1f} | no-bias {ms3:7.3f} ms {fl/ms3/1e9:7.1f}")


if __name__ == "__main__":
    main()
