"""Dev tool: the hand-written bf16 GEMM with fused epilogues (csrc/gemm.hip) against torch (hipBLASLt) + the separate
element-wise pass, at the SigLIP-so400m shapes of one 2 048-frame ViT call (GPU only).
    python timeviper_amd/devtools/bench_gemm_fused.py [--frames 2048] [--which fc1,qkv,proj,fc2]"""
import argparse
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from timeviper_amd import kernels as K  # noqa: E402


def timeit(fn, iters=5, warmup=2):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--which", default="fc1,qkv,proj,fc2")
    ap.add_argument("--qkv-n", type=int, default=3456, help="3 456 as stored, 3 584 zero-padded to the 256-wide tile")
    a = ap.parse_args()
    dev = "cuda"
    M = a.frames * 729
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).bfloat16()
    for name in a.which.split(","):
        if name == "fc1":
            N, Kd = 4352, 1152
            x, w, b = rn(M, Kd), rn(N, Kd, sc=0.02), rn(N, sc=0.1)
            ours = lambda: K.linear_fused(x, w, b, epilogue=K.GEMM_BIAS_GELU)
            ref = lambda: K.gelu(F.linear(x, w, b), inplace=True)
            plain = lambda: F.linear(x, w, b)
        elif name == "qkv":
            N, Kd = a.qkv_n, 1152
            x, w, b = rn(M, Kd), rn(N, Kd, sc=0.02), rn(N, sc=0.1)
            ours = lambda: K.linear_fused(x, w, b, epilogue=K.GEMM_BIAS)
            ref = plain = lambda: F.linear(x, w, b)
        elif name.startswith("b:"):            # plain bias epilogue on an arbitrary shape: b:N:K
            N, Kd = (int(v) for v in name[2:].split(":"))
            x, w, b = rn(M, Kd), rn(N, Kd, sc=0.02), rn(N, sc=0.1)
            ours = lambda: K.linear_fused(x, w, b, epilogue=K.GEMM_BIAS)
            ref = plain = lambda: F.linear(x, w, b)
        elif name.startswith("a:"):            # accumulate epilogue on an arbitrary shape: a:N:K
            N, Kd = (int(v) for v in name[2:].split(":"))
            x, w = rn(M, Kd), rn(N, Kd, sc=0.02)
            res = rn(M, N)
            ours = lambda: K.linear_fused(x, w, None, epilogue=K.GEMM_ACCUM, out=res)
            ref = plain = lambda: torch.addmm(res, x, w.t(), out=res)
        else:
            N, Kd = 1152, (1152 if name == "proj" else 4352)
            x, w = rn(M, Kd), rn(N, Kd, sc=0.02)
            res = rn(M, N)
            ours = lambda: K.linear_fused(x, w, None, epilogue=K.GEMM_ACCUM, out=res)
            ref = plain = lambda: torch.addmm(res, x, w.t(), out=res)
        fl = 2.0 * M * N * Kd
        K.gemm_set_persist(0, 0)
        t_t = timeit(ours)                      # one work-group per tile (csrc/gemm.hip)
        K.gemm_set_persist(-1, 0)
        t_o, t_r = timeit(ours), timeit(ref)    # automatic: the persistent kernel at these sizes
        t_p = timeit(plain) if plain is not ref else t_r
        # correctness spot check on the first rows
        if name in ("fc1", "qkv") or name.startswith("b:"):
            o, r = ours()[:512].float(), ref()[:512].float()
            err = float((o - r).abs().max())
        else:
            err = float("nan")
        print(f"{name:5s} M {M} N {N} K {Kd}: persistent {t_o:8.3f} ms = {fl / t_o / 1e9:7.1f} TFLOP/s | per-tile {t_t:8.3f} ms = "
              f"{fl / t_t / 1e9:7.1f} | torch {t_r:8.3f} ms = "
              f"{fl / t_r / 1e9:7.1f} TFLOP/s (GEMM alone {t_p:8.3f} ms = {fl / t_p / 1e9:7.1f}) | max |diff| {err:.3g}", flush=True)
        del x, w


if __name__ == "__main__":
    main()
