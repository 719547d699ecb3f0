"""Dev tool: the library GEMMs of the forward (hipBLASLt through torch) with and without PyTorch's TunableOp picking the
solution per shape.  usage (GPU box): python timeviper_amd/devtools/tune_gemm.py [--frames 2048] [--tokens 163940] [--out FILE]
Prints ms / TFLOP/s for the default heuristic and for the tuned solution; writes the tuned table to --out."""
import argparse
import sys

import torch
import torch.nn.functional as F


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
    ap.add_argument("--tokens", type=int, default=163940)
    ap.add_argument("--out", default="gpurun_out/tunableop_results.csv")
    a = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).bfloat16()
    Mv, Ml = a.frames * 729, a.tokens
    shapes = [("vit qkv (bias)", Mv, 3456, 1152, "bias"), ("vit proj (addmm)", Mv, 1152, 1152, "addmm"),
              ("vit fc2 (addmm)", Mv, 1152, 4304, "addmm"), ("vit fc1 (bias)", Mv, 4304, 1152, "bias"),
              ("llm in_proj", Ml, 22656, 4480, "plain"), ("llm out_proj", Ml, 4480, 10240, "plain"),
              ("llm up_proj", Ml, 15680, 4480, "plain"), ("llm down_proj", Ml, 4480, 15680, "plain")]
    import torch.cuda.tunable as tun
    res = {}
    for phase in ("default", "tuned"):
        if phase == "tuned":
            tun.enable(True)
            tun.tuning_enable(True)
            tun.set_max_tuning_iterations(20)
            tun.set_max_tuning_duration(200)
            tun.set_filename(a.out)
        for name, M, N, Kd, kind in shapes:
            x, w = rn(M, Kd), rn(N, Kd, sc=0.02)
            if kind == "bias":
                b = rn(N, sc=0.1)
                fn = lambda: F.linear(x, w, b)
            elif kind == "addmm":
                r = rn(M, N)
                fn = lambda: torch.addmm(r, x, w.t(), out=r)
            else:
                fn = lambda: F.linear(x, w)
            t = timeit(fn)
            res[(name, phase)] = t
            print(f"{phase:8s} {name:18s} M {M:8d} N {N:6d} K {Kd:6d}: {t:8.3f} ms = {2.0 * M * N * Kd / t / 1e9:7.1f} TFLOP/s", flush=True)
            del x, w
            torch.cuda.empty_cache()
    tun.write_file()
    for name, *_ in shapes:
        d, t = res[(name, "default")], res[(name, "tuned")]
        print(f"{name:18s}: tuned / default = {t / d:.3f}")


if __name__ == "__main__":
    main()
