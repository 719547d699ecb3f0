"""Dev tool: what torch's TunableOp (a search over the hipBLASLt / rocBLAS solutions of a GEMM shape) buys on the library
GEMMs of the forward (GPU only).  Prints default against tuned time per shape and leaves the results file behind.
    python timeviper_amd/devtools/tune_lib_gemm.py [--frames 2048] [--out gpurun_out/tunableop.csv]"""
import argparse
import os
import sys
import time
from pathlib import Path

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
    ap.add_argument("--out", default="gpurun_out/tunableop.csv")
    a = ap.parse_args()
    dev = "cuda"
    M = a.frames * 729
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).bfloat16()
    cases = []
    for name, N, Kd, kind in (("qkv", 3584, 1152, "linear"), ("proj", 1152, 1152, "addmm"), ("fc2", 1152, 4352, "addmm"),
                              ("fc1", 4352, 1152, "linear")):
        x, w = rn(M, Kd), rn(N, Kd, sc=0.02)
        if kind == "linear":
            b = rn(N, sc=0.1)
            fn = (lambda x=x, w=w, b=b: F.linear(x, w, b))
        else:
            res = rn(M, N)
            fn = (lambda x=x, w=w, res=res: torch.addmm(res, x, w.t(), out=res))
        cases.append((name, N, Kd, fn))
    base = {n: timeit(fn) for n, _, _, fn in cases}
    tun = torch.cuda.tunable
    tun.enable(True)
    tun.tuning_enable(True)
    tun.set_filename(a.out)
    try:
        tun.set_max_tuning_duration(3000)
        tun.set_max_tuning_iterations(20)
    except Exception as e:          # noqa: BLE001
        print("tunable limits:", e)
    for n, N, Kd, fn in cases:
        t0 = time.time()
        fn()
        torch.cuda.synchronize()
        dt = time.time() - t0
        t = timeit(fn)
        fl = 2.0 * M * N * Kd
        print(f"{n:5s} N {N} K {Kd}: default {base[n]:7.3f} ms = {fl / base[n] / 1e9:7.1f} TFLOP/s | tuned {t:7.3f} ms = {fl / t / 1e9:7.1f} "
              f"(tuning took {dt:.1f} s)", flush=True)
    tun.write_file()
    print(Path(a.out).read_text()[:3000])


if __name__ == "__main__":
    main()
