"""Dev tool: the 256 x 192 persistent GEMM (csrc/gemm_drip.hip) against the 256 x 256 kernels (gemm_persist.hip / gemm.hip) and
torch (hipBLASLt) at the SigLIP-so400m shapes of one 2 048-frame ViT call, with a correctness screen first (GPU only).
    python timeviper_amd/devtools/bench_gemm_drip.py [--frames 2048] [--which fc1,qkv,proj,fc2] [--check] [--iters 5]"""
import argparse
import math
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


def check():
    dev = "cuda"
    bad = 0
    shapes = [(1024, 1152, 1152, 8), (2187, 4352, 1152, 8), (1500, 1160, 1152, 16), (1300, 1152, 4352, 8),
              (729 * 8, 3456, 1408, 64), (256, 192, 1152, 8), (4096, 1152, 640, 8), (729 * 16, 1152, 1152, 64)]
    for (M, N, Kd, grid) in shapes:
        for epi in (0, 1, 2):
            g = torch.Generator().manual_seed(M + N + Kd + epi)
            a = (torch.randn(M, Kd, generator=g) * 0.5).bfloat16().to(dev)
            w = (torch.randn(N, Kd, generator=g) * (1.0 / math.sqrt(Kd))).bfloat16().to(dev)
            b = torch.randn(N, generator=g).bfloat16().float().to(dev)
            c0 = torch.randn(M, N, generator=g).bfloat16().to(dev)

            def run():
                if epi == 2:
                    return K.linear_fused(a, w, None, epilogue=2, out=c0.clone())
                return K.linear_fused(a, w, b, epilogue=epi)
            try:
                K.gemm_set_persist(1, grid)
                K.gemm_set_drip(1)
                outs = [run() for _ in range(3)]
                torch.cuda.synchronize()
                K.gemm_set_drip(0)
                K.gemm_set_persist(0, 0)
                tile = run()
                torch.cuda.synchronize()
            finally:
                K.gemm_set_persist(-1, 0)
                K.gemm_set_drip(-1)
            same = all(torch.equal(o, outs[0]) for o in outs[1:])
            nd = int((outs[0] != tile).sum())
            md = float((outs[0].float() - tile.float()).abs().max())
            # accumulating epilogue: same order of additions; bias epilogues: the bias enters first, a few final roundings flip
            ok = same and (nd == 0 if epi == 2 else (nd < 2e-3 * tile.numel() and md <= 2.0 ** -6 * max(1.0, float(tile.float().abs().max()))))
            bad += not ok
            print(f"check M {M} N {N} K {Kd} grid {grid} epi {epi}: repeat-identical {same}, differing elements vs per-tile {nd} "
                  f"(max |diff| {md:.3g}) {'ok' if ok else 'FAIL'}", flush=True)
            if nd and not ok:
                d = (outs[0] != tile).nonzero()
                rows, cols = d[:, 0], d[:, 1]
                print(f"   rows {int(rows.min())}..{int(rows.max())} cols {int(cols.min())}..{int(cols.max())}; first {d[:6].tolist()}")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--which", default="fc1,qkv,proj,fc2")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    if a.check:
        bad = check()
        print("check:", "all ok" if not bad else f"{bad} FAILED", flush=True)
    dev = "cuda"
    M = a.frames * 729
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).bfloat16()
    for name in [n for n in a.which.split(",") if n]:
        if name == "fc1":
            N, Kd, epi = 4352, 1152, K.GEMM_BIAS_GELU
        elif name == "qkv":
            N, Kd, epi = 3456, 1152, K.GEMM_BIAS
        elif name == "proj":
            N, Kd, epi = 1152, 1152, K.GEMM_ACCUM
        else:
            N, Kd, epi = 1152, 4352, K.GEMM_ACCUM
        x, w = rn(M, Kd), rn(N, Kd, sc=0.02)
        if epi == K.GEMM_ACCUM:
            res = rn(M, N)
            ours = lambda: K.linear_fused(x, w, None, epilogue=K.GEMM_ACCUM, out=res)
            ref = lambda: torch.addmm(res, x, w.t(), out=res)
        else:
            b = rn(N, sc=0.1).float()
            bb = b.bfloat16()
            ours = lambda: K.linear_fused(x, w, b, epilogue=epi)
            ref = (lambda: K.gelu(F.linear(x, w, bb), inplace=True)) if epi == K.GEMM_BIAS_GELU else (lambda: F.linear(x, w, bb))
        fl = 2.0 * M * N * Kd
        ts = {}
        for rnd in range(2):            # interleaved rounds: drip, 256-wide kernels, library
            K.gemm_set_drip(-1)
            ts.setdefault("drip", []).append(timeit(ours, a.iters))
            K.gemm_set_drip(0)
            ts.setdefault("256", []).append(timeit(ours, a.iters))
            K.gemm_set_drip(-1)
            ts.setdefault("lib", []).append(timeit(ref, a.iters))
        line = f"{name:5s} M {M} N {N} K {Kd}:"
        for k, v in ts.items():
            t = min(v)
            line += f"  {k} {t:7.3f} ms = {fl / t / 1e9:7.1f} TFLOP/s"
        print(line, flush=True)
        del x, w


if __name__ == "__main__":
    main()
