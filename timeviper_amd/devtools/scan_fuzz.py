"""Dev tool / test helper: the generated step (csrc/ssd_head_step.inc) against the C++ step of the head-per-wave march, bit for
bit, on RANDOM shapes, lengths (incl. 64 k, 64 k +- 1, shorter than a chunk), decay regimes (A over five decades, dt mean /
spread, time-correlated heads), options (softplus, dt_limit, D, dt_bias, initial states, x | B | C as strided slices of one
packed row, both group maps) — every combination goes through `mamba_chunk_scan_combined` twice (`ssd_head_set_asm(1 / 0)`).
usage: python timeviper_amd/devtools/scan_fuzz.py [ncases=120] [seed=0]      (tests/test_ops_gpu.py runs 48 cases)"""
import math
import random
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from timeviper_amd import kernels as K  # noqa: E402

DEV = "cuda"


def run(n, seed, verbose=False):
    """-> list of descriptions of the cases whose y / final state / decay total differ between the two steps"""
    rnd = random.Random(seed)
    bad = []
    for case in range(n):
        B = rnd.choice([1, 1, 1, 2])
        G = rnd.choice([1, 2, 4, 8])
        H = G * 4 * rnd.choice([1, 1, 2, 4])
        if H > 64:
            H = G * 4
        Lk = rnd.choice(["tiny", "short", "mid", "long", "exact", "plus1", "minus1"])
        L = {"tiny": rnd.randint(1, 130), "short": rnd.randint(131, 1500), "mid": rnd.randint(1500, 9000),
             "long": rnd.randint(9000, 40000), "exact": 64 * rnd.randint(2, 300), "plus1": 64 * rnd.randint(2, 300) + 1,
             "minus1": 64 * rnd.randint(2, 300) - 1}[Lk]
        if B * H * L > 64 * 40000:
            L = max(65, 64 * 40000 // (B * H))
        a_lo = 10 ** rnd.uniform(-3, 2)
        a_hi = a_lo * 10 ** rnd.uniform(0, 1.5)
        dt_mean = rnd.uniform(-4, 2)
        dt_std = rnd.choice([0.02, 0.3, 1.3, 3.0])
        softplus = rnd.random() < 0.8
        limit = rnd.choice([(0.0, float("inf")), (0.0, float("inf")), (1e-3, 0.1), (0.0, 2.0)])
        use_D = rnd.random() < 0.7
        use_bias = rnd.random() < 0.6
        use_init = rnd.random() < 0.5
        strided = rnd.random() < 0.5
        gmap = rnd.choice(["block", "block", "tile"])
        g = torch.Generator().manual_seed(1000 * seed + case)
        P, N = 80, 128
        if strided:                                   # x | B | C as slices of one packed row, like the mixer's conv output
            W = H * P + 2 * G * N
            xbc = torch.randn(B, L, W, generator=g).to(torch.bfloat16).to(DEV)
            x = xbc[..., :H * P].view(B, L, H, P)
            Bm = xbc[..., H * P:H * P + G * N].view(B, L, G, N)
            Cm = xbc[..., H * P + G * N:].view(B, L, G, N)
        else:
            x = torch.randn(B, L, H, P, generator=g).to(torch.bfloat16).to(DEV)
            Bm = (torch.randn(B, L, G, N, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
            Cm = (torch.randn(B, L, G, N, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
        dt = torch.randn(B, L, H, generator=g) * dt_std + dt_mean
        if not softplus:
            dt = dt.abs() * 0.05
        if rnd.random() < 0.3:                        # correlated in time: persistently slow / fast heads
            dt = dt + torch.randn(1, 1, H, generator=g) * 2
            if not softplus:
                dt = dt.abs() * 0.05
        dt = dt.to(torch.bfloat16).to(DEV)
        A = -(torch.rand(H, generator=g) * (a_hi - a_lo) + a_lo).to(DEV)
        D = (torch.rand(H, generator=g) + 0.5).to(DEV) if use_D else None
        bias = (torch.randn(H, generator=g)).to(DEV) if use_bias else None
        init = torch.randn(B, H, P, N, generator=g).to(DEV) if use_init else None
        outs = []
        K.ssd_scan_set_impl(6)
        try:
            for on in (1, 0):
                K.ssd_head_set_asm(on)
                y, fin, dec = K.mamba_chunk_scan_combined(x, dt, A, Bm, Cm, chunk_size=64, D=D, dt_bias=bias, dt_softplus=softplus,
                                                          dt_limit=limit, initial_states=init, return_final_states=True,
                                                          return_total_decay=True, group_map=gmap)
                impl = K.ssd_scan_last_impl()
                assert impl == 6, f"the fuzz case ran on scan implementation {impl}, not on the head march (6)"
                outs.append((y.clone(), fin.clone(), dec.clone()))
        finally:
            K.ssd_head_set_asm(-1)
            K.ssd_scan_set_impl(0)
        (y1, f1, d1), (y0, f0, d0) = outs
        ok = torch.equal(y1, y0) and torch.equal(f1, f0) and torch.equal(d1, d0) and bool(torch.isfinite(y1.float()).all())
        tag = f"case {case}: B{B} L{L} H{H} G{G} A[{a_lo:.3g},{a_hi:.3g}] dt({dt_mean:.2f},{dt_std}) sp{int(softplus)} lim{limit} D{int(use_D)} b{int(use_bias)} init{int(use_init)} str{int(strided)} {gmap} impl{impl}"
        if not ok:
            dy = (y1.float() - y0.float()).abs().max().item()
            df = (f1 - f0).abs().max().item()
            bad.append(f"{tag} dy {dy} dstate {df}")
            print("MISMATCH", bad[-1], flush=True)
        elif verbose and case % 10 == 0:
            print("ok", tag, flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    bad = run(n, seed, verbose=True)
    print(f"{n} cases (seed {seed}), {len(bad)} mismatches, {time.time() - t0:.1f} s")
    sys.exit(1 if bad else 0)
