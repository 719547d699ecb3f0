"""Dev tool (GPU box): read-only streams of weight-matrix-sized buffers (92 / 171 / 1 174 MB: out_proj, in_proj / up_proj mean,
lm_head of Nemotron-Nano-9B-v2), a different buffer every launch (no launch finds its bytes in the 256 MB Infinity Cache) — grid
sizes x loads in flight per lane x default / non-temporal: the practical roof of the decode step's matrix-vector products.
    python timeviper_amd/devtools/membench_read.py"""
import ctypes
import os
import subprocess
import tempfile
from pathlib import Path

import torch

src = Path(__file__).with_suffix(".hip")
so = Path(tempfile.gettempdir()) / "membench_read.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-shared", str(src), "-o", str(so)], check=True)
lib = ctypes.CDLL(str(so))
lib.mr_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device="cuda")

for mb in (92, 171, 1174):
    nbytes = mb * 1000 * 1000 // 4096 * 4096
    nbuf = max(2, 700_000_000 // nbytes + 1)
    bufs = [torch.empty(nbytes // 4, dtype=torch.int32, device="cuda").random_() for _ in range(nbuf)]
    best = (0.0, None)
    for nt in (0, 1):
        for unroll in (1, 2, 4, 8):
            for grid in (256, 512, 1024, 2048, 4096, 8192):
                st = torch.cuda.current_stream().cuda_stream
                k = [0]

                def go():
                    k[0] = (k[0] + 1) % nbuf
                    lib.mr_launch(bufs[k[0]].data_ptr(), sink.data_ptr(), nbytes // 16, grid, unroll, nt, st)
                for _ in range(3):
                    go()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 40 if mb < 1000 else 10
                e0.record()
                for _ in range(n):
                    go()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / n * 1e3
                gbs = nbytes / us / 1e3
                if os.environ.get("TV_MEMBENCH_ALL"):
                    print(f"  {mb} MB nt {nt} unroll {unroll} grid {grid:5d}: {us:6.1f} us {gbs:7.1f} GB/s")
                if gbs > best[0]:
                    best = (gbs, (nt, unroll, grid, us))
    print(f"{mb:5d} MB: best {best[0]:7.1f} GB/s ({best[0] / 8000:.3f} of 8 TB/s) at non-temporal {best[1][0]}, {best[1][1]} loads in flight, "
          f"grid {best[1][2]}: {best[1][3]:.1f} us a launch")
    del bufs
