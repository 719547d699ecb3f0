"""Dev tool (GPU box): a plain 16-byte-per-lane copy of 3.36 GB (the size of the scan's x at 163 840 tokens) at full occupancy —
grid sizes x loads in flight per lane x default / non-temporal — against torch's copy_: the practical roof of a read + write stream
on THIS box (MI355X_MICROARCH.md records 6.29 TB/s for a tuned float4 copy; bytes = read + written).
    python timeviper_amd/devtools/membench_copy.py"""
import ctypes
import subprocess
import tempfile
from pathlib import Path

import torch

src = Path(__file__).with_suffix(".hip")
so = Path(tempfile.gettempdir()) / "membench_copy.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-shared", str(src), "-o", str(so)], check=True)
lib = ctypes.CDLL(str(so))
lib.mc_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int]
nbytes = 163840 * 20480
x = torch.empty(nbytes // 4, dtype=torch.int32, device="cuda").random_()
y = torch.empty_like(x)


def t(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


best = (0.0, None)
for nt in (0, 1):
    for unroll in (1, 2, 4, 8):
        for grid in (1024, 2048, 4096, 8192, 16384, 65536):
            ms = min(t(lambda: lib.mc_launch(x.data_ptr(), y.data_ptr(), nbytes // 16, grid, unroll, nt)) for _ in range(2))
            gbs = 2 * nbytes / ms / 1e6
            best = max(best, (gbs, (nt, unroll, grid)))
            print(f"nt {nt} loads in flight {unroll} grid {grid:6d}: {ms * 1e3:6.0f} us  {gbs:5.0f} GB/s read + write", flush=True)
assert torch.equal(x, y)
ms = min(t(lambda: y.copy_(x)) for _ in range(3))
print(f"torch copy_: {ms * 1e3:6.0f} us  {2 * nbytes / ms / 1e6:5.0f} GB/s read + write")
print(f"best: {best[0]:.0f} GB/s at (nt, loads in flight, grid) = {best[1]}")
