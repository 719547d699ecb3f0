"""Dev tool (GPU box): what a PLAIN copy reaches with the scan's access pattern — 256 work-groups (one per CU, four waves), each reading
640 bytes of every 20 KB row of x (its four heads) and writing the same piece of y, eight sequence segments — against the same bytes
laid out work-group-major (a work-group's rows back to back), and torch's own copy_.  Round 5 on one MI355X: token-major 5.50 TB/s
read-only / 4.69 TB/s read + write, work-group-major 5.37 / 4.81, torch copy_ 4.91: the layout does not matter, and a read + write
stream tops out at ~0.6 of the 8 TB/s peak — the scan's march moves its bytes at 88 % of that rate.
    python timeviper_amd/devtools/membench_rows.py"""
import ctypes
import subprocess
import tempfile
from pathlib import Path

import torch

src = Path(__file__).with_suffix(".hip")
so = Path(tempfile.gettempdir()) / "membench_rows.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-shared", str(src), "-o", str(so)], check=True)
lib = ctypes.CDLL(str(so))
lib.mb_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int]
L, nseg = 163840, 8
x = torch.empty(L * 20480 // 4, dtype=torch.int32, device="cuda").random_()
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


for rw in (0, 1):
    for layout in (0, 1):
        ms = min(t(lambda: lib.mb_launch(x.data_ptr(), y.data_ptr(), L // nseg, nseg, layout, 20480, rw)) for _ in range(3))
        by = L * 20480 * (2 if rw else 1)
        print(f"{'read + write' if rw else 'read only   '} {'work-group-major (rows of 640 B back to back)' if layout else 'token-major (640 B of every 20 KB row)      '}: "
              f"{ms * 1e3:6.0f} us  {by / ms / 1e6:5.0f} GB/s", flush=True)
ms = min(t(lambda: y.copy_(x)) for _ in range(3))
print(f"torch copy_ of the same 3.36 GB: {ms * 1e3:6.0f} us  {2 * x.numel() * 4 / ms / 1e6:5.0f} GB/s read + write")
