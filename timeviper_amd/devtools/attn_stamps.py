"""Dev tool (needs a TV_FA_STAMP=1 build): where waves 0 and 4 of work-group 0 of the streaming attention
kernel spend their cycles, per key tile of a query block.   python timeviper_amd/devtools/attn_stamps.py"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from timeviper_amd import _capi, kernels as K  # noqa: E402

B, L, H, D = 256, 729, 16, 72
qkv = torch.randn(B, L, 3, H, D, device="cuda", dtype=torch.bfloat16)
q, k, v = qkv.unbind(2)
K.flash_attn_set_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
K.flash_attn_func(q, k, v)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 256)()
import os
fn = ctypes.CDLL(os.environ.get("TIMEVIPER_HIP_LIB", str(Path(_capi.__file__).parent / "lib" / "libtimeviper_hip.so"))).tv_fa_debug_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(out) == 0
tasks = (B * H * 3) // 256
names = ["K reads + QK^T + copies", "softmax", "PV", "vmcnt wait", "barrier", "epilogue + next block"]
for w in range(2):
    print(f"wave {4 * w}: ticks per tile (100 MHz s_memtime: 1 tick = 10 ns), mean over {tasks} query blocks")
    tot = 0
    for kt in list(range(8)) + [15]:
        row = [out[(w * 16 + kt) * 8 + p] / tasks for p in range(6)]
        tot += sum(row)
        print(f"  kt {kt:2d}: " + "  ".join(f"{n} {x:7.1f}" for n, x in zip(names, row) if x > 0))
    print(f"  total {tot:.1f} ticks per query block")
