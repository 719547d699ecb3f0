"""Dev tool (needs a build with TV_EXTRA_HIPCC_FLAGS="-DTV_HEAD_STAMP"): cycles per phase of a 64-token step of the
head-per-wave scan kernel (ssd_head.hip), waves 0 and 1 of work-group 0.
    python timeviper_amd/devtools/head_stamps.py [tokens] [nseg] [dt_std]"""
import ctypes
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from timeviper_amd import _capi, kernels as K  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 163940
    nseg = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dt_std = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02      # 1.3 = the synthetic 9B model's raw dt
    H, P, G, N = 128, 80, 8, 128
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
    x, dt, Bm, Cm = rn(1, L, H, P), (rn(1, L, H).float() * dt_std).bfloat16(), rn(1, G, L, N).transpose(1, 2), rn(1, G, L, N).transpose(1, 2)
    A = -(torch.rand(H, device=dev, generator=g) * 15 + 1)
    dtv = torch.exp(torch.rand(H, device=dev, generator=g) * (math.log(0.1) - math.log(1e-3)) + math.log(1e-3))
    D, bias = torch.ones(H, device=dev), dtv + torch.log(-torch.expm1(-dtv))
    K.ssd_scan_set_impl(6)
    for _ in range(3):
        K.mamba_chunk_scan_combined(x, dt, A, Bm, Cm, chunk_size=64, D=D, dt_bias=bias, dt_softplus=True)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    fn = ctypes.CDLL(str(Path(_capi.__file__).parent / "lib" / "libtimeviper_hip.so")).tv_ssd_head_debug_stamps
    fn.argtypes = [ctypes.c_void_p]
    assert fn(out) == 0
    steps = ((L + 63) // 64 + nseg - 1) // nseg
    names = ["barrier", "x~", "quarter 0 (+ y stores)", "quarter 1 (+ C.B^T, x copies)", "quarter 2 (+ B/C copies)", "quarter 3", "wait for C.B^T", "Ydiag", "prep", "y epilogue"]
    for w in range(2):
        tot = sum(out[16 * w + i] for i in range(10))
        print(f"wave {w}: {tot / steps:7.0f} cycles/step: " + "  ".join(f"{n} {out[16 * w + i] / steps:.0f}" for i, n in enumerate(names)))


if __name__ == "__main__":
    main()
