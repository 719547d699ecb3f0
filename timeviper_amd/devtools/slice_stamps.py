"""Dev tool (needs a TV_SLICE_STAMP=1 build): per-wave barrier-wait share of the slice-march
scan kernel's workgroup 0.   python timeviper_amd/devtools/slice_stamps.py [tokens]"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from timeviper_amd import _capi, kernels as K  # noqa: E402

ROLE3 = {0: "slice0", 1: "slice1", 2: "slice2", 3: "xio", 4: "bc0+sc", 5: "bc1+sc", 6: "bc2+sc", 11: "bc3+sc",
         7: "mask0", 8: "mask1", 9: "prep", 10: "scale4"}
ROLE4 = {0: "slice0 (2 tiles)", 1: "slice1 (2 tiles)", 2: "slice2 (1 tile)", 3: "x/dt copies + prep", 4: "bc0 + y", 5: "bc1 + y",
         6: "mask0", 7: "mask1"}


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 163940
    impl = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ROLE = ROLE4 if impl == 4 else ROLE3
    H, P, G, N = 128, 80, 8, 128
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
    x, dt, Bm, Cm = rn(1, L, H, P), rn(1, L, H), rn(1, G, L, N).transpose(1, 2), rn(1, G, L, N).transpose(1, 2)
    A = -(torch.rand(H, device=dev, generator=g) * 15 + 1)
    D, bias = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    K.ssd_scan_set_impl(impl)
    for _ in range(2):
        K.mamba_chunk_scan_combined(x, dt, A, Bm, Cm, chunk_size=64, D=D, dt_bias=bias, dt_softplus=True)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 40)()
    fn = _capi.lib()._lib.tv_ssd_slice_debug_stamps if hasattr(_capi.lib(), "_lib") else None
    if fn is None:
        fn = ctypes.CDLL(str(Path(_capi.__file__).parent / "lib" / "libtimeviper_hip.so")).tv_ssd_slice_debug_stamps
    fn.argtypes = [ctypes.c_void_p]
    assert fn(out) == 0
    steps = (L + 63) // 64
    if impl == 4:
        steps = (steps + 1) // 2          # two concurrent segments at 128 heads
    for w in range(8 if impl == 4 else 12):
        wait, total = out[w], out[16 + w]
        print(f"wave {w:2d} {ROLE.get(w, '?'):12s} total {total/steps:8.0f} ticks/step   busy {(total-wait)/steps:8.0f}   "
              f"parked {wait/total:6.1%}")


    names = ["y tile writes (prev. chunk)", "quarter 0", "quarter 1", "quarter 2", "quarter 3", "Ydiag + epilogue", "first reads issued", "barrier + loop"]
    print("slice-wave 0 phases (cycles/step): " + "  ".join(f"{n} {out[32 + i] / steps:.0f}" for i, n in enumerate(names) if n != "-"))


if __name__ == "__main__":
    main()
