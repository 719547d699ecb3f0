"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header
declares, and the ctypes table mirrors the header one to one.  No compute calls."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def header_symbols():
    txt = (ROOT / "include" / "timeviper_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tv_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_header_symbols():
    from timeviper_amd import build, _capi
    build.build()
    lib = _capi.lib()
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/timeviper_hip.h but not exported"
    assert set(_capi.SIGNATURES) == set(syms)
    src = (ROOT / "timeviper_amd" / "csrc" / "capi.cpp").read_text()
    assert lib.tv_abi_version() == _capi.ABI_VERSION == int(re.search(r"tv_abi_version\(void\) \{ return (\d+); \}", src).group(1))
    # the binary carries the hash of the sources it was built from: a kernel edit cannot run through a stale library
    assert lib.tv_build_id().decode() == build.source_id()


def test_bad_arguments_fail_loudly_without_gpu():
    from timeviper_amd import _capi
    lib = _capi.lib()
    # null pointers are rejected before any launch
    st = lib.tv_gather_rows(None, None, None, 4, 8, 8, 8, 1, None)  # 4 rows, null pointers
    assert st == -1 and b"null" in lib.tv_last_error()
    st = lib.tv_causal_conv1d_fwd(1, 1, None, None, 1, 1, 8, 8, 7, 8, 8, 8, 8, 1, 1, None)
    assert st == -2  # kernel width 7 unsupported


def test_product_path_has_no_cpu_fallback():
    import torch
    from timeviper_amd import kernels
    from timeviper_amd._capi import TimeViperHipError
    x = torch.randn(2, 8)
    with pytest.raises(TimeViperHipError):
        kernels.rms_norm(x, torch.ones(8), 1e-5)
    with pytest.raises(TimeViperHipError):
        kernels.tome_merge_round(torch.randn(1, 8, 32), None, 2, 16)
    # and nothing under timeviper_amd/ imports the oracle
    for p in (ROOT / "timeviper_amd").rglob("*.py"):
        assert "oracle" not in p.read_text().replace("# oracle", ""), p
