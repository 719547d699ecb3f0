"""ctypes binding of libtimeviper_hip.so — the C ABI declared in include/timeviper_hip.h.

This is the stub a reference maintainer would add: plain pointers
and sizes, no torch types cross the boundary.  There is NO fallback: if the
library is missing, `lib()` raises, and every operator in `kernels.py` fails
loudly rather than silently running eager PyTorch.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libtimeviper_hip.so"
_lib = None

TV_F32, TV_BF16, TV_F16 = 0, 1, 2

_p, _i, _l, _f, _z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/timeviper_hip.h one to one
SIGNATURES = {
    "tv_abi_version": (_i, []),
    "tv_last_error": (C.c_char_p, []),
    "tv_build_id": (C.c_char_p, []),
    "tv_causal_conv1d_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _l, _l, _l, _l, _i, _i, _p]),
    "tv_causal_conv1d_xbc_fwd": (_i, [_p] * 7 + [_i] * 6 + [_l, _l, _i, _i, _p]),
    "tv_ssd_cb_bytes": (_z, [_i] * 3),
    "tv_causal_conv1d_xbc_cb_fwd": (_i, [_p] * 8 + [_i] * 6 + [_l, _l, _i, _i, _p]),
    "tv_causal_conv1d_update": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "tv_rmsnorm_fwd": (_i, [_p, _p, _p, _p, _p, _l, _i, _l, _l, _l, _l, _f, _i, _i, _p]),
    "tv_layernorm_fwd": (_i, [_p] * 7 + [_l, _i, _l, _l, _l, _l, _f, _i, _p]),
    "tv_gelu_fwd": (_i, [_p, _p, _l, _i, _p]),
    "tv_relu2_fwd": (_i, [_p, _p, _l, _i, _p]),
    "tv_rmsnorm_gated_fwd": (_i, [_p, _p, _p, _p, _l, _i, _i, _l, _l, _l, _f, _i, _i, _p]),
    "tv_ssd_scan_workspace_bytes": (_z, [_i] * 7),
    "tv_ssd_scan_fwd": (_i, [_p] * 11 + [_i] * 6 + [_l] * 12 + [_i, _i, _f, _f, _i, _p, _z, _p]),
    "tv_ssd_scan_cb_fwd": (_i, [_p] * 12 + [_i] * 6 + [_l] * 12 + [_i, _i, _f, _f, _i, _p, _z, _p]),
    "tv_ssd_state_correction_workspace_bytes": (_z, [_i] * 3),
    "tv_ssd_state_correction": (_i, [_p] * 6 + [_i] * 6 + [_l] * 7 + [_i, _i, _f, _f, _i, _p, _z, _p]),
    "tv_ssd_scan_set_impl": (None, [_i]),
    "tv_ssd_scan_last_impl": (_i, []),
    "tv_ssd_head_set_asm": (None, [_i]),
    "tv_selective_state_update": (_i, [_p] * 9 + [_i] * 7 + [_p]),
    "tv_gemm_bf16_fwd": (_i, [_p, _p, _p, _p, _l, _i, _i, _l, _l, _l, _i, _i, _p]),
    "tv_gemm_set_persist": (None, [_i, _i]),
    "tv_gemm_set_drip": (None, [_i]),
    "tv_flash_attn_fwd": (_i, [_p] * 5 + [_i] * 6 + [_l] * 12 + [_f, _i, _i, _p]),
    "tv_flash_attn_set_variant": (None, [_i]),
    "tv_flash_attn_fp8_workspace_bytes": (_z, [_i] * 5),
    "tv_flash_attn_fp8_fwd": (_i, [_p] * 5 + [_i] * 6 + [_l] * 12 + [_f, _i, _i, _p, _z, _p]),
    "tv_gemv_bf16_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _p, _l, _p, _l, _p, _i, _f, _p, _l, _i,
                              _p, _p, _p, _i, _i, _p]),
    "tv_attn_decode_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "tv_attn_decode_fwd": (_i, [_p] * 5 + [_i, _i, _p, _i, _i, _i] + [_l] * 10 + [_f, _i, _p, _z, _p]),
    "tv_attn_rank_workspace_bytes": (_z, [_i, _i]),
    "tv_attn_rank_scores": (_i, [_p, _p, _p, _i, _i, _i, _i, _l, _l, _i, _i, _f, _i, _p, _z, _p]),
    "tv_attn_rank_logits": (_i, [_p, _p, _p, _i, _i, _i, _i, _l, _l, _f, _i, _p]),
    "tv_attn_rank_scores_from_logits": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "tv_gather_rows": (_i, [_p, _p, _p, _l, _i, _l, _l, _i, _p]),
    "tv_uniform_keep_indices": (_i, [_p, _l, _l, _l, _p]),
    "tv_dropped_indices": (_i, [_p, _l, _l, _l, _p, _p]),
    "tv_rope_fwd": (_i, [_p, _p, _p, _l, _i, _i, _l, _l, _i, _p]),
    "tv_silu_mul_fwd": (_i, [_p, _p, _p, _l, _i, _l, _l, _l, _i, _p]),
    "tv_tome_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "tv_tome_merge_round": (_i, [_p] * 4 + [_i] * 6 + [_p, C.c_size_t, _p]),
    "tv_patch_embed_workspace_bytes": (_z, [_i] * 3),
    "tv_patch_embed_fwd": (_i, [_p] * 5 + [_i] * 7 + [_p, _p]),
    "tv_patch_embed_strided_fwd": (_i, [_p] * 5 + [_i] * 7 + [_l] * 3 + [_i, _p, _p]),
}


class TimeViperHipError(RuntimeError):
    pass


ABI_VERSION = 12      # tv_abi_version() of the library these SIGNATURES describe (csrc/capi.cpp)


def lib_path() -> Path:
    return Path(os.environ.get("TIMEVIPER_HIP_LIB", str(_LIB_PATH)))


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise if it has not been built."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not path.exists():
            raise TimeViperHipError(
                f"{path} not found: the HIP extension has not been built. "
                "Run `python -m timeviper_amd.build` (or __graft_entry__.build())."
            )
        # torch first: it ships its own HIP runtime, and the one that is mapped first serves the
        # whole process — loading this library before torch binds it to /opt/rocm's copy and leaves
        # the two runtimes disagreeing about the devices ("no ROCm-capable device is detected")
        import torch  # noqa: F401
        handle = C.CDLL(str(path))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if handle.tv_abi_version() != ABI_VERSION:
            raise TimeViperHipError(f"{path} has ABI version {handle.tv_abi_version()}, this package binds "
                                    f"version {ABI_VERSION}: rebuild it (`python -m timeviper_amd.build --force`)")
        if "TIMEVIPER_HIP_LIB" not in os.environ:     # the in-tree library must match the in-tree sources
            from .build import source_id
            have, want = handle.tv_build_id().decode(), source_id()
            if have != want:
                raise TimeViperHipError(f"{path} was built from other sources (build id {have}, tree {want}): "
                                        "run `python -m timeviper_amd.build` (ensure_built() does it)")
        _lib = handle
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().tv_last_error().decode("utf-8", "replace")
        raise TimeViperHipError(f"{what} failed (status {status}): {msg}")
