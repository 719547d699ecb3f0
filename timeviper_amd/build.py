"""Build libtimeviper_hip.so (gfx950) from timeviper_amd/csrc with plain hipcc.

The library is a C-ABI shared object (include/timeviper_hip.h); it does not link
against torch.  Objects are cached by CONTENT: every object carries a sidecar with the
hash of its source, the shared headers and the compiler flags, and the library embeds the
hash of the whole source set (`tv_build_id()`), so a kernel edit can never be benchmarked
through a stale binary — not even on a box whose file mtimes were reset by a copy.

    python -m timeviper_amd.build [--force] [--jobs N]
"""
from __future__ import annotations

import argparse
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ = ROOT / "lib" / "obj"
LIB = ROOT / "lib" / "libtimeviper_hip.so"
LIB_ID = ROOT / "lib" / "libtimeviper_hip.id"
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-Wno-unused-result", "-DNDEBUG"]
if os.environ.get("TV_MARCH_ABLATE"):   # dev only: compile the scan kernel's ablation switches in
    FLAGS.append("-DTV_MARCH_ABLATE")
if os.environ.get("TV_FA_STAMP"):       # dev only: per-phase stamps in the streaming attention kernel
    FLAGS.append("-DTV_FA_STAMP")
if os.environ.get("TV_DRIP_STAMP"):     # dev only: per-phase cycle sums in the 256 x 192 GEMM (gemm_drip.hip)
    FLAGS.append("-DTV_DRIP_STAMP")
if os.environ.get("TV_SLICE_STAMP"):    # dev only: per-wave barrier-wait stamps in ssd_slice.hip
    FLAGS.append("-DTV_SLICE_STAMP")


FLAGS += os.environ.get("TV_EXTRA_HIPCC_FLAGS", "").split()   # dev only

# per-file flags.  gemm_drip.hip names its accumulator registers (a[0:119], csrc/gemm_drip_regs.inc) instead of handing
# them to the register allocator: the compiler must then never park a spilled vector register in an accumulator register.
FILE_FLAGS = {"gemm_drip.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]}


def _sources():
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _headers():
    incs = [p for p in CSRC.glob("*.inc") if not p.name.endswith("_stamp.inc")]      # generated asm bodies (devtools/gen_head_step.py)
    return sorted(list(CSRC.glob("*.hpp")) + incs + list((ROOT.parent / "include").glob("*.h")))


def _hash(paths, extra: str = "") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()[:16]


def source_id() -> str:
    """Hash of every source, header and compiler flag the library is built from."""
    return _hash(_sources() + _headers(), " ".join(FLAGS) + repr(sorted(FILE_FLAGS.items())))


def _compile(src: Path, force: bool, build_id: str) -> Path:
    obj = OBJ / (src.name + ".o")
    tag = OBJ / (src.name + ".hash")
    flags = list(FLAGS) + FILE_FLAGS.get(src.name, [])
    if src.name == "capi.cpp":                    # the one translation unit that carries the id
        flags.append(f'-DTV_BUILD_ID="{build_id}"')
    want = _hash([src] + _headers(), " ".join(flags))
    if not force and obj.exists() and tag.exists() and tag.read_text() == want:
        return obj
    cmd = [HIPCC, *flags, "-x", "hip", "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    tag.write_text(want)
    return obj


def build(force: bool = False, jobs: int = 4) -> Path:
    OBJ.mkdir(parents=True, exist_ok=True)
    srcs = _sources()
    bid = source_id()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, bid), srcs))
    if force or not LIB.exists() or not LIB_ID.exists() or LIB_ID.read_text() != bid:
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        LIB_ID.write_text(bid)
    return LIB


def ensure_built() -> Path:
    """Make `lib/libtimeviper_hip.so` match the sources in the tree: a no-op when the library's
    recorded build id equals the hash of csrc/ + include/ (+ flags), else an incremental rebuild.
    This is a BUILD step, not a fallback: without a working hipcc it raises, and the operators
    keep failing loudly.  (`_capi.lib()` checks `tv_build_id()` of the loaded library against the
    same hash, so a stale binary cannot be used silently either.)"""
    if LIB.exists() and LIB_ID.exists() and LIB_ID.read_text() == source_id():
        return LIB
    return build()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    a = ap.parse_args()
    print(build(a.force, a.jobs))
