"""Build libtimeviper_hip.so (gfx950) from timeviper_amd/csrc with plain hipcc.

The library is a C-ABI shared object (include/timeviper_hip.h); it does not link
against torch.  Objects are cached by source mtime so rebuilds take seconds.

    python -m timeviper_amd.build [--force] [--jobs N]
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ = ROOT / "lib" / "obj"
LIB = ROOT / "lib" / "libtimeviper_hip.so"
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-Wno-unused-result", "-DNDEBUG"]
if os.environ.get("TV_MARCH_ABLATE"):   # dev only: compile the scan kernel's ablation switches in
    FLAGS.append("-DTV_MARCH_ABLATE")
if os.environ.get("TV_SLICE_STAMP"):    # dev only: per-wave barrier-wait stamps in ssd_slice.hip
    FLAGS.append("-DTV_SLICE_STAMP")


FLAGS += os.environ.get("TV_EXTRA_HIPCC_FLAGS", "").split()   # dev only


def _sources():
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _deps_mtime():
    hdrs = list(CSRC.glob("*.hpp")) + list((ROOT.parent / "include").glob("*.h"))
    return max(h.stat().st_mtime for h in hdrs)


def _compile(src: Path, force: bool) -> Path:
    obj = OBJ / (src.name + ".o")
    newest = max(src.stat().st_mtime, _deps_mtime())
    if not force and obj.exists() and obj.stat().st_mtime >= newest:
        return obj
    cmd = [HIPCC, *FLAGS, "-x", "hip", "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, jobs: int = 4) -> Path:
    OBJ.mkdir(parents=True, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def ensure_built() -> Path:
    """Compile the library if the tree does not carry it (a checkout without build artefacts);
    a no-op when `lib/libtimeviper_hip.so` exists.  This is a BUILD step, not a fallback: without a
    working hipcc it raises, and the operators keep failing loudly."""
    return LIB if LIB.exists() else build()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    a = ap.parse_args()
    print(build(a.force, a.jobs))
