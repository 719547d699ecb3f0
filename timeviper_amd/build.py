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
if os.environ.get("TV_FA_VARIANTS"):    # dev only: the two measured-slower ViT attention kernels (attention_variants.hpp)
    FLAGS.append("-DTV_FA_VARIANTS")
if os.environ.get("TV_SLICE_STAMP"):    # dev only: per-wave barrier-wait stamps in ssd_slice.hip
    FLAGS.append("-DTV_SLICE_STAMP")


FLAGS += os.environ.get("TV_EXTRA_HIPCC_FLAGS", "").split()   # dev only


def _sources():
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _headers():
    return sorted(list(CSRC.glob("*.hpp")) + list((ROOT.parent / "include").glob("*.h")))


def _hash(paths, extra: str = "") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()[:16]


def source_id() -> str:
    """Hash of every source, header and compiler flag the library is built from."""
    return _hash(_sources() + _headers(), " ".join(FLAGS))


# Kernels whose accumulation / vector register split hipcc cannot express from source.  A 512-thread work-group has 256
# registers a lane; as soon as a kernel uses accumulation registers the backend splits them 128 / 128 unless the function
# carries the LLVM attribute "amdgpu-agpr-alloc" (SIRegisterInfo::getMaxNumVectorRegs), for which clang has no spelling.
# Such sources are compiled in the steps hipcc itself runs — device IR, code object, bundle, host object — with the
# attribute added to the kernel's attribute group in between.  {source: {kernel name fragment: attribute text}}
IR_ATTRS = {"ssd_pair.hip": {"ssd_pair_kernel": '"amdgpu-agpr-alloc"="144"'}}
LLVM_BIN = Path(os.environ.get("TV_LLVM_BIN", "/opt/rocm/lib/llvm/bin"))


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(map(str, cmd[:3]))} ... failed:\n{r.stdout}\n{r.stderr}")
    return r


def _compile_ir_patched(src: Path, obj: Path, flags) -> None:
    import re
    work = OBJ / (src.name + ".ir")
    work.mkdir(parents=True, exist_ok=True)
    ll, llp, dev_o, dev_out, fb = (work / n for n in ("dev.ll", "dev_patched.ll", "dev.o", "dev.out", "dev.hipfb"))
    dev_flags = [f for f in flags if f != "-fPIC"]
    _run([HIPCC, *dev_flags, "-x", "hip", "--cuda-device-only", "-emit-llvm", "-S", str(src), "-o", str(ll)])
    text = ll.read_text()
    for frag, attr in IR_ATTRS[src.name].items():
        m = re.search(r"define [^\n]*@[A-Za-z0-9_]*%s[A-Za-z0-9_]*\([^\n]*\)[^\n{]*#(\d+)" % re.escape(frag), text)
        if not m:
            raise RuntimeError(f"{src.name}: kernel `{frag}` not found in the device IR")
        text, n = re.subn(r"(attributes #%s = \{)" % m.group(1), r"\1 " + attr, text, count=1)
        if n != 1:
            raise RuntimeError(f"{src.name}: attribute group #{m.group(1)} not found")
    llp.write_text(text)
    _run([str(LLVM_BIN / "clang"), "-x", "ir", str(llp), "-target", "amdgcn-amd-amdhsa", f"-mcpu={ARCH}", "-O3", "-c", "-o", str(dev_o)])
    _run([str(LLVM_BIN / "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", str(dev_out), str(dev_o)])
    _run([str(LLVM_BIN / "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
          f"-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{ARCH}", "-input=/dev/null", f"-input={dev_out}",
          f"-output={fb}"])
    _run([HIPCC, *flags, "-x", "hip", "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", str(fb),
          "-c", str(src), "-o", str(obj)])


def _compile(src: Path, force: bool, build_id: str) -> Path:
    obj = OBJ / (src.name + ".o")
    tag = OBJ / (src.name + ".hash")
    flags = list(FLAGS)
    if src.name == "capi.cpp":                    # the one translation unit that carries the id
        flags.append(f'-DTV_BUILD_ID="{build_id}"')
    want = _hash([src] + _headers(), " ".join(flags) + repr(IR_ATTRS.get(src.name, "")))
    if not force and obj.exists() and tag.exists() and tag.read_text() == want:
        return obj
    if src.name in IR_ATTRS:
        _compile_ir_patched(src, obj, flags)
        tag.write_text(want)
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
