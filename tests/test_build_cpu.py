import os
"""Build-time checks on the gfx950 code objects (no GPU needed: hipcc cross-compiles, llvm-objdump / llvm-readelf read
the result).

Several kernels issue global loads from inline asm that the compiler does not track and cover them with hand-counted
`s_waitcnt vmcnt(N)` (ssdk::gload16_async for the Q fragments of the attention kernels; the C.B^T and dt loads of the
head-per-wave scan).  That is only sound while the compiler never saves or copies the destination registers between the
load and the wait — i.e. while the kernel SPILLS NOTHING: a spilled register with a load in flight is stored before its
data arrives and reloaded stale (this happened in round 3's spilling 'complete' scan kernel, since removed).
So every kernel that uses the idiom must have zero scratch; a compiler upgrade or a source edit that introduces a spill
fails here instead of producing silently wrong numbers on the GPU."""
import re
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

from timeviper_amd import build

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
needs_tools = pytest.mark.skipif(not (LLVM / "llvm-objdump").exists() or not (LLVM / "llvm-readelf").exists(),
                                 reason="ROCm LLVM tools not found")


def kernel_metadata(src_name: str, tmp_path: Path) -> dict:
    """{mangled kernel name: {private_segment_fixed_size, vgpr_spill_count, ...}} of lib/obj/<src>.o"""
    build.ensure_built()
    obj = build.OBJ / (src_name + ".o")
    work = tmp_path / src_name
    work.mkdir()
    shutil.copy(obj, work / obj.name)                      # llvm-objdump --offloading writes next to its input
    subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", obj.name], cwd=work, check=True, capture_output=True)
    dev = [p for p in work.iterdir() if "amdgcn" in p.name]
    assert len(dev) == 1, [p.name for p in work.iterdir()]
    notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", dev[0].name], cwd=work, check=True,
                           capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "name" and val.startswith("_Z"):
            cur = out.setdefault(val, {})
        elif cur is not None and key in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count",
                                         "vgpr_count", "agpr_count"):
            cur[key] = int(val)
    # (.name comes after the counts inside one kernel's block in some LLVM versions: collect per block instead)
    blocks = re.split(r"\n\s*- \.agpr_count:", notes)
    for b in blocks[1:]:
        name = re.search(r"\.name:\s*(_Z\S+)", b)
        if not name:
            continue
        d = out.setdefault(name.group(1), {})
        for key in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "vgpr_count"):
            m = re.search(r"\." + key + r":\s*(\d+)", b)
            if m:
                d[key] = int(m.group(1))
    return out


@needs_tools
def test_head_scan_fast_kernels_spill_nothing(tmp_path):
    """ssd_head_kernel<PT, 4, 2> (the kernels that run in the 9B model) carry untracked loads: zero scratch."""
    md = kernel_metadata("ssd_head.hip", tmp_path)
    fast4 = {k: v for k, v in md.items() if re.search(r"ssd_head_kernelILi\dELi4ELi2E", k)}
    assert len(fast4) >= 3, list(md)
    for k, v in fast4.items():
        assert v.get("private_segment_fixed_size") == 0 and v.get("vgpr_spill_count", 0) == 0, (k, v)
    # (the variants with fewer waves per work-group may spill: they use ordinary loads)


@needs_tools
def test_head_scan_asm_kernel_spills_nothing_and_keeps_out_of_the_body_s_registers(tmp_path):
    """ssd_head_asm_kernel (head_dim 80 x 4 heads: the 9B model's scan) keeps its state in a0..a159 and the step body's
    values in v76..v255 ACROSS statements the compiler cannot see into (csrc/ssd_head_step.inc).  That holds only while
    the compiler itself never touches those registers between the statements: no scratch, no v_accvgpr_* / AGPR operand
    and no VGPR >= TV_HEAD_STEP_V0 outside ;;#ASMSTART .. ;;#ASMEND."""
    md = kernel_metadata("ssd_head.hip", tmp_path)
    k = [v for n, v in md.items() if "ssd_head_asm_kernel" in n]
    assert len(k) == 1, list(md)
    assert k[0].get("private_segment_fixed_size") == 0 and k[0].get("vgpr_spill_count", 0) == 0, k[0]
    inc = (build.CSRC / "ssd_head_step.inc").read_text()
    v0 = int(re.search(r"#define TV_HEAD_STEP_V0 (\d+)", inc).group(1))
    asm = tmp_path / "ssd_head.s"
    subprocess.run([build.HIPCC, *build.FLAGS, f"-I{build.CSRC}", "-x", "hip", "-S", "--cuda-device-only",
                    str(build.CSRC / "ssd_head.hip"), "-o", str(asm)], check=True, capture_output=True)
    text = asm.read_text()
    body = text[text.index("ssd_head_asm_kernel"):]
    body = body[body.index(": ; @"):body.index(".amdhsa_kernel")]
    inside, bad = False, []
    for line in body.splitlines():
        if "#ASMSTART" in line:
            inside = True
        elif "#ASMEND" in line:
            inside = False
        elif not inside:
            t = line.strip()
            if not t or t.startswith((";", ".")):
                continue
            regs = [int(a or c) for a, b, c in re.findall(r"\bv(\d+)\b|v\[(\d+):(\d+)\]", t)]
            if "accvgpr" in t or re.search(r"\ba\[?\d", t) or "scratch_" in t or any(r >= v0 for r in regs):
                bad.append(t)
    assert not bad, bad[:5]


def test_head_step_include_is_what_the_generator_writes():
    """csrc/ssd_head_step.inc is generated (devtools/gen_head_step.py --scalar): the committed file must be the generator's
    output, so that the schedule, the register map and the wait counts can be read (and changed) in one place."""
    import sys
    out = subprocess.run([sys.executable, str(build.ROOT / "devtools" / "gen_head_step.py"), "--scalar"], check=True,
                         capture_output=True, text=True).stdout
    assert out == (build.CSRC / "ssd_head_step.inc").read_text()


@needs_tools
def test_attention_kernels_with_untracked_q_loads_spill_nothing(tmp_path):
    """flash_fwd_kernel / flash_fwd_stream_kernel load their Q fragments with ssdk::gload16_async."""
    for src in ("attention.hip", "attention_fp8.hip"):
        text = (build.CSRC / src).read_text()
        if "gload16_async" not in text:
            continue
        md = kernel_metadata(src, tmp_path)
        kernels = {k: v for k, v in md.items() if "flash_fwd" in k}
        assert kernels, (src, list(md))
        for k, v in kernels.items():
            assert v.get("private_segment_fixed_size") == 0 and v.get("vgpr_spill_count", 0) == 0, (src, k, v)


@needs_tools
def test_gemm_kernels_spill_nothing(tmp_path):
    """Both GEMM kernels keep five half-tiles of LDS-DMA copies in flight behind hand-counted waits and, at two waves per
    SIMD, live within 256 registers: a spill would put scratch loads — which the compiler waits for with vmcnt(0) — into
    the copy queue and drain the pipeline every K-tile.  (Scalar spills into vector lanes are allowed: no memory.)"""
    for src, pat, n in (("gemm.hip", "gemm_bf16_kernel", 3), ("gemm_persist.hip", "gemm_persist_kernel", 2),
                        ("gemm_drip.hip", "gemm_drip_kernel", 3)):
        md = kernel_metadata(src, tmp_path)
        kernels = {k: v for k, v in md.items() if pat in k}
        assert len(kernels) == n, (src, list(md))
        for k, v in kernels.items():
            assert v.get("private_segment_fixed_size") == 0 and v.get("vgpr_spill_count", 0) == 0, (src, k, v)


def test_gemm_drip_register_statements_are_current():
    """csrc/gemm_drip_regs.inc (the statements of the 256 x 192 GEMM that name accumulator registers) is what
    devtools/gen_gemm_drip_regs.py emits."""
    r = subprocess.run([sys.executable, str(ROOT / "timeviper_amd" / "devtools" / "gen_gemm_drip_regs.py"), "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@needs_tools
def test_gemm_drip_accumulator_registers_are_left_alone(tmp_path):
    """gemm_drip.hip keeps its accumulators (a[0:95]) and the parked half of the finished tile (a[96:119]) in registers the
    compiler does not allocate: outside the hand-written statements the compiled kernel must not touch an accumulator
    register (it is built with -amdgpu-spill-vgpr-to-agpr=0 for that reason), and it must use no scratch."""
    src = ROOT / "timeviper_amd" / "csrc" / "gemm_drip.hip"
    out = tmp_path / "gemm_drip.s"
    flags = [f for f in build.FLAGS if not f.startswith("-fPIC")] + build.FILE_FLAGS["gemm_drip.hip"]
    r = subprocess.run([build.HIPCC, *flags, "-x", "hip", "-S", "--cuda-device-only", str(src), "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    inasm, bad = False, []
    for ln in out.read_text().splitlines():
        if "#ASMSTART" in ln:
            inasm = True
        elif "#ASMEND" in ln:
            inasm = False
        elif not inasm and not ln.lstrip().startswith((";", ".")) and re.search(r"[\s,\[]a\[?\d", ln):
            bad.append(ln.strip())
        if "scratch_" in ln:
            bad.append(ln.strip())
    assert not bad, bad[:10]

@needs_tools
def test_vit_attention_asm_kernel_keeps_out_of_the_body_s_registers(tmp_path):
    """flash_fwd_vit_kernel (csrc/attention_vit.hpp) keeps P, the running maxima (v[TV_FAV_V0 ..]) and O / S / Q (a0 .. a231)
    in named registers ACROSS the statements of a query block: between the block's first statement (TV_FAV_BEGIN_ASM) and
    its last (TV_FAV_EPI_ASM) the compiler must use no vector register >= TV_FAV_V0 and no accumulation register below
    232, and the kernel must have no scratch."""
    md = kernel_metadata("attention.hip", tmp_path)
    k = [v for n, v in md.items() if "flash_fwd_vit_kernel" in n]
    assert len(k) == 1, list(md)
    assert k[0].get("private_segment_fixed_size") == 0 and k[0].get("vgpr_spill_count", 0) == 0, k[0]
    inc = (build.CSRC / "attention_vit_tile.inc").read_text()
    v0 = int(re.search(r"#define TV_FAV_V0 (\d+)", inc).group(1))
    asm = tmp_path / "attention.s"
    subprocess.run([build.HIPCC, *build.FLAGS, f"-I{build.CSRC}", "-x", "hip", "-S", "--cuda-device-only",
                    str(build.CSRC / "attention.hip"), "-o", str(asm)], check=True, capture_output=True)
    text = asm.read_text()
    body = text[text.index("flash_fwd_vit_kernel"):]
    body = body[body.index(": ; @"):body.index(".amdhsa_kernel")]
    inside, in_block, cur, bad = False, False, [], []
    for line in body.splitlines():
        if "#ASMSTART" in line:
            inside, cur = True, []
        elif "#ASMEND" in line:
            inside = False
            big = len(cur) > 40
            mfmas = sum("v_mfma" in c for c in cur)
            if big and mfmas == 0 and any("v_accvgpr_write_b32 a0, 0" in c for c in cur):
                in_block = True                       # TV_FAV_BEGIN_ASM
            elif big and mfmas == 36 and not any("v_exp_f32" in c for c in cur):
                in_block = False                      # TV_FAV_EPI_ASM
        elif inside:
            cur.append(line)
        elif in_block:
            t = line.strip()
            if not t or t.startswith((";", ".")):
                continue
            regs = [int(a or c) for a, b, c in re.findall(r"\bv(\d+)\b|v\[(\d+):(\d+)\]", t)]
            aregs = [int(a or c) for a, b, c in re.findall(r"\ba(\d+)\b|a\[(\d+):(\d+)\]", t)]
            if "scratch_" in t or any(r >= v0 for r in regs) or any(r < 232 for r in aregs):
                bad.append(t)
    assert not bad, bad[:5]


def test_vit_attention_tile_include_is_what_the_generator_writes():
    """csrc/attention_vit_tile.inc is generated (devtools/gen_fa_vit.py)."""
    import sys
    out = subprocess.run([sys.executable, str(build.ROOT / "devtools" / "gen_fa_vit.py")], check=True, capture_output=True,
                         text=True, env={k: v for k, v in os.environ.items() if k != "TV_FAV_ABLATE"}).stdout
    assert out == (build.CSRC / "attention_vit_tile.inc").read_text()
