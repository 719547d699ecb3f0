"""The generator of the scan's step body (timeviper_amd/devtools/gen_head_step.py): the emitter's wait counts and wait states on
small hand-made streams, and structural invariants of the generated stream (csrc/ssd_head_step.inc).  No GPU, no compiler."""
import importlib.util
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("gen_head_step", ROOT / "timeviper_amd" / "devtools" / "gen_head_step.py")
G = importlib.util.module_from_spec(spec)
spec.loader.exec_module(G)
G.PK = False          # the committed include is the --scalar stream


def test_emitter_counts_lds_operations_from_the_youngest():
    em = G.Emitter()
    for k in range(5):
        em.emit(G.Op(f"ds_read_b128 v[{100 + 4 * k}:{103 + 4 * k}], v10", "lds", w=G.regs("v", 100 + 4 * k, 4), lds_def=f"t{k}"))
    # the consumer of the 2nd read: three younger reads may stay in flight
    em.emit(G.Op("v_add_f32 v1, v104, v2", "valu", r=["v104"], w=["v1"], lds_use=["t1"]))
    assert em.lines[-2] == "s_waitcnt lgkmcnt(3)"
    # the first read is older: already known to have landed, no second wait
    em.emit(G.Op("v_add_f32 v3, v100, v2", "valu", r=["v100"], w=["v3"], lds_use=["t0"]))
    assert em.lines[-2].startswith("v_add_f32 v1")
    # an instruction that needs two of them waits once, for the younger
    em.emit(G.Op("v_add_f32 v4, v108, v116", "valu", r=["v108", "v116"], w=["v4"], lds_use=["t2", "t4"]))
    assert em.lines[-2] == "s_waitcnt lgkmcnt(0)"


def test_emitter_guards_the_first_vector_memory_wait_and_counts_stores():
    em = G.Emitter()
    em.emit(G.Op("global_load_dwordx4 v[100:103], v9, s[0:1]", "vmem", w=G.regs("v", 100, 4), vm_def="a"))
    em.emit(G.Op("global_load_lds_dwordx4 v8, s[2:3]", "dma"))
    em.emit(G.Op("global_store_dwordx4 v7, v[104:107], s[4:5]", "store", r=G.regs("v", 104, 4)))
    em.emit(G.Op("v_mov_b32 v1, v100", "valu", r=["v100"], w=["v1"], vm_use=["a"]))
    text = "\n".join(em.lines)
    assert "s_waitcnt vmcnt(2)" in text                    # the copy and the store are younger
    assert text.index("s_waitcnt vmcnt(0)") < text.index("s_waitcnt vmcnt(2)")      # without copies (FLAG_COPY clear): everything


def test_emitter_inserts_the_wait_states():
    em = G.Emitter()
    em.emit(G.Op("v_cvt_pk_bf16_f32 v100, v1, v2", "valu", w=["v100"]))
    em.emit(G.mfma(160, 100, 104, None))                                  # VALU result -> MFMA operand: 2 states
    assert em.lines[-2] == "s_nop 1"
    em.emit(G.Op("v_accvgpr_read_b32 v5, a160", "valu", r=["a160"], w=["v5"]))    # MFMA result -> any other reader: 19
    assert em.lines[-2] == "s_nop 2" and em.lines[-3] == "s_nop 15"
    em2 = G.Emitter()
    em2.emit(G.mfma(160, 100, 104, None))
    em2.emit(G.mfma(160, 100, 104, 160))                                  # whole accumulate chain: none
    assert not any(l.startswith("s_nop") for l in em2.lines)
    em2.emit(G.Op("global_store_dwordx4 v7, v[108:111], s[4:5]", "store", r=G.regs("v", 108, 4)))
    em2.emit(G.Op("v_mov_b32 v108, 0", "valu", w=["v108"]))               # store data -> overwrite: 2
    assert em2.lines[-2] == "s_nop 1"


@pytest.fixture(scope="module")
def stream():
    inc = (ROOT / "timeviper_amd" / "csrc" / "ssd_head_step.inc").read_text()
    body = inc[inc.index("#define TV_HEAD_STEP_ASM"):inc.index("#define TV_HEAD_STEP_CLOBBERS")]
    lines = [m.group(1) for m in re.finditer(r'^\s+"(.*?)\\n\\t" \\$', body, re.M)]
    v0 = int(re.search(r"#define TV_HEAD_STEP_V0 (\d+)", inc).group(1))
    return lines, v0


def test_generated_stream_structure(stream):
    lines, v0 = stream

    def region(a, b):
        return lines[lines.index(a) + 1:lines.index(b)]
    n_mfma = lambda ls: sum(l.startswith("v_mfma") for l in ls)
    # one floating / reset / standard phase A (110 MFMAs each: 80 Yoff + 30 Ydiag), two phase B (80 each)
    assert n_mfma(lines[:lines.index(".Lhs_pa_rs_%=:")]) == 110
    assert n_mfma(region(".Lhs_pa_rs_%=:", ".Lhs_pa_s_%=:")) == 110
    assert n_mfma(region(".Lhs_pa_s_%=:", ".Lhs_pa_join_%=:")) == 110
    assert n_mfma(region(".Lhs_pa_join_%=:", ".Lhs_pb_reset_%=:")) == 80
    assert n_mfma(region(".Lhs_pb_reset_%=:", ".Lhs_pb_join_%=:")) == 80
    # every branch target exists exactly once
    labels = [l[:-1] for l in lines if l.endswith(":")]
    assert len(labels) == len(set(labels))
    for l in lines:
        m = re.match(r"s_c?branch\S*\s+(\.L\S+)", l)
        if m:
            assert m.group(1) in labels, l
    # no packed fp32 arithmetic (measured slower beside MFMAs), no scratch, no compiler-owned vector register written or read
    assert not any(l.startswith(("v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "scratch_")) for l in lines)
    for l in lines:
        for a, b, c in re.findall(r"\bv(\d+)\b|v\[(\d+):(\d+)\]", l):
            lo = int(a or b)
            assert lo >= v0, l
    # the state (a0 .. a159) is written by the update MFMAs, the re-basing pass and nothing else
    for l in lines:
        m = re.match(r"(\S+)\s+a\[?(\d+)", l)
        if m and int(m.group(2)) < 160:
            assert m.group(1) in ("v_mfma_f32_16x16x32_bf16", "v_accvgpr_write_b32"), l
            if m.group(1) == "v_accvgpr_write_b32":
                assert lines.index(l) < lines.index(".Lhs_norebase_%=:"), l
    # in a floating step every Yoff / update MFMA accumulates into the tile it writes (no tile migrates through the file)
    for l in lines:
        if not l.startswith("v_mfma"):
            continue
        m = re.match(r"v_mfma_f32_16x16x32_bf16 (a\[\d+:\d+\]), \S+, \S+, (\S+)$", l)
        assert m and m.group(2) in ("0", m.group(1)), l
