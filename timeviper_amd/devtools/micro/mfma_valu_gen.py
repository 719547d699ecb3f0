"""Dev micro-benchmark (with mfma_valu.hip): how well do the matrix and the vector pipe of a SIMD overlap?
    cd timeviper_amd/devtools/micro && python mfma_valu_gen.py && hipcc --offload-arch=gfx950 -O2 mfma_valu.hip -o mfma_valu && ./mfma_valu
Measured (MI355X): A 725 ns, B 659 ns per unit and SIMD (B / A 0.91)."""
# micro-benchmark: one "unit" = the per-tile work of the ViT attention for 32 query rows: 33 MFMA 32x32x16 + 190 VALU (48 exp)
# A: units clumped (15 MFMA | 190 VALU | 18 MFMA), two waves a SIMD (each wave 1 unit per iteration)
# B: one wave a SIMD running two units per iteration, unit X's VALU interleaved behind unit Y's MFMAs
def valu_ops(base):
    ops = []
    r = lambda i: f"v{base + (i % 24)}"
    k = 0
    for i in range(24): ops.append(f"v_pk_fma_f32 v[{base + 2 * (i % 12)}:{base + 2 * (i % 12) + 1}], v[{base + 2 * (i % 12)}:{base + 2 * (i % 12) + 1}], v[{base + 24}:{base + 25}], v[{base + 26}:{base + 27}]")
    for i in range(24): ops.append(f"v_max3_f32 {r(i)}, {r(i)}, {r(i + 1)}, {r(i + 2)}")
    for i in range(48): ops.append(f"v_exp_f32 {r(i)}, {r(i)}")
    for i in range(24): ops.append(f"v_cvt_pk_bf16_f32 {r(i)}, {r(i)}, {r(i + 1)}")
    for i in range(70): ops.append(f"v_fma_f32 {r(i)}, {r(i)}, v{base + 24}, v{base + 26}")
    assert len(ops) == 190
    return ops
def mfma(acc, a, b):
    return f"v_mfma_f32_32x32x16_bf16 a[{acc}:{acc + 15}], v[{a}:{a + 3}], v[{b}:{b + 3}], a[{acc}:{acc + 15}]"
def unit_mfmas(accbase):
    qk = [mfma(accbase + 16 * (i % 3), 0, 4) for i in range(15)]
    pv = [mfma(accbase + 48 + 16 * (i % 3), 8, 12) for i in range(18)]
    return qk, pv
def body_A():
    qk, pv = unit_mfmas(0)
    return qk + valu_ops(16) + pv
def body_B():
    # two units: X (acc 0..95, valu v16..), Y (acc 96..191, valu v48..); Y lags half a unit: X's VALU behind Y's MFMAs and vice versa
    qx, px = unit_mfmas(0)
    qy, py = unit_mfmas(96)
    vx, vy = valu_ops(16), valu_ops(48)
    out = []
    mx, my = qx + px, py + qy          # X: QK then PV; Y: (previous tile's) PV then QK of the next: Y's softmax falls where X multiplies
    # phase 1: X's 15 QK MFMAs + Y's 18 PV MFMAs ... simple model: all 66 MFMAs in order, 380 VALU spread evenly behind them
    m = []
    for i in range(33):
        m.append(mx[i]); m.append(my[i])
    v = []
    for i in range(190):
        v.append(vx[i]); v.append(vy[i])
    per = len(v) / len(m)
    acc = 0.0; vi = 0
    for i, ins in enumerate(m):
        out.append(ins)
        acc += per
        while vi < len(v) and vi < round(acc):
            out.append(v[vi]); vi += 1
    out += v[vi:]
    return out
def emit(name, lines):
    s = f"#define {name} \\\n"
    s += " \\\n".join('  "' + l + '\\n\\t"' for l in lines)
    return s + "\n"
open("body.inc", "w").write(emit("BODY_A", body_A()) + emit("BODY_B", body_B()))
