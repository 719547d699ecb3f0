"""Generator of csrc/attention_vit_tile.inc: the key-tile loop body of the ViT attention kernel (csrc/attention_vit.hip) as
hand-scheduled instruction streams — ONE wave per SIMD, 64 query rows a wave (units X and Y of 32 rows), software-pipelined
ACROSS key tiles: statement i runs the softmax of tile i on the vector pipe beside P.V of tile i - 1 and Q.K^T of tile i + 1 on
the matrix pipe.  K / V fragment reads are shared by the two units.  Why: DESIGN.md section 5 — the compiled kernel runs its
two waves per SIMD phase-aligned (one barrier a tile), so the matrix pipe idles during every softmax; the tile takes 5 900
cycles against 2 800 of vector work.

    python timeviper_amd/devtools/gen_fa_vit.py > timeviper_amd/csrc/attention_vit_tile.inc      (--summary: counts)

Register map (the compiler keeps v0 .. V0-1; nothing but what is listed lives across statements):
  a0-47 / a48-95     O^T accumulators of unit X / Y (3 d-tiles x 16)            across statements
  a96-143 / a144-191 S^T accumulators of X / Y (3 key sub-tiles x 16)            across statements (tile i + 1)
  a192-211 / a212-231 Q^T fragments of X / Y (5 k-steps x 4)                     across statements
  v[KF]   K fragments of one sub-tile (5 x 4)        v[VF0], v[VF1]  V^T fragments of a 16-key step (3 d-tiles x 4), two sets
  v[SWX], v[SWY]  the 48 scores of a lane and unit (copied out of the accumulators, exponentiated in place)
  v[PX], v[PY]    P^T as bf16 pairs (24 each)                                    across statements (tile i -> statement i + 1)
  v[MU] m_used X, Y;  v[MT] m_true X, Y;  v[AL] alpha X, Y                       across statements
Lazy maximum: the exponentials use m_used, which follows the true running maximum only when that has grown by more than 8
(log2 units): P <= 2^8, the O accumulators are rescaled (TV_FAV_RESCALE_ASM, rare) only then.  The row sums of P come out of
the P.V MFMAs through the ones column of the V ring (as in flash_fwd_stream_kernel)."""
import os
import sys

ABL = os.environ.get("TV_FAV_ABLATE", "")      # dev: timing experiments (wrong results): nodma, fewwaits, noexp, nomask
V0 = 32
KF = V0                  # 20
VF = [V0 + 20, V0 + 32]  # 2 x 12
SW = [V0 + 44, V0 + 92]  # 2 x 48
PB = [V0 + 140, V0 + 164]  # 2 x 24
TMAX = [V0 + 188, V0 + 189]
MU = [V0 + 190, V0 + 191]
MT = [V0 + 192, V0 + 193]
AL = [V0 + 194, V0 + 195]
T0, T1 = V0 + 196, V0 + 197
SCL = V0 + 198           # pair (scale, scale)
NEGM = V0 + 200          # pair (-m_used, -m_used) of the unit in work
KA = V0 + 202            # 5 K read addresses
VA = V0 + 207            # 3 V read addresses
DOFF = V0 + 210          # 4 rotating DMA offset temporaries
LIM = V0 + 214           # mask limit of the lane (last tile)
RT = V0 + 215            # 8 rescale temporaries .. V0 + 222
VEND = V0 + 223
assert VEND <= 255
OACC = [0, 48]
SACC = [96, 144]
QF = [192, 212]
THRESH = "0x41c00000"    # 24.0: P <= 2^24 (bf16 and the fp32 sums hold that with room to spare); a rescale costs 300 instructions


def mfma(dst, a, b, c, akind="v", bkind="v"):
    cc = f"a[{c}:{c + 15}]" if c is not None else "0"
    return f"v_mfma_f32_32x32x16_bf16 a[{dst}:{dst + 15}], {akind}[{a}:{a + 3}], {bkind}[{b}:{b + 3}], {cc}"


class Stream:
    """merges the MFMA list (with the LDS reads and waits tied to it), the vector list and the copy groups into one order"""

    def __init__(self):
        self.out = []
        self.issued = []          # LDS read ids in issue order

    def emit(self, s):
        self.out.append(s)

    def read(self, rid, ins):
        self.issued.append(rid)
        self.out.append(ins)

    def need(self, rids):
        if not rids:
            return
        last = max(self.issued.index(r) for r in rids)
        n = len(self.issued) - 1 - last
        self.out.append(f"s_waitcnt lgkmcnt({min(n, 15)})")


def k_reads(t):
    return [((("k", t, ks)), f"ds_read_b128 v[{KF + 4 * ks}:{KF + 4 * ks + 3}], v{KA + ks} offset:{t * 8192}") for ks in range(5)]


def v_reads(s):
    out = []
    b = VF[s % 2]
    for dt in range(3):
        out.append((("v", s, dt, 0), f"ds_read_b64_tr_b16 v[{b + 4 * dt}:{b + 4 * dt + 1}], v{VA + dt} offset:{s * 4096}"))
        out.append((("v", s, dt, 1), f"ds_read_b64_tr_b16 v[{b + 4 * dt + 2}:{b + 4 * dt + 3}], v{VA + dt} offset:{s * 4096 + 2048}"))
    return out


def qk_block(t):
    m = []
    for ks in range(5):
        for u in range(2):
            m.append(dict(ins=mfma(SACC[u] + 16 * t, KF + 4 * ks, QF[u] + 4 * ks, None if ks == 0 else SACC[u] + 16 * t, "v", "a"),
                          need=[("k", t, ks)], kind="qk", t=t, u=u, last=(ks == 4 and u == 1)))
    return m


def pv_block(s):
    m = []
    b = VF[s % 2]
    for dt in range(3):
        for u in range(2):
            m.append(dict(ins=mfma(OACC[u] + 16 * dt, b + 4 * dt, PB[u] + 4 * s, OACC[u] + 16 * dt),
                          need=[("v", s, dt, 0), ("v", s, dt, 1)], kind="pv", s=s, u=u, last=(dt == 2 and u == 1)))
    return m


def copy_out():
    """S(i) out of the accumulators: sub-tile by sub-tile (the first Q.K^T MFMAs of the statement overwrite sub-tile 0 first)"""
    v = []
    for t in range(3):
        for u in range(2):
            for i in range(16):
                v.append(("copy", t, f"v_accvgpr_read_b32 v{SW[u] + 16 * t + i}, a{SACC[u] + 16 * t + i}"))
    return v


def softmax(u, masked):
    sw = SW[u]
    v = []
    if masked:        # keys past the end of the sequence: -inf (key of element (t, i) = 32 t + (i & 3) + 8 (i >> 2) + 4 hh; LIM = live - 4 hh)
        for t in range(3):            # a sub-tile that is all there skips its sixteen tests (one glued group: the branch must not jump an MFMA)
            lab = f"fav_m{u}{t}_%="
            grp = [f"s_cmp_ge_i32 %[live], {32 * (t + 1)}", f"s_cbranch_scc1 {lab}"]
            for i in range(16):
                c = 32 * t + (i & 3) + 8 * (i >> 2)
                grp.append(f"v_mov_b32_e32 v{T1}, {c}")
                grp.append(f"v_cmp_gt_i32_e32 vcc, v{LIM}, v{T1}")
                grp.append(f"v_cndmask_b32_e32 v{sw + 16 * t + i}, v{T0}, v{sw + 16 * t + i}, vcc")
            grp.append(f"{lab}:")
            v.append(grp)
    v.append(f"v_max3_f32 v{TMAX[u]}, v{sw}, v{sw + 1}, v{sw + 2}")
    for j in range(3, 47, 2):
        v.append(f"v_max3_f32 v{TMAX[u]}, v{TMAX[u]}, v{sw + j}, v{sw + j + 1}")
    v.append(f"v_max_f32_e32 v{TMAX[u]}, v{TMAX[u]}, v{sw + 47}")
    v.append(f"v_mov_b32_e32 v{T1}, v{TMAX[u]}")
    v.append("s_nop 1")
    v.append(f"v_permlane32_swap_b32 v{T1}, v{TMAX[u]}")
    v.append("s_nop 1")
    v.append(f"v_max_f32_e32 v{TMAX[u]}, v{T1}, v{TMAX[u]}")
    v.append(f"v_mul_f32_e32 v{T1}, %[scale], v{TMAX[u]}")
    v.append(f"v_max_f32_e32 v{MT[u]}, v{MT[u]}, v{T1}")
    v.append(f"v_add_f32_e32 v{T1}, {THRESH}, v{MU[u]}")
    v.append(f"v_cmp_gt_f32_e32 vcc, v{MT[u]}, v{T1}")
    v.append(f"v_cmp_class_f32_e64 s[6:7], v{MU[u]}, 4")            # m_used == -inf: the block's first tile (O is zero: no rescale)
    v.append(f"v_cndmask_b32_e32 v{T1}, v{MU[u]}, v{MT[u]}, vcc")    # new m_used
    v.append(f"v_sub_f32_e32 v{AL[u]}, v{MU[u]}, v{T1}")
    v.append(f"v_exp_f32_e32 v{AL[u]}, v{AL[u]}")
    v.append(f"v_mov_b32_e32 v{MU[u]}, v{T1}")
    v.append("s_nop 3")
    v.append("s_andn2_b64 s[6:7], vcc, s[6:7]")
    v.append("s_or_b64 s[8:9], s[8:9], s[6:7]")
    v.append(f"v_sub_f32_e32 v{NEGM}, 0, v{T1}")
    v.append(f"v_sub_f32_e32 v{NEGM + 1}, 0, v{T1}")
    for j in range(24):
        v.append(f"v_pk_fma_f32 v[{sw + 2 * j}:{sw + 2 * j + 1}], v[{sw + 2 * j}:{sw + 2 * j + 1}], v[{SCL}:{SCL + 1}], v[{NEGM}:{NEGM + 1}]")
    for j in range(48):
        v.append(f"v_exp_f32_e32 v{sw + j}, v{sw + j}")
    return [("soft", u, x) for x in v]


def cvts():
    v = []
    for u in range(2):
        for j in range(24):
            v.append(("cvt", j // 4, f"v_cvt_pk_bf16_f32 v{PB[u] + j}, v{SW[u] + 2 * j}, v{SW[u] + 2 * j + 1}"))
    return v


def dma_window(which, clamped):
    """the six pieces of K (which = 0) or V of a statement in ONE exec window (lanes of pad chunks stay out): piece i = rows
    row0 + 16 i.  Fast form: one lane offset for all pieces (an input), the scalar source pointer walks 16 rows a piece.
    Clamped form (a tile with fewer than 96 rows): rows past the end request the last row again, offsets per piece."""
    pre = "k" if which == 0 else "v"
    out = []
    if clamped:
        for i in range(6):
            t = DOFF + i % 4
            out += [f"v_add_u32_e32 v{t}, {16 * i}, %[row0]", f"v_min_i32_e32 v{t}, %[{pre}left], v{t}",
                    f"v_mad_u32_u24 v{t}, v{t}, %[{pre}stride], %[{pre}co]"]
            if i % 4 == 3 or i == 5:      # four offset temporaries: issue what is ready
                lo = i - (i % 4)
                out.append(f"s_mov_b64 exec, %[{pre}live]")
                for j in range(lo, i + 1):
                    out += [f"s_add_u32 m0, %[{pre}m], {4096 * j}", "s_nop 0", f"global_load_lds_dwordx4 v{DOFF + j % 4}, %[{pre}src]"]
                out.append("s_mov_b64 exec, -1")
        return out
    out.append(f"s_mov_b64 s[10:11], %[{pre}src]")
    out.append(f"s_mov_b64 exec, %[{pre}live]")
    for i in range(6):
        out += [f"s_add_u32 m0, %[{pre}m], {4096 * i}", "s_nop 0", f"global_load_lds_dwordx4 %[{pre}off], s[10:11]"]
        if i < 5:
            out += [f"s_add_u32 s10, s10, %[{pre}step]", "s_addc_u32 s11, s11, 0"]
    out.append("s_mov_b64 exec, -1")
    return out


def q_loads():
    """the next block's Q^T fragments straight into the accumulation registers (the last tile of a block runs no Q.K^T)"""
    out = []
    for u, (o, o4) in enumerate((("qox", "qox4"), ("qoy", "qoy4"))):
        for ks in range(4):
            out.append(f"global_load_dwordx4 a[{QF[u] + 4 * ks}:{QF[u] + 4 * ks + 3}], %[{o}], %[qsrc] offset:{32 * ks}")
        out.append(f"global_load_dwordx4 a[{QF[u] + 16}:{QF[u] + 19}], %[{o4}], %[qsrc]")
    return out


def setup(kind):
    s = []
    if kind in ("tile", "pro"):
        for ks in range(5):
            s.append(f"v_add_u32_e32 v{KA + ks}, %[sk], %[kr{ks}]")
    if kind in ("tile", "last", "epi"):
        for dt in range(3):
            s.append(f"v_add_u32_e32 v{VA + dt}, %[sv], %[vr{dt}]")
    if kind in ("tile", "last"):
        s.append(f"v_mov_b32_e32 v{SCL}, %[scale]")
        s.append(f"v_mov_b32_e32 v{SCL + 1}, %[scale]")
        s.append("s_mov_b64 s[8:9], 0")
    if kind == "last":
        s.append(f"v_mov_b32_e32 v{T0}, 0xff800000")
        s.append(f"v_mov_b32_e32 v{LIM}, %[lim]")
    return s


def build(kind0):
    """kind: 'tile' (steady; 'tilec': its copies clamped to a partial tile), 'tile0' (a block's first: no P.V, copies clamped),
    'last' (masked softmax, no Q.K^T, the next block's Q loads), 'pro' (Q.K^T of tile 0), 'epi' (P.V of the last tile)"""
    clamped = kind0 in ("tilec", "tile0", "last")
    first = kind0 == "tile0"
    kind = "tile" if kind0 in ("tilec", "tile0") else kind0
    st = Stream()
    for x in setup(kind):
        st.emit(x)
    if first:
        blocks = [qk_block(t) for t in range(3)]
        order = [("k", t) for t in range(3)]
    elif kind == "tile":
        blocks = [pv_block(0), pv_block(1), qk_block(0), pv_block(2), pv_block(3), qk_block(1), pv_block(4), pv_block(5), qk_block(2)]
        order = [("v", 0), ("v", 1), ("k", 0), ("v", 2), ("v", 3), ("k", 1), ("v", 4), ("v", 5), ("k", 2)]
    elif kind == "last" or kind == "epi":
        blocks = [pv_block(s) for s in range(6)]
        order = [("v", s) for s in range(6)]
    else:
        blocks = [qk_block(t) for t in range(3)]
        order = [("k", t) for t in range(3)]
    m = [x for b in blocks for x in b]
    starts = []
    p = 0
    for b in blocks:
        starts.append(p)
        p += len(b)
    reads = [k_reads(n) if k == "k" else v_reads(n) for k, n in order]
    # a block's fragments are requested behind the FIRST MFMA of the block in front of it that uses another register set:
    #   V sets alternate by step, K has one set: its reads go behind the last MFMA of the previous Q.K^T block
    at = {}
    last_k_block_end = None
    for j, (k, n) in enumerate(order):
        if k == "v":
            # previous user of this V set: step n - 2
            prev = [jj for jj, (kk, nn) in enumerate(order) if kk == "v" and nn == n - 2]
            pos = (starts[prev[0]] + len(blocks[prev[0]])) if prev else -1
        else:
            pos = last_k_block_end if last_k_block_end is not None else -1
            last_k_block_end = starts[j] + len(blocks[j])
        at.setdefault(pos, []).extend(reads[j])
    valu = []
    if kind in ("tile", "last"):
        valu += copy_out() + softmax(0, kind == "last") + softmax(1, kind == "last") + cvts()
    dmas = [dma_window(0, clamped), dma_window(1, clamped)] if kind in ("tile", "last") else []
    if "nodma" in ABL:
        dmas = []
    dma_at = [len(m) // 6, (len(m) * 3) // 5]
    if "novalu" in ABL:
        valu = []
    if "nolds" in ABL:
        at = {}
        for x in m:
            x["need"] = []
    if "nomfma" in ABL:
        for x in m:
            x["ins"] = "s_nop 0"
    if "noexp" in ABL:
        valu = [x for x in valu if isinstance(x[2], list) or "v_exp_f32_e32 v" not in x[2] or "soft" != x[0]]
    # the start: reads that have no predecessor, then the Q loads (older than the copies)
    if kind in ("tile", "last"):
        st.emit("s_nop 7")
        st.emit("s_nop 7")        # (the S accumulators were written by the previous statement's last MFMAs)
    for rid, ins in at.get(-1, []):
        st.read(rid, ins)
    if kind == "last":
        for x in q_loads():
            st.emit(x)
    nm = len(m)
    nv = len(valu)
    ncopy_cvt = sum(1 for x in valu if x[0] == "cvt")
    vi = 0

    def emit_v(x):
        for y in (x if isinstance(x, list) else [x]):
            st.emit(y)
    quota = (nv - ncopy_cvt) / max(nm - 2, 1)       # the conversions come behind the last P.V MFMA
    acc = 0.0
    di = 0
    copied = {}          # sub-tile -> index of the vector instruction that finished its copy
    for i, x in enumerate(m):
        if x["kind"] == "qk" and kind == "tile":
            # the MFMA overwrites the scores of sub-tile t: they must have been copied out
            done = max(j for j, vv in enumerate(valu) if vv[0] == "copy" and vv[1] == x["t"])
            while vi <= done:
                emit_v(valu[vi][2])
                vi += 1
        if i in starts:          # one wait a block: for the block's last fragment read
            blk = starts.index(i)
            st.need([rid for y in blocks[blk] for rid in y["need"]])
        st.emit(x["ins"])
        for rid, ins in at.get(i + 1, []):
            st.read(rid, ins)
        if dmas and di < len(dmas) and i == dma_at[di]:
            for d in dmas[di]:
                st.emit(d)
            di += 1
        acc += quota
        while vi < nv - ncopy_cvt and vi < round(acc):
            emit_v(valu[vi][2])
            vi += 1
    while vi < nv - ncopy_cvt:
        emit_v(valu[vi][2])
        vi += 1
    while di < len(dmas):
        for d in dmas[di]:
            st.emit(d)
        di += 1
    for x in valu[nv - ncopy_cvt:]:
        st.emit(x[2])
    if kind in ("tile", "last"):
        st.emit("s_mov_b32 %[flag], s8")
        st.emit("s_or_b32 %[flag], %[flag], s9")
        st.emit("s_waitcnt vmcnt(12)" if "nodma" not in ABL else "s_waitcnt vmcnt(0)")
    st.emit("s_waitcnt lgkmcnt(0)")
    if kind == "epi":
        st.emit("s_nop 7")
        st.emit("s_nop 7")
        st.emit("s_nop 7")        # (the O accumulators are read by vector instructions next)
    return st.out


def rescale():
    out = ["s_nop 7", "s_nop 7", "s_nop 7"]
    for u in range(2):
        for j in range(0, 48, 8):
            for k in range(8):
                out.append(f"v_accvgpr_read_b32 v{RT + k}, a{OACC[u] + j + k}")
            for k in range(8):
                out.append(f"v_mul_f32_e32 v{RT + k}, v{AL[u]}, v{RT + k}")
            for k in range(8):
                out.append(f"v_accvgpr_write_b32 a{OACC[u] + j + k}, v{RT + k}")
    out += ["s_nop 3"]
    return out


def begin_block():
    """O = 0, m_used = m_true = -inf (a block's first statement runs no P.V)"""
    out = []
    for u in range(2):
        for j in range(48):
            out.append(f"v_accvgpr_write_b32 a{OACC[u] + j}, 0")
        out.append(f"v_mov_b32_e32 v{MU[u]}, 0xff800000")
        out.append(f"v_mov_b32_e32 v{MT[u]}, 0xff800000")
        out.append(f"v_mov_b32_e32 v{AL[u]}, 1.0")
    return out


def store_block(D):
    """O[q][d] = O^T accumulators / l of both units as bf16, 16 bytes a lane: a lane (r, hh) holds 4 consecutive d per group of four
    accumulator registers (d = 32 dt + 8 g + 4 hh + 0..3); v_permlane32_swap joins the halves of two groups of the lane pair
    (r, r + 32) into 16 contiguous bytes each.  l = the ones column's row: d-tile 2, register 4 (D - 64) / 8 of the hh = 0 lanes."""
    ng = D // 8
    out = []
    W = SW[0]                       # scratch: the score registers are dead here
    for u in range(2):
        ob = OACC[u]
        L, T, INV = W + 40, W + 41, W + 42        # INV: pair (inv, inv)
        out += [f"v_accvgpr_read_b32 v{L}, a{ob + 32 + 4 * ((D - 64) // 8)}", f"v_accvgpr_read_b32 v{T}, a{ob + 32 + 4 * ((D - 64) // 8)}", "s_nop 1",
                f"v_permlane32_swap_b32 v{T}, v{L}", "s_nop 1", f"v_rcp_f32_e32 v{INV}, v{T}", "s_nop 0", f"v_mov_b32_e32 v{INV + 1}, v{INV}",
                f"s_mov_b64 exec, %[mask{'xy'[u]}]"]
        gi = 0
        k = 0
        while gi < ng:
            pair = gi + 1 < ng
            R = W + 8 * (k % 4)       # 8 fp32 temporaries; the packed words land in R .. R + 3
            k += 1
            n = 8 if pair else 4
            for j in range(n):
                g2 = gi + j // 4
                out.append(f"v_accvgpr_read_b32 v{R + j}, a{ob + 16 * (g2 // 4) + 4 * (g2 % 4) + j % 4}")
            for j in range(0, n, 2):
                out.append(f"v_pk_mul_f32 v[{R + j}:{R + j + 1}], v[{R + j}:{R + j + 1}], v[{INV}:{INV + 1}]")
            # words: group gi -> R, R + 1; group gi + 1 -> R + 2, R + 3
            for j in range(0, n, 2):
                out.append(f"v_cvt_pk_bf16_f32 v{R + j // 2}, v{R + j}, v{R + j + 1}")
            if pair:
                out += ["s_nop 1", f"v_permlane32_swap_b32 v{R}, v{R + 2}", f"v_permlane32_swap_b32 v{R + 1}, v{R + 3}", "s_nop 1",
                        f"global_store_dwordx4 %[oa{'xy'[u]}], v[{R}:{R + 3}], %[obase] offset:{16 * gi}"]
                gi += 2
            else:
                out += ["s_nop 1", f"global_store_dwordx2 %[ob{'xy'[u]}], v[{R}:{R + 1}], %[obase] offset:{16 * gi}"]
                gi += 1
        out.append("s_mov_b64 exec, -1")
    return out


def macro(name, lines):
    return f"#define {name} \\\n" + " \\\n".join('  "' + l + '\\n\\t"' for l in lines) + "\n"


def clobbers():
    v = ", ".join(f'"v{i}"' for i in range(V0, VEND + 1))
    a = ", ".join(f'"a{i}"' for i in range(0, 232))
    return f'#define TV_FAV_CLOBBERS "memory", "vcc", "scc", "m0", "s6", "s7", "s8", "s9", "s10", "s11", {v}, {a}\n'


def main():
    parts = {k: build(k) for k in ("tile", "tilec", "tile0", "last", "pro", "epi")}
    if "--summary" in sys.argv:
        for k, lines in parts.items():
            kinds = {}
            for l in lines:
                op = l.split()[0]
                key = "mfma" if "mfma" in op else "exp" if "v_exp" in op else "lds" if op.startswith("ds_") else "copy" if "lds_dword" in op else \
                    "valu" if op.startswith("v_") else "salu/wait"
                kinds[key] = kinds.get(key, 0) + 1
            print(k, len(lines), kinds)
        return
    s = "// GENERATED by timeviper_amd/devtools/gen_fa_vit.py — do not edit (tests/test_build_cpu.py compares).\n"
    s += f"#define TV_FAV_V0 {V0}\n#define TV_FAV_MU_X {MU[0]}\n#define TV_FAV_MU_Y {MU[1]}\n"
    s += clobbers()
    s += macro("TV_FAV_TILE_ASM", parts["tile"]) + macro("TV_FAV_TILEC_ASM", parts["tilec"]) + macro("TV_FAV_TILE0_ASM", parts["tile0"])
    s += macro("TV_FAV_LAST_ASM", parts["last"]) + macro("TV_FAV_PRO_ASM", parts["pro"])
    s += macro("TV_FAV_STORE72_ASM", store_block(72)) + macro("TV_FAV_STORE80_ASM", store_block(80))
    s += macro("TV_FAV_EPI_ASM", parts["epi"]) + macro("TV_FAV_RESCALE_ASM", rescale()) + macro("TV_FAV_BEGIN_ASM", begin_block())
    sys.stdout.write(s)


if __name__ == "__main__":
    main()
