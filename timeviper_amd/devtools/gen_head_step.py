#!/usr/bin/env python3
"""Generator of csrc/ssd_head_step.inc: the 64-token step of the head-per-wave SSD march (ssd_head.hip,
head_dim 80, 4 waves) as ONE hand-scheduled instruction stream.

Why generated assembly: at one wave per SIMD nothing hides a stall, and hipcc's stream for the step clumps the
vector work between groups of MFMAs, shuttles accumulator tiles between the two register files at the joins of the
mode branches and exposes every LDS round trip (profiles/r05_ssd_head_isa.md).  Here every MFMA is followed by its
share of the step's vector / LDS / copy instructions, registers are assigned by this file, and the wait counts and
wait states are computed by the emitter from the final order.

Step order (one chunk c of 64 tokens, one head = one wave):
    [re-basing of the frame, when due]
    preamble   addresses, weights, C.B^T / dt loads, bf16 copy of state rows 0..31
    phase A    Yoff^T = X'^T C^T in four quarters of 32 state rows (80 MFMAs) beside: the bf16 copy of the next
               quarter's state rows, x~ = w_s x on the transposed x fragments, the LDS-DMA copies of chunk c + 1;
               then Ydiag on top (30 MFMAs, A = x~, B = causal C.B^T)
    [reset steps: x~ again with the new frame's weights]
    phase B    X' += B^T x~ (80 MFMAs) beside the y epilogue (row factor, D x, bf16, permlane swap, 16-byte stores)
The arithmetic and its order per accumulator are those of ssd_head.hip's C++ step (bit-identical results).

    python timeviper_amd/devtools/gen_head_step.py  > timeviper_amd/csrc/ssd_head_step.inc
"""
import os
import sys

PT = 5
XROW = 160            # bytes per x row in LDS
RPI = 6               # whole x rows per copy instruction
NXI = 11              # copy instructions per x tile

# ---------------------------------------------------------------- registers owned by the asm body
def A_STATE(ct, i): return 32 * ct + 4 * i           # a[..+3]
def A_YO(ct, ti): return 160 + 16 * ct + 4 * ti
def A_RING(k): return 240 + 4 * (k % 4)

V0 = 76
_v = [V0]
def valloc(n):
    b = _v[0]
    assert n == 1 or b % 2 == 0
    _v[0] += n
    return b
SB = [[valloc(4) for ct in range(PT)] for b in range(2)]
XW = [[valloc(4) for ks in range(2)] for ct in range(PT)]
CBV = [valloc(4) for f in range(6)]
WQ = [[valloc(4) for h in range(2)] for ks in range(2)]
EV = [valloc(2) for ti in range(4)]
CA = [valloc(1) for q in range(4)]
BA = [valloc(1) for m in range(4)]
XT = valloc(1)
XV = valloc(1)
XS = valloc(1)                            # the x / y tile as 16-byte pieces (lane = piece): the row stores of y
_pad = valloc(1)
XRP = [valloc(4) for k in range(2)]       # epilogue: x of tiles in flight; store data
SD = [valloc(4) for k in range(2)]        # store data, two sets
WQ2 = [[XRP[0], XRP[1]], [SD[0], SD[1]]]   # reset steps, phase A: the second set of weights (registers of phase B)
T8 = [valloc(8) for k in range(1)][0]     # accumulator reads (snap / epilogue)
U8 = valloc(8)                            # unpacked halves
OUT = [valloc(4) for k in range(2)]
assert _v[0] == 256, _v[0]

S_M0, S_CP, S_Y, S_D, S_T = 80, 82, 84, 86, 88      # s80 saved m0, s[82:83] copy base, s[84:85] y rows, s[86:87] D, s[88:89] temp

FLAG_REBASE, FLAG_RESET, FLAG_STD, FLAG_COPY = 0, 1, 2, 3
FLAG_SOFTPLUS = 8


def vr(b, n=1): return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"
def ar(b, n=1): return f"a{b}" if n == 1 else f"a[{b}:{b + n - 1}]"
def regs(p, b, n=1): return [f"{p}{b + i}" for i in range(n)]


class Op:
    def __init__(self, text, kind, r=(), w=(), lds_def=None, lds_use=(), vm_def=None, vm_use=(), cost=None, glue=False):
        self.text, self.kind, self.r, self.w = text, kind, list(r), list(w)
        self.lds_def, self.lds_use, self.vm_def, self.vm_use = lds_def, list(lds_use), vm_def, list(vm_use)
        self.glue = glue
        self.cost = cost if cost is not None else {"mfma": 8, "valu": 4, "lds": 4, "vmem": 8, "salu": 1, "dma": 8}.get(kind, 1)


# required wait states between a producer of a register and a consumer
def need_states(pk, ck, is_write):
    if pk == "mfma":
        if ck == "mfmaC": return 0             # whole accumulate chain
        return 19                              # any other reader / writer of an MFMA result (covers 16-pass)
    if pk == "valu":
        if ck in ("mfma", "mfmaC"): return 2
        if ck == "perm": return 2
        return 0
    if pk == "storedata" and is_write: return 2
    return 0


class Emitter:
    def __init__(self):
        self.lines = []
        self.pos = 0
        self.wr = {}            # reg -> (pos, kind) of the last write
        self.rd_store = {}      # reg -> pos of a store reading it
        self.lds_seq, self.lds_done, self.lds_idx = 0, 0, {}
        self.vm_seq, self.vm_done, self.vm_idx = 0, 0, {}
        self.vm_guarded = False
        self.uid = ""
        self.stats = {}

    def raw(self, text, states=1):
        self.lines.append(text)
        self.pos += states

    def nop(self, n):
        while n > 0:
            k = min(n, 16)
            self.raw(f"s_nop {k - 1}", k)
            self.stats["nop"] = self.stats.get("nop", 0) + k
            n -= k

    def emit(self, op):
        # memory waits
        pend = [self.lds_idx[t] for t in op.lds_use if self.lds_done <= self.lds_idx[t]]
        if pend:                       # one wait for the youngest of the operations this instruction needs
            n = min(self.lds_seq - 1 - max(pend), 15)
            self.raw(f"s_waitcnt lgkmcnt({n})")
            self.lds_done = self.lds_seq - n
        for t in op.vm_use:
            i = self.vm_idx[t]
            if self.vm_done <= i:
                n = min(self.vm_seq - 1 - i, 63)
                if n > 0 and not self.vm_guarded:
                    # the counts assume the copies of chunk c + 1 were issued; without them (FLAG_COPY clear) wait for everything
                    self.raw(f"s_bitcmp1_b32 %[flags], {FLAG_COPY}")
                    self.raw(f"s_cbranch_scc1 .Lhs_cbw{self.uid}_%=")
                    self.raw("s_waitcnt vmcnt(0)")
                    self.raw(f".Lhs_cbw{self.uid}_%=:", 0)
                    self.vm_guarded = True
                self.raw(f"s_waitcnt vmcnt({n})")
                self.vm_done = self.vm_seq - n
        # wait states
        need = 0
        ck = op.kind
        for reg in op.r:
            if reg in self.wr:
                p, pk = self.wr[reg]
                c = ck
                if ck == "mfma" and reg in op.w: c = "mfmaC"
                need = max(need, p + need_states(pk, c, False) + 1 - self.pos)
        for reg in op.w:
            if reg in self.wr and not (ck == "mfma" and reg in op.r):
                p, pk = self.wr[reg]
                if pk == "mfma": need = max(need, p + 19 + 1 - self.pos)
            if reg in self.rd_store:
                need = max(need, self.rd_store[reg] + 2 + 1 - self.pos)
        if need > 0: self.nop(need)
        self.lines.append(op.text)
        kind_w = {"mfma": "mfma", "valu": "valu", "perm": "valu"}.get(ck, "other")
        for reg in op.w: self.wr[reg] = (self.pos, kind_w)
        if ck == "store":
            for reg in op.r: self.rd_store[reg] = self.pos
        self.pos += 1
        if op.lds_def is not None or ck in ("lds",):      # ("lds_nc": an LDS write inside an optional region, not counted: waits stay safe)
            if op.lds_def is not None: self.lds_idx[op.lds_def] = self.lds_seq
            self.lds_seq += 1
        if op.vm_def is not None or ck in ("vmem", "dma", "store"):
            if op.vm_def is not None: self.vm_idx[op.vm_def] = self.vm_seq
            self.vm_seq += 1
        self.stats[ck] = self.stats.get(ck, 0) + 1

    def drain_lds(self):
        if self.lds_done < self.lds_seq:
            self.raw("s_waitcnt lgkmcnt(0)")
            self.lds_done = self.lds_seq


STAMPS = False
XNT = "--xnt" in sys.argv      # experiment: the x copies (bytes read once, by one wave) with the non-temporal hint
NOSTORE = False  # timing experiment: no y stores
PK = True        # packed f32 multiplies / FMAs (False: scalar ones)
def stamp(em, k):
    """debug builds: cycles since the previous stamp are added to accumulator k (s_memtime; drains the LDS queue)"""
    if not STAMPS: return
    em.raw("s_memtime s[90:91]")
    em.raw("s_waitcnt lgkmcnt(0)")
    em.lds_done = em.lds_seq
    em.raw("s_sub_u32 s92, s90, %[stl]")
    em.raw(f"s_add_u32 %[st{k}], %[st{k}], s92")
    em.raw("s_mov_b32 %[stl], s90")

# offsets of HeadVec's fields (bytes)
VEC_CS, VEC_DTV, VEC_UT, VEC_WTS, VEC_WTD, VEC_ECS, VEC_WS, VEC_ONE = 0, 256, 512, 768, 1024, 1536, 1792, 2304
FLAG_DEAD0 = 4                   # flags bit 4 + ti: a standard step's row factors of t-tile ti are all zero


def std_scale_ops(ti):
    """standard step: the accumulators of t-tile ti get their row factor 2^(cs_t + E) (one glued group; a t-tile whose
    factors have all underflowed is zeroed instead)"""
    F = T8 + 4
    ops = [Op(f"ds_read_b32 {vr(F)}, %[alc] offset:{VEC_ECS + 64 * ti}", "lds", w=regs("v", F), lds_def=f"sev{ti}"),
           Op(f"s_bitcmp1_b32 %[flags], {FLAG_DEAD0 + ti}", "salu"),
           Op(f"s_cbranch_scc1 .Lhs_dead{ti}_%=", "salu")]
    for ct in range(PT):
        y = A_YO(ct, ti)
        for r in range(4): ops.append(Op(f"v_accvgpr_read_b32 {vr(T8 + r)}, {ar(y + r)}", "valu", r=regs("a", y + r), w=regs("v", T8 + r)))
        for r in range(4): ops.append(Op(f"v_mul_f32 {vr(T8 + r)}, {vr(F)}, {vr(T8 + r)}", "valu", r=regs("v", T8 + r) + regs("v", F), w=regs("v", T8 + r),
                                         lds_use=[f"sev{ti}"]))
        for r in range(4): ops.append(Op(f"v_accvgpr_write_b32 {ar(y + r)}, {vr(T8 + r)}", "valu", r=regs("v", T8 + r), w=regs("a", y + r)))
    ops.append(Op(f"s_branch .Lhs_scaled{ti}_%=", "salu"))
    ops.append(Op(f".Lhs_dead{ti}_%=:", "label", cost=0))
    ops.append(Op("s_nop 15", "salu"))          # (the other path's wait states for the MFMAs' results are not known to have passed here)
    ops.append(Op("s_nop 3", "salu"))
    for ct in range(PT):
        y = A_YO(ct, ti)
        for r in range(4): ops.append(Op(f"v_accvgpr_write_b32 {ar(y + r)}, 0", "valu", w=regs("a", y + r)))
    ops.append(Op(f".Lhs_scaled{ti}_%=:", "label", cost=0))
    for o in ops[:-1]: o.glue = True
    return ops


def std_mask_ops():
    """Standard step (a chunk that decays by more than 2^-199: no single frame holds its weights), as in ssd_head.hip's
    `ustd_step`: the per-head mask M = CB .* 2^(cs_t - cs_s) dt_s [s <= t] replaces C.B^T in its registers (diagonal
    16x16 blocks one exponential per element, the others separable around the first token of their t-tile: ut[t] ws[s]).
    Temporaries: registers of phase B."""
    ops = []
    E_D, E_S = XRP[0], SD[0]
    CV, DV = OUT[0], OUT[1]
    CST, UU, TA, TB, UN = EV[0], EV[0] + 1, EV[1], EV[1] + 1, EV[2]
    ops.append(Op("s_mov_b32 s88, 0", "salu"))          # s[88:89] = lanes 32..63 (hi = kq >> 1)
    ops.append(Op("s_mov_b32 s89, -1", "salu"))
    ops.append(Op(f"v_mov_b32 {vr(TA)}, 0xff800000", "valu", w=regs("v", TA)))      # -inf
    ops.append(Op(f"v_mov_b32 {vr(TB)}, 0", "valu", w=regs("v", TB)))
    tagn = [0]

    def diag(t_off, s_off, e):
        """e[j] = 2^(cs_t - cs_(s0 + j)) dt_(s0 + j) for s0 + j <= t, else 0;  t = t_off + 16 hi + lc, s0 = s_off + 8 kq"""
        tagn[0] += 1
        tg = f"d{tagn[0]}"
        ops.append(Op(f"ds_read_b32 {vr(CST)}, %[at] offset:{VEC_CS + 4 * t_off}", "lds", w=regs("v", CST), lds_def=tg + "c"))
        for hh in range(2):
            ops.append(Op(f"ds_read_b128 {vr(CV, 4)}, %[as] offset:{VEC_CS + 4 * s_off + 16 * hh}", "lds", w=regs("v", CV, 4), lds_def=tg + f"v{hh}"))
            ops.append(Op(f"ds_read_b128 {vr(DV, 4)}, %[as] offset:{VEC_DTV + 4 * s_off + 16 * hh}", "lds", w=regs("v", DV, 4), lds_def=tg + f"w{hh}"))
            for j in range(4):
                d = e + 4 * hh + j
                ops.append(Op(f"v_sub_f32 {vr(d)}, {vr(CST)}, {vr(CV + j)}", "valu", r=regs("v", CST) + regs("v", CV + j), w=regs("v", d),
                              lds_use=[tg + "c", tg + f"v{hh}"]))
                ops.append(Op(f"v_cmp_ge_i32 vcc, %[d0], {4 * hh + j}", "valu", glue=True))
                ops.append(Op("s_nop 1", "salu", glue=True))
                ops.append(Op(f"v_cndmask_b32 {vr(d)}, {vr(TA)}, {vr(d)}, vcc", "valu", r=regs("v", d) + regs("v", TA), w=regs("v", d)))
                ops.append(Op(f"v_exp_f32 {vr(d)}, {vr(d)}", "valu", r=regs("v", d), w=regs("v", d), cost=8, glue=True))
                ops.append(Op("s_nop 0", "salu", glue=True))
                ops.append(Op(f"v_mul_f32 {vr(d)}, {vr(d)}, {vr(DV + j)}", "valu", r=regs("v", d) + regs("v", DV + j), w=regs("v", d), lds_use=[tg + f"w{hh}"]))

    def sepf(t_off, w_operand, w_off, e):
        """e[j] = ut[t_off + lc] ws[w_off + j]"""
        tagn[0] += 1
        tg = f"s{tagn[0]}"
        ops.append(Op(f"ds_read_b32 {vr(UU)}, %[alc] offset:{VEC_UT + 4 * t_off}", "lds", w=regs("v", UU), lds_def=tg + "u"))
        for hh in range(2):
            ops.append(Op(f"ds_read_b128 {vr(CV, 4)}, %[{w_operand}] offset:{VEC_WS + 4 * w_off + 16 * hh}", "lds", w=regs("v", CV, 4), lds_def=tg + f"v{hh}"))
            for j in range(4):
                d = e + 4 * hh + j
                ops.append(Op(f"v_mul_f32 {vr(d)}, {vr(UU)}, {vr(CV + j)}", "valu", r=regs("v", UU) + regs("v", CV + j), w=regs("v", d), lds_use=[tg + "u", tg + f"v{hh}"]))

    def select(e_lo, e_hi, dst):
        """dst[j] = hi ? e_hi[j] : e_lo[j]   (e_hi None = 0)"""
        for j in range(8):
            hi_ = vr(TB) if e_hi is None else vr(e_hi + j)
            ops.append(Op(f"v_cndmask_b32 {vr(dst + j)}, {vr(e_lo + j)}, {hi_}, s[88:89]", "valu", r=regs("v", e_lo + j), w=regs("v", dst + j)))

    def apply(f, fac):
        """cbv[f] <- bf16(CB_f .* fac)"""
        for jp in range(4):
            c = CBV[f] + jp
            ops.append(Op(f"v_lshlrev_b32 {vr(UN)}, 16, {vr(c)}", "valu", r=regs("v", c), w=regs("v", UN), vm_use=[f"cb{f}"]))
            ops.append(Op(f"v_and_b32 {vr(UN + 1)}, 0xffff0000, {vr(c)}", "valu", r=regs("v", c), w=regs("v", UN + 1)))
            ops.append(Op(f"v_mul_f32 {vr(UN)}, {vr(UN)}, {vr(fac + 2 * jp)}", "valu", r=regs("v", UN) + regs("v", fac + 2 * jp), w=regs("v", UN)))
            ops.append(Op(f"v_mul_f32 {vr(UN + 1)}, {vr(UN + 1)}, {vr(fac + 2 * jp + 1)}", "valu", r=regs("v", UN + 1) + regs("v", fac + 2 * jp + 1), w=regs("v", UN + 1)))
            ops.append(Op(f"v_cvt_pk_bf16_f32 {vr(c)}, {vr(UN)}, {vr(UN + 1)}", "valu", r=regs("v", UN, 2), w=regs("v", c)))

    diag(0, 0, E_D)                                 # diagonal blocks of fragments (0,0) [hi = 0] and (1,0) [hi = 1]
    select(E_D, None, E_S)
    apply(0, E_S)
    sepf(16, "as15", 0, E_S)                        # block (1,0)
    select(E_S, E_D, E_S)
    apply(1, E_S)
    sepf(32, "as", 16, E_S)                         # blocks (2,0), (2,1)
    apply(2, E_S)
    diag(32, 32, E_D)                               # diagonal blocks of fragments (2,1) [hi = 0] and (3,1) [hi = 1]
    select(E_D, None, E_S)
    apply(3, E_S)
    sepf(48, "as", 48, E_S)                         # blocks (3,0), (3,1)
    apply(4, E_S)
    sepf(48, "as15", 80, E_S)                       # block (3,2)
    select(E_S, E_D, E_S)
    apply(5, E_S)
    return ops


def prep_ops(uid):
    """The vectors of chunk c + 1 (what ssd_head.hip's `prep` lambda does, in its compiled order of operations): dt -> softplus ->
    clamp, inclusive scan of dt A over the wave (DPP), the mode decision, row factors, weights, and a standard step's separable
    factors — written into the wave's vectors in LDS (every reader of chunk c's has finished before phase B).  Results that the
    next step's scalar set-up needs come back in SGPR outputs.  Temporaries: the registers of the state's bf16 copy."""
    ops = []
    P = [SB[0][0] + i for i in range(40)]
    X, T1, T2, T3, T4, T5, D_, CS, CS2, EE, CL2, ECL, ROW, A0, A1, A2, ARG, PV, K32, MSH = P[:20]
    VALID, MA, MB, MC = 72, 74, 76, 78          # SGPR pairs
    SC, SCL, SMODE, SDEAD = 94, 95, 96, 97

    def V(text, w=(), r=(), **kw): ops.append(Op(text, "valu", r=[f"v{x}" for x in r], w=[f"v{x}" for x in w], **kw))
    def S(text, **kw): ops.append(Op(text, "salu", **kw))
    def G(n):           # glue the last n ops to their successors (a compare and its consumer, a transcendental and its wait state)
        for o in ops[-n:]: o.glue = True

    V(f"v_cmp_gt_i32 vcc, %[lrem1], %[lane]"); G(1)
    S(f"s_mov_b64 s[{VALID}:{VALID + 1}], vcc")
    V(f"v_lshlrev_b32 {vr(X)}, 16, %[dtin]", w=[X])
    V(f"v_add_f32 {vr(X)}, %[bias], {vr(X)}", w=[X], r=[X])
    V(f"v_mul_f32 {vr(T1)}, 0x3fb8aa3b, {vr(X)}", w=[T1], r=[X])
    V(f"v_exp_f32 {vr(T2)}, {vr(T1)}", w=[T2], r=[T1], cost=8); G(1)
    S("s_nop 0")
    V(f"v_add_f32 {vr(T1)}, 1.0, {vr(T2)}", w=[T1], r=[T2])
    S(f"s_mov_b32 s{SC}, 0x800000")
    V(f"v_cmp_gt_f32 vcc, s{SC}, {vr(T1)}", r=[T1]); G(1)
    V(f"v_cndmask_b32 {vr(T3)}, 0, 32, vcc", w=[T3]); G(1)
    V(f"v_mov_b32 {vr(K32)}, 0x41b17218", w=[K32]); G(1)
    V(f"v_cndmask_b32 {vr(K32)}, 0, {vr(K32)}, vcc", w=[K32], r=[K32])
    V(f"v_ldexp_f32 {vr(T1)}, {vr(T1)}, {vr(T3)}", w=[T1], r=[T1, T3])
    V(f"v_log_f32 {vr(T1)}, {vr(T1)}", w=[T1], r=[T1], cost=8); G(1)
    S("s_nop 0")
    V(f"v_mul_f32 {vr(T4)}, 0x3f317217, {vr(T1)}", w=[T4], r=[T1])
    S(f"s_mov_b32 s{SC}, 0x3f317217")
    V(f"v_fma_f32 {vr(T5)}, {vr(T1)}, s{SC}, -{vr(T4)}", w=[T5], r=[T1, T4])
    V(f"v_fmac_f32 {vr(T5)}, 0x3377d1cf, {vr(T1)}", w=[T5], r=[T5, T1])
    V(f"v_add_f32 {vr(T4)}, {vr(T4)}, {vr(T5)}", w=[T4], r=[T4, T5])
    S(f"s_mov_b32 s{SC}, 0x7f800000")
    V(f"v_cmp_lt_f32 s[{MA}:{MA + 1}], |{vr(T1)}|, s{SC}", r=[T1]); G(1)
    S("s_nop 1"); G(1)
    V(f"v_cndmask_b32 {vr(T1)}, {vr(T1)}, {vr(T4)}, s[{MA}:{MA + 1}]", w=[T1], r=[T1, T4])
    V(f"v_sub_f32 {vr(T1)}, {vr(T1)}, {vr(K32)}", w=[T1], r=[T1, K32])             # log(1 + e)
    V(f"v_mul_f32 {vr(T4)}, -0.5, {vr(T2)}", w=[T4], r=[T2])
    V(f"v_mul_f32 {vr(T4)}, {vr(T2)}, {vr(T4)}", w=[T4], r=[T2, T4])
    V(f"v_add_f32 {vr(T4)}, {vr(T2)}, {vr(T4)}", w=[T4], r=[T2, T4])               # e - e^2 / 2
    S(f"s_mov_b32 s{SC}, 0x3a83126f")
    V(f"v_cmp_ngt_f32 vcc, s{SC}, {vr(T2)}", r=[T2]); G(1)
    V(f"v_cndmask_b32 {vr(T1)}, {vr(T4)}, {vr(T1)}, vcc", w=[T1], r=[T1, T4])
    S(f"s_mov_b32 s{SC}, 0x41a00000")
    V(f"v_cmp_lt_f32 vcc, s{SC}, {vr(X)}", r=[X]); G(1)
    V(f"v_cndmask_b32 {vr(T1)}, {vr(T1)}, {vr(X)}, vcc", w=[T1], r=[T1, X])        # softplus(x)
    S(f"s_bitcmp1_b32 %[flags], {FLAG_SOFTPLUS}"); G(1)
    S(f"s_cselect_b64 s[{MA}:{MA + 1}], -1, 0")
    V(f"v_cndmask_b32 {vr(X)}, {vr(X)}, {vr(T1)}, s[{MA}:{MA + 1}]", w=[X], r=[X, T1])
    V(f"v_max_f32 {vr(X)}, {vr(X)}, {vr(X)}", w=[X], r=[X])
    V(f"v_max_f32 {vr(X)}, %[dtmin], {vr(X)}", w=[X], r=[X])
    V(f"v_min_f32 {vr(X)}, %[dtmax], {vr(X)}", w=[X], r=[X])
    V(f"v_cndmask_b32 {vr(D_)}, 0, {vr(X)}, s[{VALID}:{VALID + 1}]", w=[D_], r=[X])        # d (0 past the end of the sequence)
    # inclusive scan of d A over the 64 lanes
    V(f"v_mul_f32 {vr(CS)}, %[ah], {vr(D_)}", w=[CS], r=[D_])
    for sh in (1, 2, 4, 8):
        S("s_nop 1"); G(1)
        V(f"v_add_f32_dpp {vr(CS)}, {vr(CS)}, {vr(CS)} row_shr:{sh} row_mask:0xf bank_mask:0xf bound_ctrl:1", w=[CS], r=[CS])
    V(f"v_mov_b32 {vr(T1)}, 0", w=[T1])
    S("s_nop 1"); G(2)
    V(f"v_mov_b32_dpp {vr(T1)}, {vr(CS)} row_bcast:15 row_mask:0xa bank_mask:0xf", w=[T1], r=[CS, T1])
    V(f"v_add_f32 {vr(CS)}, {vr(CS)}, {vr(T1)}", w=[CS], r=[CS, T1])
    V(f"v_mov_b32 {vr(T1)}, 0", w=[T1])
    S("s_nop 1"); G(2)
    V(f"v_mov_b32_dpp {vr(T1)}, {vr(CS)} row_bcast:31 row_mask:0xc bank_mask:0xf", w=[T1], r=[CS, T1])
    V(f"v_add_f32 {vr(CS)}, {vr(CS)}, {vr(T1)}", w=[CS], r=[CS, T1])
    V(f"v_mul_f32 {vr(CS2)}, 0x3fb8aa3b, {vr(CS)}", w=[CS2], r=[CS])
    S("s_nop 0"); G(1)
    V(f"v_readlane_b32 s{SCL}, {vr(CS)}, 63", r=[CS])
    V(f"v_mov_b32 {vr(EE)}, %[ein]", w=[EE])
    V(f"v_sub_f32 {vr(T3)}, 0x42c80000, {vr(EE)}", w=[T3], r=[EE])
    V(f"v_floor_f32 {vr(T3)}, {vr(T3)}", w=[T3], r=[T3])
    V(f"v_mov_b32 {vr(T1)}, 0x3fb8aa3b", w=[T1])
    V(f"v_mul_f32 {vr(CL2)}, s{SCL}, {vr(T1)}", w=[CL2], r=[T1])                    # log2 decay of the chunk
    V(f"v_add_f32 {vr(ECL)}, {vr(EE)}, {vr(CL2)}", w=[ECL], r=[EE, CL2])
    # mode: 0 floating (-(E + cl2) <= RMAX), 1 re-base first (-cl2 <= 2 RMAX - 1), 2 standard step
    S(f"s_mov_b32 s{SC}, 0xc3470000")
    V(f"v_cmp_le_f32 s[{MA}:{MA + 1}], s{SC}, {vr(CL2)}", r=[CL2])
    S(f"s_mov_b32 s{SC}, 0xc2c80000")
    V(f"v_cmp_nle_f32 vcc, s{SC}, {vr(ECL)}", r=[ECL]); G(1)
    S("s_nop 1"); G(1)
    S(f"s_and_b64 s[{MB}:{MB + 1}], s[{MA}:{MA + 1}], vcc"); G(1)          # mode 1
    S(f"s_andn2_b64 s[{MC}:{MC + 1}], exec, vcc")                        # mode 0 = not (mode != 0)
    S(f"s_or_b64 s[{MC}:{MC + 1}], s[{MC}:{MC + 1}], s[{MA}:{MA + 1}]")    # mode != 2 = mode 0 or (-cl2 <= 199)
    V(f"v_cndmask_b32 {vr(MSH)}, 0, {vr(T3)}, s[{MB}:{MB + 1}]", w=[MSH], r=[T3])   # m = floor(RMAX - E) when re-basing, else 0
    V(f"v_add_f32 {vr(EE)}, {vr(EE)}, {vr(MSH)}", w=[EE], r=[EE, MSH])                # the frame in which this chunk reads the state
    V(f"v_add_f32 {vr(ROW)}, {vr(CS2)}, {vr(EE)}", w=[ROW], r=[CS2, EE])
    V(f"v_exp_f32 {vr(ROW)}, {vr(ROW)}", w=[ROW], r=[ROW], cost=8); G(1)
    S("s_nop 0")
    ops.append(Op(f"ds_write2st64_b32 %[vvec], {vr(CS2)}, {vr(D_)} offset1:1", "lds", r=[f"v{CS2}", f"v{D_}"]))
    ops.append(Op(f"ds_write_b32 %[vvec], {vr(ROW)} offset:{VEC_ECS}", "lds", r=[f"v{ROW}"]))
    # reset: the chunk decays by more than 2^-64 and is not a standard step
    S(f"s_mov_b32 s{SC}, 0xc2800000")
    V(f"v_cmp_ge_f32 s[{MA}:{MA + 1}], s{SC}, {vr(CL2)}", r=[CL2]); G(1)
    S("s_nop 1"); G(1)
    S(f"s_and_b64 s[{MA}:{MA + 1}], s[{MA}:{MA + 1}], s[{MC}:{MC + 1}]")          # rst
    # weights of the state update: 2^arg d, arg = standard: cl2 - cs2; reset: cl2 - cs2 - RMAX; else -cs2 - E
    V(f"v_sub_f32 {vr(A0)}, -{vr(CS2)}, {vr(EE)}", w=[A0], r=[CS2, EE])
    V(f"v_sub_f32 {vr(A1)}, {vr(CL2)}, {vr(CS2)}", w=[A1], r=[CL2, CS2])
    V(f"v_add_f32 {vr(A2)}, 0xc2c80000, {vr(A1)}", w=[A2], r=[A1])
    V(f"v_cndmask_b32 {vr(ARG)}, {vr(A0)}, {vr(A2)}, s[{MA}:{MA + 1}]", w=[ARG], r=[A0, A2])
    V(f"v_cndmask_b32 {vr(ARG)}, {vr(A1)}, {vr(ARG)}, s[{MC}:{MC + 1}]", w=[ARG], r=[A1, ARG])
    V(f"v_exp_f32 {vr(ARG)}, {vr(ARG)}", w=[ARG], r=[ARG], cost=8); G(1)
    S("s_nop 0")
    V(f"v_mul_f32 {vr(ARG)}, {vr(D_)}, {vr(ARG)}", w=[ARG], r=[D_, ARG])
    ops.append(Op(f"ds_write_b32 %[vvec], {vr(ARG)} offset:{VEC_WTS}", "lds", r=[f"v{ARG}"]))
    V(f"v_exp_f32 {vr(A0)}, {vr(A0)}", w=[A0], r=[A0], cost=8); G(1)              # reset steps: Ydiag's weights (the old frame's)
    S("s_nop 0")
    V(f"v_mul_f32 {vr(A0)}, {vr(D_)}, {vr(A0)}", w=[A0], r=[D_, A0])
    V(f"v_lshl_add_u32 {vr(T1)}, %[par1], 8, %[vvec]", w=[T1])
    ops.append(Op(f"ds_write_b32 {vr(T1)}, {vr(A0)} offset:{VEC_WTD}", "lds", r=[f"v{T1}", f"v{A0}"]))
    # frame after this chunk: standard 0, reset RMAX, else E + cl2;  decay total
    V(f"v_add_f32 {vr(T2)}, {vr(CL2)}, {vr(EE)}", w=[T2], r=[CL2, EE])
    V(f"v_mov_b32 {vr(T3)}, 0x42c80000", w=[T3])
    V(f"v_cndmask_b32 {vr(T2)}, {vr(T2)}, {vr(T3)}, s[{MA}:{MA + 1}]", w=[T2], r=[T2, T3])
    V(f"v_cndmask_b32 {vr(T2)}, 0, {vr(T2)}, s[{MC}:{MC + 1}]", w=[T2], r=[T2])
    V(f"v_mov_b32 {vr(T4)}, %[dtot]", w=[T4])
    V(f"v_add_f32 {vr(T4)}, s{SCL}, {vr(T4)}", w=[T4], r=[T4])
    # dead t-tiles (every row factor zero), mode bits
    V(f"v_cmp_neq_f32 vcc, 0, {vr(ROW)}", r=[ROW]); G(1)
    S("s_nop 1"); G(1)
    S(f"s_and_b32 s{SC}, vcc_lo, 0xffff"); G(1)
    S(f"s_cmp_eq_u32 s{SC}, 0"); G(1)
    S(f"s_cselect_b32 s{SDEAD}, 16, 0"); G(1)
    S(f"s_lshr_b32 s{SC}, vcc_lo, 16"); G(1)
    S(f"s_cmp_eq_u32 s{SC}, 0"); G(1)
    S(f"s_cselect_b32 s{SC}, 32, 0"); G(1)
    S(f"s_or_b32 s{SDEAD}, s{SDEAD}, s{SC}"); G(1)
    S(f"s_and_b32 s{SC}, vcc_hi, 0xffff"); G(1)
    S(f"s_cmp_eq_u32 s{SC}, 0"); G(1)
    S(f"s_cselect_b32 s{SC}, 64, 0"); G(1)
    S(f"s_or_b32 s{SDEAD}, s{SDEAD}, s{SC}"); G(1)
    S(f"s_lshr_b32 s{SC}, vcc_hi, 16"); G(1)
    S(f"s_cmp_eq_u32 s{SC}, 0"); G(1)
    S(f"s_cselect_b32 s{SC}, 128, 0"); G(1)
    S(f"s_or_b32 s{SDEAD}, s{SDEAD}, s{SC}")
    S(f"s_cmp_lg_u64 s[{MB}:{MB + 1}], 0"); G(1)
    S(f"s_cselect_b32 s{SMODE}, 1, 0")                                     # bit 0: re-base
    S(f"s_cmp_lg_u64 s[{MA}:{MA + 1}], 0"); G(1)
    S(f"s_cselect_b32 s{SC}, 2, 0"); G(1)
    S(f"s_or_b32 s{SMODE}, s{SMODE}, s{SC}")                               # bit 1: reset
    S(f"s_cmp_lg_u64 s[{MC}:{MC + 1}], 0"); G(1)
    S(f"s_cselect_b32 s{SC}, 0, 6"); G(1)
    S(f"s_or_b32 s{SMODE}, s{SMODE}, s{SC}"); G(1)                          # standard: bits 1 and 2
    S(f"s_or_b32 %[oflags], s{SMODE}, s{SDEAD}")
    V(f"v_cvt_i32_f32 {vr(T3)}, -{vr(MSH)}", w=[T3], r=[MSH])
    S("s_nop 0"); G(1)
    V(f"v_readfirstlane_b32 %[osh], {vr(T3)}", r=[T3])
    V(f"v_readfirstlane_b32 %[oe], {vr(T2)}", r=[T2])
    V(f"v_readfirstlane_b32 %[odtot], {vr(T4)}", r=[T4])
    V(f"v_readfirstlane_b32 %[ocl2], {vr(CL2)}", r=[CL2])
    # standard steps: separable factors of the off-diagonal mask blocks (pivot = first token of a t-tile); LDS writes not counted
    S(f"s_cmp_lg_u64 s[{MC}:{MC + 1}], 0")
    S(f"s_cbranch_scc1 .Lhs_prep_nostd{uid}_%=")
    k0 = len(ops) - 2
    for j, ln in enumerate((0, 16, 32, 48)):
        V(f"v_readlane_b32 s{90 + j}, {vr(CS2)}, {ln}", r=[CS2])
    V(f"v_mov_b32 {vr(PV)}, s93", w=[PV])
    for j, bound in ((2, 48), (1, 32), (0, 16)):
        V(f"v_cmp_gt_i32 vcc, {bound}, %[lane]")
        V(f"v_mov_b32 {vr(T1)}, s{90 + j}", w=[T1])
        V(f"v_cndmask_b32 {vr(PV)}, {vr(PV)}, {vr(T1)}, vcc", w=[PV], r=[PV, T1])
    V(f"v_sub_f32 {vr(T1)}, {vr(CS2)}, {vr(PV)}", w=[T1], r=[CS2, PV])
    V(f"v_min_f32 {vr(T1)}, 0, {vr(T1)}", w=[T1], r=[T1])
    V(f"v_exp_f32 {vr(T1)}, {vr(T1)}", w=[T1], r=[T1])
    S("s_nop 0")
    ops.append(Op(f"ds_write_b32 %[vvec], {vr(T1)} offset:{VEC_UT}", "lds_nc", r=[f"v{T1}"]))
    for j, (bound, woff) in enumerate(((16, 0), (32, 16), (48, 48))):
        V(f"v_sub_f32 {vr(T1)}, s{91 + j}, {vr(CS2)}", w=[T1], r=[CS2])
        V(f"v_min_f32 {vr(T1)}, 0, {vr(T1)}", w=[T1], r=[T1])
        V(f"v_exp_f32 {vr(T1)}, {vr(T1)}", w=[T1], r=[T1])
        S("s_nop 0")
        V(f"v_mul_f32 {vr(T1)}, {vr(D_)}, {vr(T1)}", w=[T1], r=[D_, T1])
        V(f"v_cmp_gt_i32 vcc, {bound}, %[lane]")
        S("s_nop 1")
        S("s_mov_b64 exec, vcc")
        S("s_nop 0")
        ops.append(Op(f"ds_write_b32 %[vvec], {vr(T1)} offset:{VEC_WS + 4 * woff}", "lds_nc", r=[f"v{T1}"]))
        S("s_mov_b64 exec, -1")
    ops.append(Op(f".Lhs_prep_nostd{uid}_%=:", "label", cost=0))
    for o in ops[k0:-1]: o.glue = True
    return ops



class Task:
    def __init__(self, name, ops, after=-1, deadline=10 ** 9, prio=None):
        self.name, self.ops, self.after, self.deadline = name, list(ops), after, deadline
        self.prio = deadline if prio is None else prio


# ---------------------------------------------------------------- building blocks
def mfma(dst, a, b, c, a_is_acc=False, b_is_acc=False):
    """dst/c accumulator tile (base AGPR; c None = 0); a, b operand bases (VGPR unless flagged)"""
    at = ar(a, 4) if a_is_acc else vr(a, 4)
    bt = ar(b, 4) if b_is_acc else vr(b, 4)
    ct = "0" if c is None else ar(c, 4)
    r = regs("a" if a_is_acc else "v", a, 4) + regs("a" if b_is_acc else "v", b, 4) + ([] if c is None else regs("a", c, 4))
    return Op(f"v_mfma_f32_16x16x32_bf16 {ar(dst, 4)}, {at}, {bt}, {ct}", "mfma", r=r, w=regs("a", dst, 4))


def snap_ops(q, ct, dst, tmp):
    """bf16 copy of state rows 32 q + 8 kq + 0..7 of column tile ct (state tiles 2q, 2q + 1) -> v[dst:dst+3]"""
    ops = []
    for ii in range(2):
        s = A_STATE(ct, 2 * q + ii)
        for r in range(4):
            ops.append(Op(f"v_accvgpr_read_b32 {vr(tmp + 4 * ii + r)}, {ar(s + r)}", "valu", r=regs("a", s + r), w=regs("v", tmp + 4 * ii + r)))
    for ii in range(2):
        for h in range(2):
            t = tmp + 4 * ii + 2 * h
            ops.append(Op(f"v_cvt_pk_bf16_f32 {vr(dst + 2 * ii + h)}, {vr(t)}, {vr(t + 1)}", "valu", r=regs("v", t, 2), w=regs("v", dst + 2 * ii + h)))
    return ops


def xw_read_ops(ct, ks, tag):
    d = XW[ct][ks]
    off = 32 * ct + ks * 32 * XROW
    return [Op(f"ds_read_b64_tr_b16 {vr(d, 2)}, {vr(XT)} offset:{off}", "lds", r=regs("v", XT), w=regs("v", d, 2), lds_def=f"{tag}{ct}{ks}a"),
            Op(f"ds_read_b64_tr_b16 {vr(d + 2, 2)}, {vr(XT)} offset:{off + 4 * XROW}", "lds", r=regs("v", XT), w=regs("v", d + 2, 2), lds_def=f"{tag}{ct}{ks}b")]


def xw_comp_ops(ct, ks, tag, wtag):
    """x~ = w_s x on fragment (ct, ks) in place: element pair e of the fragment times wq[ks][e >> 1][2 (e & 1) ..]"""
    d = XW[ct][ks]
    ops = []
    for e in range(4):
        t = U8 + 2 * (e & 1) + 4 * (ks & 1)
        w = WQ[ks][e >> 1] + 2 * (e & 1)
        use = [f"{tag}{ct}{ks}{'a' if e < 2 else 'b'}"]
        ops.append(Op(f"v_lshlrev_b32 {vr(t)}, 16, {vr(d + e)}", "valu", r=regs("v", d + e), w=regs("v", t), lds_use=use))
        ops.append(Op(f"v_and_b32 {vr(t + 1)}, 0xffff0000, {vr(d + e)}", "valu", r=regs("v", d + e), w=regs("v", t + 1)))
        if PK:
            ops.append(Op(f"v_pk_mul_f32 {vr(t, 2)}, {vr(w, 2)}, {vr(t, 2)}", "valu", r=regs("v", t, 2) + regs("v", w, 2), w=regs("v", t, 2),
                          lds_use=[f"{wtag}{ks}{e >> 1}"]))
        else:
            for k in range(2):
                ops.append(Op(f"v_mul_f32 {vr(t + k)}, {vr(w + k)}, {vr(t + k)}", "valu", r=regs("v", t + k) + regs("v", w + k), w=regs("v", t + k),
                              lds_use=[f"{wtag}{ks}{e >> 1}"]))
        ops.append(Op(f"v_cvt_pk_bf16_f32 {vr(d + e)}, {vr(t)}, {vr(t + 1)}", "valu", r=regs("v", t, 2), w=regs("v", d + e)))
    return ops


def wq_read_ops(vw_operand, wtag, dst=None):
    dst = WQ if dst is None else dst
    ops = []
    for ks in range(2):
        for h in range(2):
            ops.append(Op(f"ds_read_b128 {vr(dst[ks][h], 4)}, %[{vw_operand}] offset:{(32 * ks + 4 * h) * 4}", "lds", w=regs("v", dst[ks][h], 4),
                          lds_def=f"{wtag}{ks}{h}"))
    return ops


def xw_dual_ops(ct, ks, tag, wtag, wtag2, dst2):
    """fragment (ct, ks) of x times two sets of weights from one unpacking: WQ2 -> v[dst2 ..], WQ -> in place"""
    d = XW[ct][ks]
    ops = []
    for e in range(4):
        t = U8 + 4 * (e & 1)
        w, w2 = WQ[ks][e >> 1] + 2 * (e & 1), WQ2[ks][e >> 1] + 2 * (e & 1)
        use = [f"{tag}{ct}{ks}{'a' if e < 2 else 'b'}"]
        ops.append(Op(f"v_lshlrev_b32 {vr(t)}, 16, {vr(d + e)}", "valu", r=regs("v", d + e), w=regs("v", t), lds_use=use))
        ops.append(Op(f"v_and_b32 {vr(t + 1)}, 0xffff0000, {vr(d + e)}", "valu", r=regs("v", d + e), w=regs("v", t + 1)))
        for k in range(2):
            ops.append(Op(f"v_mul_f32 {vr(t + 2 + k)}, {vr(w2 + k)}, {vr(t + k)}", "valu", r=regs("v", t + k) + regs("v", w2 + k), w=regs("v", t + 2 + k),
                          lds_use=[f"{wtag2}{ks}{e >> 1}"]))
        ops.append(Op(f"v_cvt_pk_bf16_f32 {vr(dst2 + e)}, {vr(t + 2)}, {vr(t + 3)}", "valu", r=regs("v", t + 2, 2), w=regs("v", dst2 + e)))
        for k in range(2):
            ops.append(Op(f"v_mul_f32 {vr(t + k)}, {vr(w + k)}, {vr(t + k)}", "valu", r=regs("v", t + k) + regs("v", w + k), w=regs("v", t + k),
                          lds_use=[f"{wtag}{ks}{e >> 1}"]))
        ops.append(Op(f"v_cvt_pk_bf16_f32 {vr(d + e)}, {vr(t)}, {vr(t + 1)}", "valu", r=regs("v", t, 2), w=regs("v", d + e)))
    return ops


def copy_group_ops(name, base_operand, add_operand, m0_expr_ops, voffs, ioffs):
    """one M0 set-up + up to four LDS-DMA pieces, skipped as a whole when FLAG_COPY is clear"""
    ops = [Op(f"s_bitcmp1_b32 %[flags], {FLAG_COPY}", "salu"),
           Op(f"s_cbranch_scc0 .Lhs_{name}_%=", "salu")]
    ops.append(Op(f"s_mov_b64 s[{S_CP}:{S_CP + 1}], %[{base_operand}]", "salu", w=[f"s{S_CP}", f"s{S_CP + 1}"]))
    for k in range(add_operand[1]):
        ops.append(Op(f"s_add_u32 s{S_CP}, s{S_CP}, %[{add_operand[0]}]", "salu"))
        ops.append(Op(f"s_addc_u32 s{S_CP + 1}, s{S_CP + 1}, 0", "salu"))
    ops += m0_expr_ops
    ops.append(Op("s_nop 0", "salu"))
    for v, io in zip(voffs, ioffs):
        ops.append(Op(f"global_load_lds_dwordx4 %[{v}], s[{S_CP}:{S_CP + 1}]" + (f" offset:{io}" if io else "") + (" nt" if XNT and name.startswith("cpx") else ""), "dma"))
    ops.append(Op(f".Lhs_{name}_%=:", "label", cost=0))
    for o in ops[:-1]: o.glue = True
    return ops


def epi_tile_ops(ct, ti, xr, acc_t, unp, xtag):
    """tile (ct, ti): the lane's 8 bytes of y = bf16(yo * ev + D x), written over the x they were made from (LDS)"""
    y = A_YO(ct, ti)
    ops = []
    for r in range(4):
        ops.append(Op(f"v_accvgpr_read_b32 {vr(acc_t + r)}, {ar(y + r)}", "valu", r=regs("a", y + r), w=regs("v", acc_t + r)))
    for h in range(2):
        ops.append(Op(f"v_lshlrev_b32 {vr(unp + 2 * h)}, 16, {vr(xr + h)}", "valu", r=regs("v", xr + h), w=regs("v", unp + 2 * h), lds_use=[xtag]))
        ops.append(Op(f"v_and_b32 {vr(unp + 2 * h + 1)}, 0xffff0000, {vr(xr + h)}", "valu", r=regs("v", xr + h), w=regs("v", unp + 2 * h + 1)))
    if PK:
        for h in range(2):
            ops.append(Op(f"v_pk_mul_f32 {vr(unp + 2 * h, 2)}, s[{S_D}:{S_D + 1}], {vr(unp + 2 * h, 2)}", "valu", r=regs("v", unp + 2 * h, 2), w=regs("v", unp + 2 * h, 2)))
        for h in range(2):
            ops.append(Op(f"v_pk_fma_f32 {vr(acc_t + 2 * h, 2)}, {vr(acc_t + 2 * h, 2)}, {vr(EV[ti], 2)}, {vr(unp + 2 * h, 2)} op_sel_hi:[1,0,1]", "valu",
                          r=regs("v", acc_t + 2 * h, 2) + regs("v", EV[ti]) + regs("v", unp + 2 * h, 2), w=regs("v", acc_t + 2 * h, 2), lds_use=[f"ev{ti}"]))
    else:
        for k in range(4):
            ops.append(Op(f"v_mul_f32 {vr(unp + k)}, s{S_D}, {vr(unp + k)}", "valu", r=regs("v", unp + k), w=regs("v", unp + k)))
        for k in range(4):
            ops.append(Op(f"v_fma_f32 {vr(acc_t + k)}, {vr(acc_t + k)}, {vr(EV[ti])}, {vr(unp + k)}", "valu",
                          r=regs("v", acc_t + k) + regs("v", EV[ti]) + regs("v", unp + k), w=regs("v", acc_t + k), lds_use=[f"ev{ti}"]))
    for h in range(2):
        ops.append(Op(f"v_cvt_pk_bf16_f32 {vr(unp + h)}, {vr(acc_t + 2 * h)}, {vr(acc_t + 2 * h + 1)}", "valu", r=regs("v", acc_t + 2 * h, 2), w=regs("v", unp + h)))
    ops.append(Op(f"ds_write_b64 {vr(XV)}, {vr(unp, 2)} offset:{32 * ct + ti * 16 * XROW}", "lds", r=regs("v", XV) + regs("v", unp, 2), lds_def=f"yw{ct}{ti}"))
    return ops


def epi_xread_op(ct, ti, xr):
    return Op(f"ds_read_b64 {vr(xr, 2)}, {vr(XV)} offset:{32 * ct + ti * 16 * XROW}", "lds", r=regs("v", XV), w=regs("v", xr, 2), lds_def=f"ex{ct}{ti}")


# ---------------------------------------------------------------- scheduler
def schedule(em, mfmas, tasks, budget, first_index=0, stamps=None, blocks=None):
    """emit MFMAs in order; behind each one, filler ops of the tasks that may run (after < index) by earliest
    deadline, up to `budget` cycles; tasks due before an MFMA are completed in front of it"""
    tasks = list(tasks)
    cur = None
    for k, m in enumerate(mfmas):
        idx = first_index + k
        # complete what is due
        if cur is not None and cur.ops and any(t.deadline <= idx and t.ops and t is not cur for t in tasks):
            while cur.ops: em.emit(cur.ops.pop(0))
        while True:
            due = [t for t in tasks if t.deadline <= idx and t.ops]
            if cur is not None and cur.ops and cur.deadline <= idx and cur not in due: due.append(cur)
            if not due: break
            t = min(due, key=lambda t: t.prio)
            assert t.after < idx, (t.name, t.after, idx)
            while t.ops: em.emit(t.ops.pop(0))
        if blocks and idx in blocks:
            while cur is not None and cur.ops: em.emit(cur.ops.pop(0))
            blocks[idx](em)
        if stamps and idx in stamps: stamp(em, stamps[idx])
        em.emit(m)
        left = budget
        glued = False
        while left > 0 or glued:
            if cur is None or not cur.ops:
                ready = [t for t in tasks if t.ops and t.after < idx + 1]
                if not ready: break
                cur = min(ready, key=lambda t: t.prio)
            op = cur.ops.pop(0)
            em.emit(op)
            left -= op.cost
            glued = op.glue and bool(cur.ops)
    # whatever is left
    last = first_index + len(mfmas)
    for t in sorted(tasks, key=lambda t: t.prio):
        while t.ops:
            assert t.after < last, (t.name, t.after)
            em.emit(t.ops.pop(0))


def gen_rebase(em):
    """X' *= 2^sh (sh a negative integer in an SGPR): out of and back into the accumulation registers"""
    em.raw(f"s_bitcmp1_b32 %[flags], {FLAG_REBASE}")
    em.raw("s_cbranch_scc0 .Lhs_norebase_%=")
    for b in range(0, 160, 8):
        for k in range(8): em.raw(f"v_accvgpr_read_b32 {vr(T8 + k)}, {ar(b + k)}")
        for k in range(8): em.raw(f"v_ldexp_f32 {vr(T8 + k)}, {vr(T8 + k)}, %[sh]")
        for k in range(8): em.raw(f"v_accvgpr_write_b32 {ar(b + k)}, {vr(T8 + k)}")
    em.raw("s_nop 1")
    em.raw(".Lhs_norebase_%=:", 0)


REGION_STATS = {}


def gen_step():
    em = Emitter()
    em.raw("s_nop 4", 5)
    stamp(em, 0)
    em.raw(f"s_mov_b32 s{S_M0}, m0")
    em.raw(f"s_mov_b32 s{S_D}, %[dh]")
    em.raw(f"s_mov_b32 s{S_D + 1}, %[dh]")
    gen_rebase(em)

    # ---------------- preamble
    for q in range(4): em.emit(Op(f"v_add_u32 {vr(CA[q])}, %[sbc], %[ca{q}]", "valu", w=regs("v", CA[q])))
    em.emit(Op(f"v_add_u32 {vr(XT)}, %[sxs], %[xtr]", "valu", w=regs("v", XT)))
    em.emit(Op(f"v_add_u32 {vr(XV)}, %[sxs], %[xvr]", "valu", w=regs("v", XV)))
    em.emit(Op(f"v_add_u32 {vr(XS)}, %[sxs], %[xsr]", "valu", w=regs("v", XS)))
    for m in range(4): em.emit(Op(f"v_add_u32 {vr(BA[m])}, %[sbc], %[ba{m}]", "valu", w=regs("v", BA[m])))
    for op in wq_read_ops("vw", "wq"): em.emit(op)
    for k in range(4):       # C fragments of quarter 0
        em.emit(Op(f"ds_read_b128 {ar(A_RING(k), 4)}, {vr(CA[0])} offset:{k * 4096}", "lds", r=regs("v", CA[0]), w=regs("a", A_RING(k), 4), lds_def=f"c{k}"))
    for op in xw_read_ops(0, 0, "x"): em.emit(op)
    for f in range(6):       # causal C.B^T of this chunk (the pointer is 2 KiB into the chunk's 6 KiB: 13-bit signed offsets)
        em.emit(Op(f"global_load_dwordx4 {vr(CBV[f], 4)}, %[cbo], %[pcb] offset:{f * 1024 - 2048}", "vmem", w=regs("v", CBV[f], 4), vm_def=f"cb{f}"))
    em.emit(Op("global_load_ushort %[dtout], %[dto], %[pdt]", "vmem", vm_def="dt"))
    for ct in range(PT):
        for op in snap_ops(0, ct, SB[0][ct], T8): em.emit(op)

    REGION_STATS["preamble"] = dict(em.stats)
    stamp(em, 1)

    def b_read_ops(m):
        i, ks = m // 2, m % 2
        base = BA[i // 2]
        off = (i & 1) * 8 + ks * 8192
        d = A_RING(m)
        return [Op(f"ds_read_b64_tr_b16 {ar(d, 2)}, {vr(base)} offset:{off}", "lds", r=regs("v", base), w=regs("a", d, 2), lds_def=f"b{m}a"),
                Op(f"ds_read_b64_tr_b16 {ar(d + 2, 2)}, {vr(base)} offset:{off + 1024}", "lds", r=regs("v", base), w=regs("a", d + 2, 2), lds_def=f"b{m}b")]

    # ---------------- phase A: Yoff (80 MFMAs), Ydiag (30), in three variants
    #   F  floating step: x~ = wts x serves Ydiag and the state update
    #   R  reset step: Ydiag takes x~ in the old frame (wtd), the state update x~ in the new one (wts): both from one
    #      unpacking of x, the first into the registers of the state's bf16 copy once those are free
    #   S  standard step: Ydiag takes x itself (read into the same registers) and the per-head mask built into the C.B^T
    #      registers beside the Yoff MFMAs; the accumulators get their row factor between Yoff and Ydiag
    def phase_a(em, var):
        YA = (lambda ct, ks: XW[ct][ks]) if var == "F" else (lambda ct, ks: SB[ks][ct])        # Ydiag's A operand
        mf = []
        for q in range(4):
            for ti in range(4):
                for ct in range(PT):
                    mf.append(mfma(A_YO(ct, ti), SB[q & 1][ct], A_RING(4 * q + ti), None if q == 0 else A_YO(ct, ti), b_is_acc=True))
                    mf[-1].lds_use = [f"c{4 * q + ti}"]
        ydiag = [(0, 0, 0), (1, 0, 1), (2, 0, 2), (3, 0, 4), (2, 1, 3), (3, 1, 5)]        # (ti, ks, fragment)
        for ti, ks, f in ydiag:
            for ct in range(PT):
                m = mfma(A_YO(ct, ti), YA(ct, ks), CBV[f], A_YO(ct, ti))
                m.vm_use = [f"cb{f}"]
                if var == "S": m.lds_use = [f"xr{ct}{ks}a", f"xr{ct}{ks}b"]
                mf.append(m)
        tasks = []
        for k in range(4, 16):       # C fragments of quarters 1..3 through the ring
            q, ti = k // 4, k % 4
            tasks.append(Task(f"c{k}", [Op(f"ds_read_b128 {ar(A_RING(k), 4)}, {vr(CA[q])} offset:{ti * 4096}", "lds", r=regs("v", CA[q]),
                                           w=regs("a", A_RING(k), 4), lds_def=f"c{k}")], after=(k - 4) * 5 + 4, deadline=k * 5 - 6, prio=k * 5 - 40))
        for q in range(1, 4):
            for ct in range(PT):
                after = -1 if q == 1 else (q - 2) * 20 + 15 + ct
                tasks.append(Task(f"snap{q}{ct}", snap_ops(q, ct, SB[q & 1][ct], T8), after=after, deadline=q * 20 + ct, prio=q * 20 + ct - 2))
        order = [(ct, 0) for ct in range(PT)] + [(ct, 1) for ct in range(PT)]
        if var in ("F", "S"):
            for n, (ct, ks) in enumerate(order):
                ops = []
                if n + 1 < len(order): ops += xw_read_ops(order[n + 1][0], order[n + 1][1], "x")
                ops += xw_comp_ops(ct, ks, "x", "wq")
                sf = var == "F" or os.environ.get("GEN_SXW")
                dl = 80 + (0 if ks == 0 else 20) + ct if sf else 109
                tasks.append(Task(f"xw{ct}{ks}", ops, after=-1, deadline=dl, prio=8 * n + 3 if sf else 60 + 4 * n))
        if var == "R":
            # both x~ from one unpacking: old frame (weights WQ2 <- vw2) into SB[ks][ct], new frame (WQ) in place
            for n, (ct, ks) in enumerate(order):
                ops = []
                if n == 0: ops += wq_read_ops("vw2", "w2", WQ2)
                if n + 1 < len(order): ops += xw_read_ops(order[n + 1][0], order[n + 1][1], "x")
                ops += xw_dual_ops(ct, ks, "x", "wq", "w2", SB[ks][ct])
                tasks.append(Task(f"xw{ct}{ks}", ops, after=(55 if ks == 0 else 75) + ct, deadline=80 + (0 if ks == 0 else 20) + ct, prio=56 + 5 * n))
        if var == "S":
            # x itself for Ydiag, into the registers of the state's bf16 copy
            for n, (ct, ks) in enumerate(order):
                d = SB[ks][ct]
                off = 32 * ct + ks * 32 * XROW
                ops = [Op(f"ds_read_b64_tr_b16 {vr(d, 2)}, {vr(XT)} offset:{off}", "lds", r=regs("v", XT), w=regs("v", d, 2), lds_def=f"xr{ct}{ks}a"),
                       Op(f"ds_read_b64_tr_b16 {vr(d + 2, 2)}, {vr(XT)} offset:{off + 4 * XROW}", "lds", r=regs("v", XT), w=regs("v", d + 2, 2), lds_def=f"xr{ct}{ks}b")]
                tasks.append(Task(f"xr{ct}{ks}", ops, after=(55 if ks == 0 else 75) + ct, deadline=76 + (0 if ks == 0 else 20) + ct, prio=56 + 5 * n))
            tasks.append(Task("mask", std_mask_ops(), after=int(os.environ.get("GEN_MASK_AFTER", 20)), deadline=80, prio=30))
            for ti in range(4):
                tasks.append(Task(f"scale{ti}", std_scale_ops(ti), after=max(64 + 5 * ti + 4, int(os.environ.get("GEN_SCALE_AFTER", 0))), deadline=80 + 5 * ti, prio=70 + 5 * ti))
        # copies of chunk c + 1 (B, C, three groups of x)
        m0b = [Op("s_mov_b32 m0, %[lb]", "salu")]
        m0c = [Op("s_mov_b32 m0, %[lc]", "salu")]
        tasks.append(Task("cpB", copy_group_ops("cpb" + var, "pb", ("z", 0), m0b, ["ob0", "ob1", "ob2", "ob3"], [0, 1024, 2048, 3072]), after=4, deadline=100, prio=10))
        tasks.append(Task("cpC", copy_group_ops("cpc" + var, "pc", ("z", 0), m0c, ["oc0", "oc1", "oc2", "oc3"], [0, 1024, 2048, 3072]), after=18, deadline=100, prio=24))
        for g in range(3):
            n = min(4, NXI - 4 * g)
            vo = ["ox0", "ox1", "ox2", "ox3"][:n]
            if 4 * g + n == NXI: vo[-1] = "oxl"
            m0x = [Op(f"s_add_u32 m0, %[lx], {RPI * 4 * g * XROW}" if g else "s_mov_b32 m0, %[lx]", "salu")]
            tasks.append(Task(f"cpX{g}", copy_group_ops(f"cpx{g}" + var, "px", ("xg4", g), m0x, vo, [j * RPI * XROW for j in range(n)]),
                              after=32 + 14 * g, deadline=100, prio=38 + 14 * g))
        # phase B operands that may already be fetched: ev, the first B fragments (ring slots free after the last C use)
        pre_b = [Op(f"ds_read_b32 {vr(EV[ti])}, %[vev] offset:{64 * ti}", "lds", w=regs("v", EV[ti]), lds_def=f"ev{ti}") for ti in range(4)]
        tasks.append(Task("preB", pre_b, after=81 if var == "S" else 70, deadline=109, prio=108 if var == "S" else 95))
        for m in range(4):
            tasks.append(Task(f"b{m}", b_read_ops(m), after=(12 + m) * 5 + 4, deadline=110 + m * 5 - 4, prio=100 + m))
        va = sum(op.cost for t in tasks for op in t.ops)
        schedule(em, mf, tasks, budget=va / len(mf) + 1.0, stamps={20: 2, 40: 3, 60: 4, 80: 5})
        em.drain_lds()

    import copy
    def fork(em):
        e2 = Emitter()
        e2.__dict__.update(copy.deepcopy({k: v for k, v in em.__dict__.items() if k != "lines"}))
        e2.lines = []
        return e2
    em.raw(f"s_bitcmp1_b32 %[flags], {FLAG_RESET}")
    em.raw("s_cbranch_scc1 .Lhs_pa_rs_%=")
    emR, emS = fork(em), fork(em)
    em.uid, emR.uid, emS.uid = "f", "r", "s"
    base = dict(em.stats)
    phase_a(em, "F")
    REGION_STATS["phase A, floating step"] = {k: v - base.get(k, 0) for k, v in em.stats.items()}
    em.raw("s_branch .Lhs_pa_join_%=")
    emR.raw(".Lhs_pa_rs_%=:", 0)
    emR.raw(f"s_bitcmp1_b32 %[flags], {FLAG_STD}")
    emR.raw("s_cbranch_scc1 .Lhs_pa_s_%=")
    emS.pos, emS.lines = emR.pos, []
    phase_a(emR, "R")
    REGION_STATS["phase A, reset step"] = {k: v - base.get(k, 0) for k, v in emR.stats.items()}
    emR.raw("s_branch .Lhs_pa_join_%=")
    emS.raw(".Lhs_pa_s_%=:", 0)
    phase_a(emS, "S")
    REGION_STATS["phase A, standard step (both paths of the four row-factor groups counted: 60 + 20 each)"] = {k: v - base.get(k, 0) for k, v in emS.stats.items()}
    assert em.vm_seq == emR.vm_seq == emS.vm_seq and em.lds_seq == em.lds_done, (em.vm_seq, emR.vm_seq, emS.vm_seq)
    em.lines += emR.lines + emS.lines
    em.raw(".Lhs_pa_join_%=:", 0)
    # after the join: every accumulator tile may have just been written by an MFMA, every vector register by the vector ALU
    em.pos = max(em.pos, emR.pos, emS.pos) + 1
    for ct in range(PT):
        for ti in range(4):
            for r in regs("a", A_YO(ct, ti), 4): em.wr[r] = (em.pos, "mfma")
        for ks in range(2):
            for r in regs("v", XW[ct][ks], 4): em.wr[r] = (em.pos, "valu")
    em.vm_done = min(em.vm_done, emR.vm_done, emS.vm_done)
    em.lds_idx.update(emR.lds_idx)      # (tags defined in every variant: b0 .. b3, ev0 .. ev3; everything has landed)
    em.lds_done = em.lds_seq = max(em.lds_seq, emR.lds_seq, emS.lds_seq)
    for t in list(em.lds_idx): em.lds_idx[t] = min(em.lds_idx[t], em.lds_seq - 1)
    stamp(em, 6)

    # ---------------- phase B: state update (80 MFMAs) beside the y epilogue; two copies (reset: first k-step onto zero)
    def phase_b(em, reset):
        mf = []
        for i in range(8):
            for ks in range(2):
                for ct in range(PT):
                    m = mfma(A_STATE(ct, i), A_RING(2 * i + ks), XW[ct][ks], None if (reset and ks == 0) else A_STATE(ct, i), a_is_acc=True)
                    m.lds_use = [f"b{2 * i + ks}a", f"b{2 * i + ks}b"]
                    mf.append(m)
        tasks = []
        for m in range(4, 16):
            tasks.append(Task(f"b{m}", b_read_ops(m), after=110 + (m - 4) * 5 + 4, deadline=110 + m * 5 - 6, prio=110 + m * 5 - 40))
        # y epilogue: tiles t-tile by t-tile (x read two tiles ahead), y back into the x tile; the chunk then leaves as whole
        # 160-byte rows: store k = rows 6 k .. 6 k + 5 (60 lanes x 16 bytes, lane = piece; the last one rows 60 .. 63), rows
        # past the end of the sequence switched off (row < rem)
        seq = [(ti, ct) for ti in range(4) for ct in range(PT)]
        xrs = [XRP[0], XRP[0] + 2, XRP[1], XRP[1] + 2]
        pri = 200
        stores_after_tile = {4: [0, 1], 9: [2, 3, 4], 14: [5, 6, 7], 19: [8, 9, 10]}      # rows 6 k + 5 <= 16 ti + 15
        for n, (ti, ct) in enumerate(seq):
            ops = []
            if n == 0:
                ops.append(epi_xread_op(ct, ti, xrs[0]))
                ops.append(epi_xread_op(seq[1][1], seq[1][0], xrs[1]))
            if n + 2 < len(seq): ops.append(epi_xread_op(seq[n + 2][1], seq[n + 2][0], xrs[(n + 2) % 4]))
            tb = (n & 1) * 4
            ops += epi_tile_ops(ct, ti, xrs[n % 4], T8 + tb, U8 + tb, f"ex{ct}{ti}")
            tasks.append(Task(f"epi{ti}{ct}", ops, after=111 if ti < 2 else 114, prio=pri))
            pri += 1
            grp = stores_after_tile.get(n, [])
            def sd_read(k):
                return Op(f"ds_read_b128 {vr(SD[k & 1], 4)}, {vr(XS)} offset:{k * RPI * XROW}", "lds", r=regs("v", XS), w=regs("v", SD[k & 1], 4), lds_def=f"sd{k}")
            for j, k in enumerate(grp):
                sops = []
                if j == 0:
                    sops.append(sd_read(k))
                    if len(grp) > 1: sops.append(sd_read(grp[1]))
                if k == 0:
                    sops.append(Op(f"s_mov_b64 s[{S_Y}:{S_Y + 1}], %[py]", "salu"))
                else:
                    sops.append(Op(f"s_add_u32 s{S_Y}, s{S_Y}, %[y6]", "salu"))
                    sops.append(Op(f"s_addc_u32 s{S_Y + 1}, s{S_Y + 1}, 0", "salu"))
                sops.append(Op(f"s_sub_i32 s{S_T}, %[rem], {6 * k}", "salu"))
                sops.append(Op(f"s_min_i32 s{S_T}, s{S_T}, 6", "salu"))
                sops.append(Op(f"v_cmpx_gt_i32 vcc, s{S_T}, %[lrow]", "valu", glue=True, lds_use=[f"sd{k}"]))      # EXEC = rows of this store inside the sequence
                if not NOSTORE: sops.append(Op(f"global_store_dwordx4 %[yst], {vr(SD[k & 1], 4)}, s[{S_Y}:{S_Y + 1}]", "store", r=regs("v", SD[k & 1], 4), glue=True))
                sops.append(Op("s_mov_b64 exec, -1", "salu"))
                if j + 2 < len(grp): sops.append(sd_read(grp[j + 2]))
                tasks.append(Task(f"st{k}", sops, after=114, prio=pri))
                pri += 1
        tasks.append(Task("prep", prep_ops("r" if reset else "n"), after=111, prio=150))
        va = sum(op.cost for t in tasks for op in t.ops)
        schedule(em, mf, tasks, budget=va / len(mf) + 1.0, first_index=110, stamps={150: 8})
        stamp(em, 9)

    import copy
    em.raw(f"s_bitcmp1_b32 %[flags], {FLAG_RESET}")
    em.raw("s_cbranch_scc1 .Lhs_pb_reset_%=")
    base_state = copy.deepcopy({k: v for k, v in em.__dict__.items() if k != "lines"})
    base_b = dict(em.stats)
    phase_b(em, False)
    REGION_STATS["phase B (state update + y epilogue + the next chunk's vectors; a standard chunk's 42 extra counted)"] = {k: v - base_b.get(k, 0) for k, v in em.stats.items()}
    em.raw("s_branch .Lhs_pb_join_%=")
    end_state = {k: v for k, v in em.__dict__.items() if k != "lines"}
    em2 = Emitter()
    em2.__dict__.update(copy.deepcopy(base_state))
    em2.lines = []
    em2.raw(".Lhs_pb_reset_%=:", 0)
    phase_b(em2, True)
    em.lines += em2.lines
    em.raw(".Lhs_pb_join_%=:", 0)
    assert em2.vm_seq == em.vm_seq and em2.lds_seq == em.lds_seq

    # ---------------- end: the copies of chunk c + 1 have landed (the y stores may stay in flight); without copies: everything
    em.raw(f"s_bitcmp1_b32 %[flags], {FLAG_COPY}")
    em.raw("s_cbranch_scc1 .Lhs_endw_%=")
    em.raw("s_waitcnt vmcnt(0)")
    em.raw(".Lhs_endw_%=:", 0)
    em.raw(f"s_waitcnt vmcnt({0 if NOSTORE else 11}) lgkmcnt(0)")
    stamp(em, 10)
    em.raw(f"s_mov_b32 m0, s{S_M0}")
    return em


def clobbers():
    c = ["memory", "vcc", "scc"]
    c += [f"s{i}" for i in range(72, 100)]
    c += [f"v{i}" for i in range(V0, 256)]
    c += [f"a{i}" for i in range(256)]
    return c


def summary():
    """static instruction counts per region of the generated body (`--summary`; profiles/r05_ssd_head_isa.md)"""
    rows = []
    for name, st in REGION_STATS.items():
        valu = st.get("valu", 0) + st.get("perm", 0)
        rows.append(f"| {name} | {st.get('mfma', 0)} | {valu} | {st.get('lds', 0) + st.get('lds_nc', 0)} | {st.get('vmem', 0) + st.get('dma', 0) + st.get('store', 0)} | "
                    f"{st.get('salu', 0)} | {st.get('nop', 0)} |")
    return "\n".join(["| region | MFMA | vector ALU | LDS | vector memory | scalar | inserted wait states |", "|---|---:|---:|---:|---:|---:|---:|"] + rows)


def main():
    global STAMPS
    STAMPS = "--stamps" in sys.argv
    global PK
    PK = "--scalar" not in sys.argv
    global NOSTORE
    NOSTORE = "--nostore" in sys.argv
    em = gen_step()
    if "--summary" in sys.argv:
        print(summary())
        return
    lines = em.lines
    print("// generated by timeviper_amd/devtools/gen_head_step.py — do not edit")
    print(f"// {sum(1 for l in lines if l.startswith('v_mfma'))} MFMA lines (phase B twice), {len(lines)} lines; emitter stats {em.stats}")
    print("#define TV_HEAD_STEP_ASM \\")
    for ln in lines:
        print(f'  "{ln}\\n\\t" \\')
    print('  ""')
    print("#define TV_HEAD_STEP_CLOBBERS " + ", ".join(f'"{c}"' for c in clobbers()))
    print(f"#define TV_HEAD_STEP_V0 {V0}")
    # small statements of the kernel's prologue / epilogue (the state lives in a0 .. a159 between the steps)
    print("#define TV_HEAD_STATE_ZERO \\")
    for i in range(160): print(f'  "v_accvgpr_write_b32 a{i}, 0\\n\\t" \\')
    print('  ""')
    print("#define TV_HEAD_STATE_CLOBBERS " + ", ".join(f'"a{i}"' for i in range(160)))


if __name__ == "__main__":
    main()
