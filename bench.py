"""bench.py — TimeViper-9B long-video forward (prefill) on N MI355X, synthetic data.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames T]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of `GenericTimeViperVLM.forward` (ViT over 256-frame clips, eight per launch ->
ToMe+MLP projector -> fusion -> 56-layer Nemotron-Nano-9B-v2 hybrid stack with TransV
pdrop -> last-token logits) over T frames that are already resident in HBM.  With N > 1
the frame/token sequence is sharded over the ranks (timeviper_amd.distributed; frame ranges sized so
that causal attention does not leave the last rank behind) and the total work is fixed
("scaling": "strong").  Rank 0 prints ONE JSON line.

roofline: the SSD selective-scan kernel — algorithmic bytes (45 312 B per
token per Mamba layer at Nano dims, bf16) / its launch durations measured live with
events on the launch stream inside the timed steps.  rooflines: the same for the ViT
attention, the causal LLM attention (useful FLOPs against the dense bf16 MFMA peak) and the
patch-embedding GEMM (MFMA and HBM).
cpu_baseline: the CPU oracle (eager PyTorch fp32 restatement of the reference path),
timed on this box's host cores on a bounded sample and scaled to frames/s.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import math

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PDROP = "uni_14_0.8-attn_21_0.6-attn_30_0.4-attn_39_0.2"   # evaluate.py:170 defaults
TOK_PER_FRAME = 16                                         # arch_specifier tome_mlp-16
HBM_PEAK_GBS = 8000.0                                      # MI355X_MICROARCH.md


MFMA_BF16_PEAK_TFLOPS = 2500.0                             # dense bf16, MI355X_MICROARCH.md
SCAN_SOURCES = ("ssd_head.hip", "ssd_head_step.inc", "ssd_slice.hip", "ssd_correct.hip", "ssd_scan.hip", "ssd_common.hpp", "conv1d.hip")


def copy_rate(dev, rl) -> dict:
    """GB/s (bytes read + bytes written) of `y.copy_(x)` on 3.4 GB — the size of the scan's x at 163 940 tokens — timed with events
    on the current stream after the bench's timed region, and where the scan's HBM traffic rate stands against it."""
    import torch
    n = 163940 * 20480 // 4
    x = torch.empty(n, dtype=torch.int32, device=dev).random_()
    y = torch.empty_like(x)
    best = None
    for _ in range(3):
        y.copy_(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y.copy_(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        best = ms if best is None else min(best, ms)
    gbs = 2 * n * 4 / (best * 1e-3) / 1e9
    out = {"copy_rate": round(gbs, 1), "copy_rate_unit": "GB/s (read + write) of torch copy_ on 3.36 GB, this box"}
    if rl.get("traffic"):
        out["traffic_over_copy_rate"] = round(rl["traffic"] / gbs, 4)
    return out


def scan_bytes_per_token(cfg) -> int:
    H, P, G, N = cfg.mamba_num_heads, cfg.mamba_head_dim, cfg.n_groups, cfg.ssm_state_size
    return 2 * H * P + 2 * H + 2 * 2 * G * N + 2 * H * P   # x, dt, B+C read; y written (bf16)


def scan_source_id() -> str:
    """Hash of the scan kernels' sources: the PMC traffic file is only valid for the kernels it was
    taken on."""
    import hashlib
    h = hashlib.sha256()
    for name in SCAN_SOURCES:
        h.update((ROOT / "timeviper_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


class OpTimers:
    """Event-timed wrappers (events on the launch stream, inside the timed steps) around the operators
    whose roofline the bench line carries: the SSD scan (HBM), the ViT and the causal LLM attention
    (MFMA), the patch-embedding GEMM (MFMA and HBM)."""

    def __init__(self, K, event=None, model=None):
        self.K, self.on, self.rec, self.saved = K, False, {}, {}
        self.event = event or (lambda: torch.cuda.Event(enable_timing=True))
        # library GEMMs (hipBLASLt through F.linear / torch.addmm) by ROLE: the last component of the owning nn.Linear's
        # name (in_proj, out_proj, up_proj, down_proj, q_proj ... of the language model; qkv, proj, fc1, fc2 of the ViT;
        # the projector's and the merge module's linears), looked up by the weight's storage, then by its shape (the ViT's
        # zero-padded inference copies)
        self.role_by_ptr, self.role_by_shape, self.lib_saved = {}, {}, {}
        if model is not None:
            for name, mod in model.named_modules():
                if isinstance(mod, torch.nn.Linear):
                    parts = name.split(".")
                    role = parts[-1]
                    if "vision" in name or "vit" in name.lower() or "blocks" in parts:
                        role = "vit." + role
                    elif "projector" in name:
                        role = "projector." + role
                    self.role_by_ptr[mod.weight.data_ptr()] = role
                    self.role_by_shape.setdefault(tuple(mod.weight.shape), role)

    def _useful(self, w):
        """(rows, columns) of a weight (N, K) without the zero padding of the ViT's inference copies"""
        return self.K.PADDED_USEFUL.get(w.data_ptr(), (w.shape[0], w.shape[1]))

    def _lib_role(self, w):
        r = self.role_by_ptr.get(w.data_ptr())
        if r is None:
            n, k = self._useful(w)
            r = self.role_by_shape.get((n, k)) or f"{n}x{k}"
        return r

    def _record(self, key, val, fn):
        e0, e1 = self.event(), self.event()
        e0.record()
        out = fn()
        e1.record()
        self.rec.setdefault(key, []).append((e0, e1, val))
        return out

    def _wrap(self, name, classify):
        orig = getattr(self.K, name)
        self.saved[name] = orig

        def timed(*a, **kw):
            key = classify(*a, **kw) if self.on else None
            if key is None:
                return orig(*a, **kw)
            e0, e1 = self.event(), self.event()
            e0.record()
            out = orig(*a, **kw)
            e1.record()
            self.rec.setdefault(key[0], []).append((e0, e1, key[1]))
            return out
        setattr(self.K, name, timed)

    def __enter__(self):
        def scan(x, *a, **kw):
            return ("scan", x.shape[0] * x.shape[1])                       # tokens

        def attn(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, return_lse=False):
            B, Lq, Hq, D = q.shape
            Lk = k.shape[1]
            if causal and Lq > 1:
                pairs = Lq * Lk - Lq * (Lq - 1) // 2                       # bottom-right aligned mask
                return ("attn_causal", 4.0 * B * Hq * D * pairs)
            if not causal and Lq == Lk and Lq <= 1025:                      # ViT frames / tubes
                return ("attn_vit", 4.0 * B * Hq * D * Lq * Lk)
            return None                                                      # TransV cross-attention, decode

        def patch(pixels, weight, *a, **kw):
            F_, Cin, Hh, Ww = pixels.shape
            pp = weight.shape[-1]
            n = (Hh // pp) * (Ww // pp)
            flops = 2.0 * F_ * n * Cin * pp * pp * weight.shape[0]
            byts = F_ * (Cin * Hh * Ww + n * weight.shape[0]) * pixels.element_size()
            return ("patch_embed", (flops, byts))
        # HBM-bound companions of the scan (bytes per token and Mamba layer) and the ViT's element-wise passes:
        # key[1] = algorithmic bytes of the call
        def conv(xBC, weight, bias, d_inner, ngroups, dstate, *a, **kw):
            return ("conv", xBC.shape[0] * xBC.shape[1] * 2 * xBC.shape[2] * xBC.element_size())   # xBC read, x|B|C written

        def gnorm(x, weight, bias=None, z=None, *a, **kw):
            rows = x.numel() // x.shape[-1]
            return ("gated_norm", rows * x.shape[-1] * x.element_size() * (3 if z is not None else 2))

        def rms(x, weight, eps, residual=None, return_sum=False):
            n = x.numel() * x.element_size()
            return ("rmsnorm", n * (2 + (residual is not None) + bool(return_sum and residual is not None)))

        def ln(x, weight, bias, eps, residual=None, return_sum=False, row_bias=None):
            n = x.numel() * x.element_size()
            return ("layernorm", n * (2 + (residual is not None) + bool(return_sum and residual is not None)))

        def gelu(x, inplace=False):
            return ("gelu", 2 * x.numel() * x.element_size())
        def gemm(x, weight, bias=None, epilogue=0, out=None):
            rows = x.numel() // x.shape[-1]
            # USEFUL flops: zero-padded weight rows (so400m's fc1: 4 352 rows for 4 304 outputs) are not counted, nor is
            # the activation.  (The padded rows are all-zero by construction: ZeroPaddedLinears.)
            n_useful = getattr(weight, "_tv_useful_rows", weight.shape[0])
            return ("gemm_fused", 2.0 * rows * n_useful * weight.shape[1])
        self._wrap("linear_fused", gemm)
        self._wrap("mamba_chunk_scan_combined", scan)
        self._wrap("flash_attn_func", attn)
        self._wrap("patch_embed", patch)
        self._wrap("causal_conv1d_xbc", conv)
        self._wrap("rmsnorm_fn", gnorm)
        self._wrap("rms_norm", rms)
        self._wrap("layer_norm", ln)
        self._wrap("gelu", gelu)
        # the library GEMMs: F.linear (nn.Linear.forward goes through it) and torch.addmm (x += h W^T of the ViT stream)
        import torch.nn.functional as F
        lin, addmm = F.linear, torch.addmm
        self.lib_saved = {"linear": lin, "addmm": addmm}

        def timed_linear(x, w, b=None):
            if not self.on or x.dim() < 2 or not x.is_cuda:
                return lin(x, w, b)
            rows = x.numel() // x.shape[-1]
            n, k = self._useful(w)
            return self._record("lib:" + self._lib_role(w), 2.0 * rows * n * k, lambda: lin(x, w, b))

        def timed_addmm(inp, m1, m2, *a, **kw):
            if not self.on or not m1.is_cuda:
                return addmm(inp, m1, m2, *a, **kw)
            wt = m2.t()                                    # (N, K) as stored (same storage)
            n, k = self._useful(wt)
            return self._record("lib:" + self._lib_role(wt) + " (+=)", 2.0 * m1.shape[0] * n * k,
                                lambda: addmm(inp, m1, m2, *a, **kw))
        F.linear, torch.addmm = timed_linear, timed_addmm
        return self

    def __exit__(self, *exc):
        for name, orig in self.saved.items():
            setattr(self.K, name, orig)
        if self.lib_saved:
            import torch.nn.functional as F
            F.linear, torch.addmm = self.lib_saved["linear"], self.lib_saved["addmm"]

    def _ms(self, key):
        r = self.rec.get(key, [])
        return sum(e0.elapsed_time(e1) for e0, e1, _ in r), r

    def scan_roofline(self, bytes_per_token):
        ms, r = self._ms("scan")
        if not r:
            return None
        tokens = sum(n for _, _, n in r)
        gbs = tokens * bytes_per_token / (ms * 1e-3) / 1e9
        # HBM bytes actually moved per algorithmic byte: rocprofv3 PMC passes (FETCH_SIZE x2, WRITE_SIZE,
        # MI355X_MICROARCH.md) on this op, committed under profiles/ (PMC cannot be collected from
        # here).  The file names the kernel sources it was taken on: a ratio from other kernels is
        # refused, not silently reused.
        traffic, src = None, "no profiles/r*_ssd_scan_traffic.json"
        files = sorted((ROOT / "profiles").glob("r*_ssd_scan_traffic.json"))
        if files:
            tf = json.loads(files[-1].read_text())
            if tf.get("scan_source_id") == scan_source_id():
                ratio = tf["hbm_over_algorithmic"]
                traffic = round(gbs * ratio, 1)
                src = f"profiles/{files[-1].name} ({tf.get('kernel')}): HBM bytes = {ratio:.3f} x algorithmic"
            else:
                src = (f"profiles/{files[-1].name} was taken on other scan kernels (source id "
                       f"{tf.get('scan_source_id')}, tree {scan_source_id()}): re-run devtools/pmc_scan.sh")
        return {"bound": "hbm", "kernel": "ssd_scan (tv_ssd_scan_cb_fwd: ssd_head_asm_kernel — head-per-wave march, the 64-token step one "
                                          "generated, hand-scheduled instruction stream (csrc/ssd_head_step.inc) — x 8-16 sequence "
                                          "segments (floating, reset and standard steps) + ssd_dt_transpose + ssd_chain_prefix + "
                                          "ssd_correct_list kernels)",
                "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": src, "launches": len(r),
                "avg_launch_us": round(ms * 1e3 / len(r), 1), "bytes_per_token": bytes_per_token}

    def mfma_roofline(self, key, kernel):
        ms, r = self._ms(key)
        if not r:
            return None
        flops = sum(f for _, _, f in r)
        tf = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": kernel, "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                "launches": len(r), "avg_launch_us": round(ms * 1e3 / len(r), 1)}

    def patch_rooflines(self):
        ms, r = self._ms("patch_embed")
        if not r:
            return []
        flops = sum(f[0] for _, _, f in r)
        byts = sum(f[1] for _, _, f in r)
        tf, gbs = flops / (ms * 1e-3) / 1e12, byts / (ms * 1e-3) / 1e9
        base = {"kernel": "patch_embed_kernel (tv_patch_embed_fwd)", "launches": len(r),
                "avg_launch_us": round(ms * 1e3 / len(r), 1), "traffic": None}
        return [dict(base, bound="mfma", achieved=round(tf, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                     frac=round(tf / MFMA_BF16_PEAK_TFLOPS, 4)),
                dict(base, bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=round(gbs / HBM_PEAK_GBS, 4))]

    def hbm_roofline(self, key, kernel):
        ms, r = self._ms(key)
        if not r:
            return None
        gbs = sum(b for _, _, b in r) / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "launches": len(r),
                "avg_launch_us": round(ms * 1e3 / len(r), 1)}

    def mixer_trio_roofline(self, cfg):
        """conv + scan + gated norm of the Mamba layers as ONE operator against SURVEY 8d's fused lower bound: read
        xBC + dt + gate, write the normed y (65 792 B per token and layer at Nano dims).  Work that moves between the
        three kernels (C.B^T into the conv, the carried-in correction into the norm, ...) stays inside this entry."""
        ms = sum(self._ms(k)[0] for k in ("conv", "scan", "gated_norm"))
        _, r = self._ms("scan")
        if not r or not ms:
            return None
        H, P, G, N = cfg.mamba_num_heads, cfg.mamba_head_dim, cfg.n_groups, cfg.ssm_state_size
        per_token = 2 * ((H * P + 2 * G * N) + H + H * P) + 2 * H * P
        gbs = sum(n for _, _, n in r) * per_token / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "Mamba mixer trio: causal_conv1d_xbc(+C.B^T) + mamba_chunk_scan_combined + rmsnorm_fn, "
                                          "against the fused bound (read xBC, dt, gate; write y)",
                "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "traffic": None, "launches": len(r), "avg_launch_us": round(ms * 1e3 / len(r), 1),
                "bytes_per_token": per_token}

    def all_rooflines(self, bytes_per_token, cfg=None):
        out = [self.scan_roofline(bytes_per_token),
               self.mfma_roofline("attn_vit", "flash_fwd_vit_kernel (generated tile loop; other dtypes: flash_fwd_stream_kernel), ViT frames (non-causal, head_dim 72; useful FLOPs)"),
               self.mfma_roofline("attn_causal", "flash_fwd_kernel, causal GQA (LLM attention layers; useful FLOPs)")]
        out.append(self.mfma_roofline("gemm_fused", "gemm_persist_kernel<bias + erf-GELU> (tv_gemm_bf16_fwd: ViT fc1 with the activation in "
                                                    "the epilogue, persistent work-groups; useful GEMM FLOPs only: 4 304 of the "
                                                    "4 352 zero-padded columns)"))
        out += self.patch_rooflines()
        out += [self.hbm_roofline("conv", "conv1d_xbc_kernel + conv1d_bc_cb_kernel (tv_causal_conv1d_xbc_cb_fwd: conv + SiLU + "
                                          "x|B|C split + causal C.B^T fragments; bytes: xBC read, x|B|C written)"),
                self.hbm_roofline("gated_norm", "rmsnorm_gated_kernel (tv_rmsnorm_gated_fwd; bytes: y, gate read, out written)"),
                self.mixer_trio_roofline(cfg) if cfg is not None else None,
                self.hbm_roofline("rmsnorm", "rmsnorm_kernel (tv_rmsnorm_fwd, residual add fused; bytes: every row read / written once)"),
                self.hbm_roofline("layernorm", "layernorm_rows_kernel (tv_layernorm_fwd, ViT, 8 rows per wave; bytes: every row read / written once)"),
                self.hbm_roofline("gelu", "gelu_kernel (tv_gelu_fwd, in place: ViT MLP and projector; bytes: read + write)")]
        # hipBLASLt GEMMs by role (useful FLOPs: zero-padded rows / columns of the ViT's inference copies are not counted)
        for key in sorted(k for k in self.rec if k.startswith("lib:")):
            out.append(self.mfma_roofline(key, f"hipBLASLt GEMM, role {key[4:]} (F.linear / torch.addmm; useful FLOPs)"))
        return [o for o in out if o]

    def accounted_ms(self):
        """Event time of everything this class times, in ms (all steps)"""
        return sum(self._ms(k)[0] for k in self.rec)


def cpu_baseline(cfg, frames_sample=256, vit_frames=16):
    """Oracle (port of the reference's eager path) on the host cores, bounded sample:
    one Mamba / attention / MLP layer at Nano-9B dims over `frames_sample` frames of tokens,
    one SigLIP block on `vit_frames` frames; scaled by the layer counts to a whole forward."""
    from oracle import model as om
    from oracle import ops as R
    # eager PyTorch stops scaling (and then degrades) well below the 256 hardware threads of the GPU
    # box on these layer sizes: use at most 32 threads and say so in `cores`
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    ocfg = om.OracleConfig.from_hf(cfg)
    L = frames_sample * TOK_PER_FRAME + 100
    g = torch.Generator().manual_seed(0)
    D, H, P, N, G = cfg.hidden_size, cfg.mamba_num_heads, cfg.mamba_head_dim, cfg.ssm_state_size, cfg.n_groups
    d_in, conv = H * P, H * P + 2 * G * N
    rn = lambda *s: torch.randn(*s, generator=g) * 0.02
    sd = {"m.in_proj.weight": rn(d_in + conv + H, D), "m.conv1d.weight": rn(conv, 1, 4) * 10,
          "m.conv1d.bias": rn(conv), "m.A_log": torch.log(torch.rand(H, generator=g) * 15 + 1),
          "m.D": torch.ones(H), "m.dt_bias": torch.full((H,), -3.0), "m.norm.weight": torch.ones(d_in),
          "m.out_proj.weight": rn(D, d_in),
          "a.q_proj.weight": rn(cfg.num_attention_heads * cfg.head_dim, D),
          "a.k_proj.weight": rn(cfg.num_key_value_heads * cfg.head_dim, D),
          "a.v_proj.weight": rn(cfg.num_key_value_heads * cfg.head_dim, D),
          "a.o_proj.weight": rn(D, cfg.num_attention_heads * cfg.head_dim),
          "f.up_proj.weight": rn(cfg.intermediate_size, D), "f.down_proj.weight": rn(D, cfg.intermediate_size)}
    h = torch.randn(1, L, D, generator=g)

    torch.randn(256, 256) @ torch.randn(256, 256)        # spin the thread pool up outside the clock

    def clock(fn):                                        # one timed call each: the sample is bounded
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    with torch.no_grad():
        t_m = clock(lambda: om.mamba_mixer_ref(sd, "m.", ocfg, h))
        t_a = clock(lambda: om.attention_mixer_ref(sd, "a.", ocfg, h))
        t_f = clock(lambda: om.mlp_mixer_ref(sd, "f.", h))
        # one ViT block (so400m dims) on vit_frames frames
        Dv, Hv, Mv, Np = 1152, 16, 4304, 729
        xv = torch.randn(vit_frames, Np, Dv, generator=g)
        wq, wp, w1, w2 = rn(3 * Dv, Dv), rn(Dv, Dv), rn(Mv, Dv), rn(Dv, Mv)

        def vit_block():
            qkv = (xv @ wq.t()).view(vit_frames, Np, 3, Hv, Dv // Hv)
            o, _ = R.attention_ref(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], False)
            y = xv + o.reshape(vit_frames, Np, Dv) @ wp.t()
            return y + torch.nn.functional.gelu(y @ w1.t()) @ w2.t()
        t_v = clock(vit_block)
    bt = cfg.layers_block_type
    llm_s = t_m * bt.count("mamba") + t_a * bt.count("attention") + t_f * bt.count("mlp")
    vit_s = t_v * 26 * (frames_sample / vit_frames)
    total = llm_s + vit_s
    return {"value": round(frames_sample / total, 4), "unit": "frames/s", "cores": cores,
            "kind": "port",
            "sample": (f"oracle (eager PyTorch fp32) at Nano-9B dims, {frames_sample} frames = {L} tokens: "
                       f"1 Mamba layer {t_m:.2f}s x27, 1 attention layer {t_a:.2f}s x4, 1 MLP layer "
                       f"{t_f:.2f}s x25, 1 SigLIP block on {vit_frames} frames {t_v:.2f}s x26 x{frames_sample // vit_frames}; "
                       f"no pdrop; scaled to a whole forward (extrapolated, attention is quadratic so this "
                       f"over-estimates CPU throughput at long lengths)")}


def config1_scan(args):
    """BASELINE configs[0]: one Mamba-2 selective scan, B=1 L=1024, 32 heads x 64, d_state 16, fp32 — the one configuration
    the reference's eager CPU path runs as it is.  Prints one JSON line: the HIP kernels' time (tv_ssd_scan_fwd picks the
    generic path's chunk-parallel form for this dtype / d_state), the oracle's time on the host cores, and the largest
    deviation of the HIP result from the oracle and of the oracle from the reference's golden fixture."""
    import numpy as np
    from timeviper_amd.build import ensure_built
    ensure_built()
    from oracle import ops as R
    from timeviper_amd import kernels as K
    dev = torch.device("cuda", 0)
    Bz, L, H, P, G, N = 1, 1024, 32, 64, 1, 16
    g = torch.Generator().manual_seed(0)
    x = torch.randn(Bz, L, H, P, generator=g)
    dt = torch.randn(Bz, L, H, generator=g) * 0.5
    A = -(torch.rand(H, generator=g) * 15 + 1)
    Bm, Cm = torch.randn(Bz, L, G, N, generator=g) * 0.5, torch.randn(Bz, L, G, N, generator=g) * 0.5
    D = torch.rand(H, generator=g) + 0.5
    dtv = torch.exp(torch.rand(H, generator=g) * (math.log(0.1) - math.log(1e-3)) + math.log(1e-3))
    bias = dtv + torch.log(-torch.expm1(-dtv))
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    R.ssd_chunk_scan_ref(x, dt, A, Bm, Cm, 128, D=D, dt_bias=bias)                     # warm the thread pool
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        y_ref, fin_ref = R.ssd_chunk_scan_ref(x, dt, A, Bm, Cm, 128, D=D, dt_bias=bias)[:2]
    cpu_ms = (time.perf_counter() - t0) / reps * 1e3
    d = lambda t: t.to(dev)
    fn = lambda: K.mamba_chunk_scan_combined(d(x), d(dt), d(A), d(Bm), d(Cm), chunk_size=128, D=d(D), dt_bias=d(bias),
                                             dt_softplus=True, return_final_states=True)
    xs = [d(t) for t in (x, dt, A, Bm, Cm, D, bias)]
    fn2 = lambda: K.mamba_chunk_scan_combined(xs[0], xs[1], xs[2], xs[3], xs[4], chunk_size=128, D=xs[5], dt_bias=xs[6],
                                              dt_softplus=True, return_final_states=True)
    for _ in range(max(args.warmup, 1)):
        y, fin = fn2()
    torch.cuda.synchronize()
    impl = K.ssd_scan_last_impl()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = max(args.steps, 20)
    e0.record()
    for _ in range(steps):
        y, fin = fn2()
    e1.record()
    torch.cuda.synchronize()
    us_eager = e0.elapsed_time(e1) / steps * 1e3        # host-paced: the Python operator call costs more than the kernels
    # the kernels' own time: `steps` scans captured into one hipGraph and replayed (a launch-bound operator is run that way)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(steps):
            y, fin = fn2()
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / steps * 1e3
    err_y = float((y.float().cpu() - y_ref).abs().max())
    err_f = float((fin.cpu() - fin_ref).abs().max())
    gold = np.load(ROOT / "tests" / "golden" / "mixer_g1.npz")
    A_g = -torch.exp(torch.from_numpy(gold["w.A_log"]))
    yg = R.ssd_chunk_scan_ref(*(torch.from_numpy(gold[k]) for k in ("scan_x", "scan_dt")), A_g,
                              torch.from_numpy(gold["scan_B"]), torch.from_numpy(gold["scan_C"]), 16,
                              D=torch.from_numpy(gold["w.D"]), dt_bias=torch.from_numpy(gold["w.dt_bias"]))[0]
    err_gold = float((yg - torch.from_numpy(gold["scan_y"])).abs().max())
    bytes_alg = Bz * L * (2 * H * P + H + 2 * G * N) * 4
    print(json.dumps({
        "metric": "Mamba-2 selective scan, B=1 L=1024 32x64 d_state 16 fp32 (BASELINE configs[0])",
        "value": round(us, 1), "unit": "us per scan", "n_gpus": 1, "steps": steps, "warmup": max(args.warmup, 1),
        "ms_per_step": round(us / 1e3, 4), "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "tv_ssd_scan_fwd (d_state 16 is not an MFMA shape: the generic path, chunk-parallel form — "
                               "ssd_chunk_local_kernel + ssd_chunk_carry_kernel, csrc/ssd_chunked.hip), "
                               f"tokens {L}, heads {H} x {P}, groups {G}, d_state {N}; scan implementation {impl}",
                   "timing": f"{steps} scans captured in one hipGraph and replayed; the eager Python loop takes "
                             f"{us_eager:.1f} us per scan (host-paced)"},
        "roofline": {"bound": "hbm", "kernel": "ssd_chunk_local_kernel + ssd_chunk_carry_kernel" if impl == 8 else "ssd_generic_kernel",
                     "achieved": round(bytes_alg / us / 1e3, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bytes_alg / us / 1e3 / HBM_PEAK_GBS, 5),
                     "traffic": None, "note": "0.3 MB per scan: a launch-latency-sized problem, not a bandwidth one"},
        "cpu_baseline": {"value": round(cpu_ms, 3), "unit": "ms per scan", "cores": cores, "kind": "port",
                         "sample": f"oracle.ops.ssd_chunk_scan_ref (the reference's chunked formulation, modeling_nano.py:775-851), "
                                   f"mean of {reps} calls"},
        "max_abs_err": {"hip_vs_oracle_y": err_y, "hip_vs_oracle_final_state": err_f,
                        "oracle_vs_reference_golden_y (tests/golden/mixer_g1.npz)": err_gold}}), flush=True)


def config_decode(args):
    """One generated token against the cache of a 2 048-frame prefill (32 868 tokens, no token drop): the decode step
    of `generate()` (modeling_nano.py:484-546, 1666-1689) on tv_causal_conv1d_update, tv_selective_state_update and
    tv_attn_decode_fwd.  `value`: tokens/s of the step replayed as ONE hipGraph (llm/decode_graph.py: static K / V
    buffers, key count on the device), greedy, the token fed back; the host-driven loop is timed beside it
    (`config.eager_tokens_per_s`); the per-kernel rooflines come from the launches of one token, each operator's replayed
    back to back as a graph of its own between two events: the
    matrix-vector products (HBM: every weight matrix once per token — the dominant kernel, 16.6 GB of the token's
    17.5 GB), the state update (the fp32 state of every head read and written once per token, 2 x 5.24 MB per Mamba
    layer) and the split-KV attention (K and V of the cache read once per token and attention layer)."""
    from timeviper_amd.build import ensure_built
    ensure_built()
    from timeviper_amd import kernels as K
    from timeviper_amd.model import build_synthetic_timeviper
    from timeviper_amd.model.llm.decode_graph import GraphedDecodeStep
    from timeviper_amd.model.llm.nano import HybridMambaAttentionDynamicCache, NemotronHConfig
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    cfg = NemotronHConfig.nemotron_nano_9b_v2()
    vlm = build_synthetic_timeviper(cfg, "siglip-vit-so400m-384px", pdrop_type=None, merge_module="no_merge",
                                    device=dev, seed=0)
    llm = vlm.llm_backbone.llm
    L = 2048 * TOK_PER_FRAME + 100
    g = torch.Generator(device=dev).manual_seed(1)
    emb = (torch.randn(1, L, cfg.hidden_size, device=dev, generator=g) * 0.02).bfloat16()
    steps, warm = max(args.steps, 16), max(args.warmup, 4)
    host_pos = torch.ones(1, dtype=torch.long)
    calls = {"ssu": [], "attn": [], "gemv": []}

    def recorded(name, orig, nbytes):
        def f(*a, **kw):
            calls[name].append((orig, a, kw, nbytes(*a, **kw)))
            return orig(*a, **kw)
        return f

    def roofline(name, kernel):
        """The launches of one token's operator `name`, with the arguments the step gave them, replayed back to back as
        one hipGraph between two events on the replay stream: per-launch time as inside the step's graph (events around
        single calls of the host loop measured the launch gaps of 4 - 30 us kernels, not the kernels)."""
        rec = calls[name]
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for fn, a, kw, _ in rec:
                fn(*a, **kw)
            st.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=st):
                for fn, a, kw, _ in rec:
                    fn(*a, **kw)
            graph.replay()
            st.synchronize()
            reps = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                graph.replay()
            e1.record(st)
            st.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = sum(b for *_, b in rec)
        gbs = nbytes / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "launches": len(rec), "replays": reps,
                "avg_launch_us": round(ms * 1e3 / len(rec), 2), "bytes_per_launch": round(nbytes / len(rec))}
    with torch.inference_mode():
        cache = HybridMambaAttentionDynamicCache(cfg, 1, dtype=torch.bfloat16, device=dev)
        out = llm(inputs_embeds=emb, past_key_values=cache, use_cache=True, cache_position=torch.zeros(1, dtype=torch.long))
        tok = out.logits[:, -1].argmax(-1).view(1, 1)
        # ---- host-driven loop
        for _ in range(warm):
            tok = llm(input_ids=tok, past_key_values=cache, use_cache=True, cache_position=host_pos).logits[:, -1].argmax(-1).view(1, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tok = llm(input_ids=tok, past_key_values=cache, use_cache=True, cache_position=host_pos).logits[:, -1].argmax(-1).view(1, 1)
        torch.cuda.synchronize()
        eager_s = time.perf_counter() - t0
        # ---- the same step as one graph launch per token
        gsteps = max(steps, 64)
        cache.begin_static_decode(warm + gsteps + 8)
        stepper = GraphedDecodeStep(
            lambda ids: llm(input_ids=ids, past_key_values=cache, use_cache=True, cache_position=host_pos).logits,
            cache, 1, dev)
        for _ in range(warm):                 # two eager steps in static mode, the capture, replays
            tok = stepper.step(tok).view(1, 1)
        torch.cuda.synchronize()
        assert stepper.graph is not None
        t0 = time.perf_counter()
        for _ in range(gsteps):
            tok = stepper.step(tok).view(1, 1)
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        out = stepper.logits.clone()
        # ---- per-operator rooflines: one more host-driven token with the operator calls recorded, each operator's launches
        # of that token replayed as a graph of their own (after everything that is timed: the replays advance the states)
        orig_ssu, orig_attn, orig_gemv = K.selective_state_update, K.flash_attn_decode, K.gemv_fused
        K.gemv_fused = recorded("gemv", orig_gemv, lambda x, w, *a, **kw: w.numel() * w.element_size())
        K.selective_state_update = recorded("ssu", orig_ssu, lambda state, *a, **kw: 2 * state.numel() * state.element_size())
        K.flash_attn_decode = recorded("attn", orig_attn, lambda q, k, v, *a, **kw: 2 * k.shape[0] * k.shape[1] * k.shape[2] * k.shape[3] * k.element_size())
        try:
            llm(input_ids=tok, past_key_values=cache, use_cache=True, cache_position=host_pos)
        finally:
            K.selective_state_update, K.flash_attn_decode, K.gemv_fused = orig_ssu, orig_attn, orig_gemv
        torch.cuda.synchronize()
        rl_gemv = roofline("gemv", "gemv_rows_kernel / gemv_bf16_kernel / gemv_splitk_reg_kernel (tv_gemv_bf16_fwd: every linear layer of the "
                                   "token, norm / activation prologues, conv-update epilogue; bytes: the weight matrices)")
        rl_ssu = roofline("ssu", "state_update_chain_kernel (tv_selective_state_update)")
        rl_attn = roofline("attn", "attn_decode_kernel + attn_decode_merge_kernel (tv_attn_decode_fwd)")
    assert torch.isfinite(out.float()).all()
    weights = sum(p.numel() * p.element_size() for n, p in llm.named_parameters() if "embed" not in n)
    print(json.dumps({
        "metric": "decode tokens/s, TimeViper-9B, batch 1, one token against a 2 048-frame prefill cache",
        "value": round(gsteps / dt_s, 2), "unit": "tokens/s", "n_gpus": 1, "steps": gsteps, "warmup": warm,
        "ms_per_step": round(dt_s / gsteps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"Nemotron-Nano-9B-v2 hybrid decode step (27 Mamba2 / 25 MLP / 4 attention layers), cache of "
                               f"{L} tokens (2 048 frames, no token drop), greedy, one hipGraph launch per token, matrix-vector "
                               f"products on tv_gemv_bf16_fwd (norm / activation prologues)",
                   "cache_tokens": L, "weights": "random init, seed 0",
                   "eager_tokens_per_s": round(steps / eager_s, 2), "eager_ms_per_step": round(eager_s / steps * 1e3, 3),
                   "weight_bytes_per_token": weights,
                   "weight_stream_floor_ms": round(weights / (HBM_PEAK_GBS * 1e9) * 1e3, 3)},
        "roofline": rl_gemv, "rooflines": [rl_gemv, rl_ssu, rl_attn]}), flush=True)


class CudaEnv:
    """What `run` needs from the platform: the device of this rank, the collective backend, a device synchronisation,
    event timers on the launch stream, the model and its input size.  tests/bench_dryrun.py supplies a CPU / gloo
    stand-in (oracle-backed kernel shims, a toy model) so that the launcher, the sharded step, the max-over-ranks timing
    and the one-JSON-line contract of `--gpus N` run on a box without GPUs; THIS class is the product."""
    backend = "nccl"
    pixels = 384

    def __init__(self, local_rank: int):
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)

    def init_process_group(self, world, rank):
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=self.dev)
        else:
            dist.init_process_group("nccl", device_id=self.dev)

    def sync(self):
        torch.cuda.synchronize()

    def event(self):
        return torch.cuda.Event(enable_timing=True)

    def kernels(self):
        import contextlib
        return contextlib.nullcontext()

    def ensure_built(self):
        from timeviper_amd.build import ensure_built
        ensure_built()

    def build_model(self, pd):
        from timeviper_amd.model import build_synthetic_timeviper
        from timeviper_amd.model.llm.nano import NemotronHConfig
        cfg = NemotronHConfig.nemotron_nano_9b_v2()
        vlm = build_synthetic_timeviper(cfg, "siglip-vit-so400m-384px", pdrop_type=pd,
                                        merge_module="CrossAttention" if pd else "no_merge",
                                        device=self.dev, seed=0)
        return cfg, vlm, ("TimeViper-Nano-9B forward (prefill), {T} frames x 384px, SigLIP-so400m ViT (26 blocks) + ToMe "
                          "729->16 + 56-layer Nemotron-Nano-9B-v2 hybrid (27 Mamba2 / 25 MLP / 4 attention), {L} tokens, batch 1")


def launch_children(script: str, argv, gpus: int) -> int:
    """`python bench.py --gpus N` typed by hand: start the N ranks as a CHILD job (one process per GPU,
    torch.distributed.run) and relay its output and exit code.  Nothing has touched the GPU in this process yet
    (importing torch does not), and nothing is exec'ed over it."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(script).resolve()), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=int(os.environ.get("TV_BENCH_FRAMES", 10240)))
    ap.add_argument("--no-pdrop", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="0",
                    help="one of BASELINE.json's other configurations, counted from 1 (1 scan only, 2 256 frames, "
                         "3 2 048 frames + TransV + pdrop, 5 Qwen2.5 + dual encoder + fp8 attention), or `decode` (one "
                         "generated token against a 2 048-frame prefill cache); default: the headline run (10 240 "
                         "frames); 4 is the headline run with --gpus 8")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.config == "decode":
        return config_decode(args)
    try:
        args.config = int(args.config)
    except ValueError:
        sys.exit("bench.py: --config takes 1, 2, 3, 4, 5 or decode")
    if args.config == 1:
        return config1_scan(args)
    if args.config == 2:
        args.frames, args.no_pdrop = 256, True
    elif args.config == 3:
        args.frames = 2048
    elif args.config == 5:
        # a child process (nothing has touched the GPU here): devtools/run_config5.py prints the JSON line
        import subprocess
        sys.exit(subprocess.run([sys.executable, str(ROOT / "timeviper_amd" / "devtools" / "run_config5.py"), "4096",
                                 str(max(args.steps, 1)), "1"], cwd=str(ROOT)).returncode)
    elif args.config not in (0, 4):
        sys.exit("bench.py: --config takes 1, 2, 3, 4, 5 or decode")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_children(__file__, sys.argv[1:], args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    run(args, CudaEnv(int(os.environ.get("LOCAL_RANK", "0"))))


def run(args, env):
    """The timed job on this rank (one process per GPU); rank 0 prints the ONE JSON line."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with\n  python -m "
                 f"torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                 f"--master-port 29500 bench.py --gpus {args.gpus} ...")
    dev = env.dev
    # TV_BENCH_FORCE_SP=1 (dev): run the sequence-sharded runner and its RCCL collectives even
    # with one rank, so the N>1 code path can be exercised on a 1-GPU box
    sharded = world > 1 or os.environ.get("TV_BENCH_FORCE_SP") == "1"
    if sharded:
        env.init_process_group(world, rank)

    if local_rank == 0:         # a tree without build artefacts: compile once, the others wait
        env.ensure_built()
    if sharded:
        torch.distributed.barrier()
    from timeviper_amd import kernels as K

    pd = None if args.no_pdrop else PDROP
    cfg, vlm, workload = env.build_model(pd)
    T = args.frames
    g = torch.Generator(device=dev).manual_seed(1)
    tok = vlm.default_token_id
    hi_id = min(1000, cfg.vocab_size - 1)
    ids = torch.cat([torch.randint(3, hi_id, (20,), device=dev, generator=g),
                     torch.full((T,), tok, device=dev),
                     torch.randint(3, hi_id, (80,), device=dev, generator=g)])[None]
    px, pdt = env.pixels, (torch.bfloat16 if dev.type == "cuda" else torch.float32)
    if sharded:
        from timeviper_amd.distributed import SequenceParallelTimeViper
        runner = SequenceParallelTimeViper(vlm, rank, world)
        lo, hi = runner.frame_range(T)
        pix = torch.randn(hi - lo, 3, px, px, device=dev, dtype=pdt, generator=g)
        step = lambda: runner.forward(ids, pix, T)
    else:
        pix = torch.randn(T, 3, px, px, device=dev, dtype=pdt, generator=g)
        step = lambda: vlm(input_ids=ids, pixel_values_videos=pix).logits

    def barrier():
        if sharded:
            torch.distributed.barrier()
        env.sync()

    with torch.inference_mode(), env.kernels(), OpTimers(K, env.event, vlm) as st:
        for _ in range(args.warmup):
            step()
        barrier()
        st.on = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        dt_s = time.perf_counter() - t0
        st.on = False
    assert torch.isfinite(out.float()).all(), "non-finite logits"
    if os.environ.get("TV_BENCH_DRYRUN_FAIL_RANK") == str(rank):          # tests: a rank that dies must fail the job
        raise RuntimeError("TV_BENCH_DRYRUN_FAIL_RANK: this rank fails on purpose")
    t = torch.tensor([dt_s], device=dev, dtype=torch.float64)
    if sharded:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt_s = float(t.item())

    if rank == 0:
        ms = dt_s / args.steps * 1e3
        L = T * TOK_PER_FRAME + 100
        line = {
            "metric": "video frames/sec fwd, TimeViper-9B @10k frames, 1/2/4/8 MI355X",
            "value": round(T * args.steps / dt_s, 2), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": workload.format(T=T, L=L),
                       "frames": T, "tokens": L, "pdrop": pd, "merge_module": "CrossAttention" if pd else "no_merge",
                       "parallelism": "single GPU" if world == 1 else f"sequence-sharded x{world} (RCCL)",
                       "weights": "random init, seed 0"},
            # the kernel north_star names (dominant among the hand-written HBM-bound ones) ...
            "roofline": st.scan_roofline(scan_bytes_per_token(cfg)),
            # ... and every kernel with a stated roof, all event-timed inside the timed steps
            "rooflines": st.all_rooflines(scan_bytes_per_token(cfg), cfg),
        }
        # how much of the step the event-timed operators (hand-written kernels and library GEMMs) account for
        line["timed_ms_per_step"] = round(st.accounted_ms() / args.steps, 2)
        line["timed_share_of_step"] = round(st.accounted_ms() / args.steps / ms, 4)
        if world == 1 and dev.type == "cuda" and line["roofline"]:
            # What a plain copy of the same size reaches on THIS box (read + write streams, outside the timed region): the
            # practical ceiling of an operator that reads x and writes y once — `peak` stays the 8 TB/s of the data sheet.
            line["roofline"].update(copy_rate(dev, line["roofline"]))
        if world == 1 and not args.no_cpu_baseline and dev.type == "cuda":
            line["cpu_baseline"] = cpu_baseline(cfg)
    # RCCL writes a version banner through C stdio, which is block-buffered on a pipe and would
    # otherwise surface at process exit, AFTER the result: every rank drains it before the last
    # barrier, so that rank 0's JSON line is the last line of the job's stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if sharded:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
