"""Standalone timing of the HIP operators at Nano-9B shapes (dev tool; GPU only).
    python timeviper_amd/devtools/bench_ops.py [--tokens 163940] [--ops scan,conv,gnorm,rmsnorm,attn,patch]"""
import argparse
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from timeviper_amd import kernels as K  # noqa: E402


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=163940)
    ap.add_argument("--ops", default="scan,conv,gnorm,rmsnorm,attn,patch,gather")
    ap.add_argument("--impl", type=int, default=0)
    ap.add_argument("--token-major", action="store_true", help="B/C as column slices of the conv rows")
    ap.add_argument("--model-dt", action="store_true", help="dt_bias / dt as in the 9B model's init (slowly forgetting heads)")
    ap.add_argument("--dt-std", type=float, default=0.02, help="with --model-dt: standard deviation of the raw dt (the synthetic 9B model: ~1.3)")
    ap.add_argument("--model-A", action="store_true", help="A = -(1 .. H) as the 9B model's A_log init (modeling_nano.py): fast-forgetting heads, standard steps")
    ap.add_argument("--no-cb", action="store_true", help="scan recomputes C.B^T in its pre-pass (round 2)")
    a = ap.parse_args()
    L = a.tokens
    dev = "cuda"
    H, P, G, N, Dm = 128, 80, 8, 128, 4480
    d_in, conv_dim = H * P, H * P + 2 * G * N
    ops = a.ops.split(",")
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g, dtype=torch.float32).bfloat16()
    if "scan" in ops or "conv" in ops or "gnorm" in ops:
        proj = rn(1, L, d_in + conv_dim + H)
        gate, xBC, dt = proj.split([d_in, conv_dim, H], dim=-1)
        w, b = rn(conv_dim, 4), rn(conv_dim)
    if "conv" in ops:
        ms = timeit(lambda: K.causal_conv1d_fn(xBC.transpose(1, 2), w, b, activation="silu"))
        by = L * 2 * 2 * conv_dim
        print(f"conv1d      {ms*1e3:9.1f} us  {by/ms/1e6:8.1f} GB/s  ({by/ms/1e6/8000:.1%} of 8 TB/s)")
    if "scan" in ops:
        K.ssd_scan_set_impl(a.impl)
        if a.token_major:
            conv = K.causal_conv1d_fn(xBC.transpose(1, 2), w, b, activation="silu").transpose(1, 2)
            x, Bm, Cm = conv.split([d_in, G * N, G * N], dim=-1)
            Bm, Cm = Bm.view(1, L, G, N), Cm.view(1, L, G, N)
            cb = None
        else:
            x, Bm, Cm, cb = K.causal_conv1d_xbc(xBC, w, b, d_in, G, N, return_cb=True)
            if a.no_cb:
                cb = None
            ms = timeit(lambda: K.causal_conv1d_xbc(xBC, w, b, d_in, G, N, return_cb=not a.no_cb))
            by = L * 2 * 2 * conv_dim
            print(f"conv1d xbc  {ms*1e3:9.1f} us  {by/ms/1e6:8.1f} GB/s  ({by/ms/1e6/8000:.1%} of 8 TB/s)"
                  f"  {'+ C.B^T fragments' if cb is not None else ''}")
        A = -(torch.rand(H, device=dev, generator=g) * 15 + 1)
        if a.model_A:
            A = -torch.arange(1, H + 1, device=dev, dtype=torch.float32)
        D = torch.ones(H, device=dev)
        if a.model_dt:      # the 9B model's initialisation (modeling_nano.py:1345-1357): dt in [1e-3, 0.1], slow heads exist
            dtv = torch.exp(torch.rand(H, device=dev, generator=g) * (math.log(0.1) - math.log(1e-3)) + math.log(1e-3))
            dtb = dtv + torch.log(-torch.expm1(-dtv))
            dt = (dt.float() * a.dt_std).bfloat16()
        else:
            dtb = torch.full((H,), -3.0, device=dev)
        fn = lambda: K.mamba_chunk_scan_combined(x.view(1, L, H, P), dt, A, Bm, Cm,
                                                 chunk_size=128, D=D, dt_bias=dtb,
                                                 dt_softplus=True, return_final_states=True, cb=cb)
        ms = timeit(fn, iters=5 if a.impl == 1 else 10)
        by = L * (2 * d_in * 2 + 2 * H + 4 * G * N)
        print(f"ssd_scan    {ms*1e3:9.1f} us  {by/ms/1e6:8.1f} GB/s  ({by/ms/1e6/8000:.1%} of 8 TB/s)  impl={a.impl}")
        K.ssd_scan_set_impl(0)
    if "gnorm" in ops:
        y = rn(1, L, d_in)
        wn = torch.ones(d_in, device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: K.rmsnorm_fn(y, wn, None, gate, 1e-5, d_in // G, norm_before_gate=False))
        by = L * 3 * 2 * d_in
        print(f"gated norm  {ms*1e3:9.1f} us  {by/ms/1e6:8.1f} GB/s  ({by/ms/1e6/8000:.1%} of 8 TB/s)")
    if "rmsnorm" in ops:
        hcur, dl = rn(1, L, Dm), rn(1, L, Dm)
        wn = torch.ones(Dm, device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: K.rms_norm(hcur, wn, 1e-5, residual=dl, return_sum=True))
        by = L * 4 * 2 * Dm
        print(f"add+rmsnorm {ms*1e3:9.1f} us  {by/ms/1e6:8.1f} GB/s  ({by/ms/1e6/8000:.1%} of 8 TB/s)")
    if "gather" in ops:
        hcur = rn(L, Dm)
        idx = torch.arange(0, L, 5, device=dev)
        ms = timeit(lambda: K.gather_rows(hcur, idx))
        by = idx.numel() * 2 * 2 * Dm
        print(f"gather      {ms*1e3:9.1f} us  {by/ms/1e6:8.1f} GB/s")
    if "attn" in ops:
        La = min(L, 32868)
        q, k, v = rn(1, La, 40, 128), rn(1, La, 8, 128), rn(1, La, 8, 128)
        ms = timeit(lambda: K.flash_attn_func(q, k, v, causal=True), iters=3, warmup=1)
        fl = 2 * La * La * 40 * 128 * 2 / 2
        print(f"attn causal L={La} {ms:9.2f} ms  {fl/ms/1e9:8.1f} TFLOP/s ({fl/ms/1e9/2500:.1%} of 2.5 PF)")
        qv = rn(256, 729, 3, 16, 72)
        ms = timeit(lambda: K.flash_attn_func(qv[:, :, 0], qv[:, :, 1], qv[:, :, 2], causal=False), iters=5)
        fl = 4 * 256 * 16 * 729 * 729 * 72
        print(f"attn ViT 256x729 d72 {ms:9.2f} ms  {fl/ms/1e9:8.1f} TFLOP/s")
    if "gelu" in ops:
        hx = rn(256 * 729, 4304)
        ms = timeit(lambda: K.gelu(hx, inplace=True))
        by = hx.numel() * 2 * 2
        print(f"gelu 186624x4304 {ms*1e3:9.1f} us  {by/ms/1e6:8.1f} GB/s  ({by/ms/1e6/8000:.1%} of 8 TB/s)")
        xl, dl_ = rn(256 * 729, 1152), rn(256 * 729, 1152)
        wl, bl = rn(1152), rn(1152)
        ms = timeit(lambda: K.layer_norm(xl, wl, bl, 1e-6, residual=dl_, return_sum=True))
        by = xl.numel() * 2 * 4
        print(f"layernorm+residual 186624x1152 {ms*1e3:9.1f} us  {by/ms/1e6:8.1f} GB/s  ({by/ms/1e6/8000:.1%} of 8 TB/s)")
    if "patch" in ops:
        pix = rn(256, 3, 384, 384)
        wp, bp, pos = rn(1152, 3, 14, 14), rn(1152), rn(729, 1152)
        ms = timeit(lambda: K.patch_embed(pix, wp, bp, pos), iters=5)
        fl = 2 * 256 * 729 * 588 * 1152
        by = pix.numel() * 2 + 256 * 729 * 1152 * 2
        print(f"patch embed 256 frames {ms:9.2f} ms  {fl/ms/1e9:8.1f} TFLOP/s  {by/ms/1e6:8.1f} GB/s")
    if "iv2" in ops:
        from timeviper_amd.model.vit.internvideo2 import InternVideo2ViTBackbone
        with torch.device("meta"):
            vb = InternVideo2ViTBackbone()
        vb = vb.to_empty(device=dev)
        with torch.no_grad():
            for n, p in vb.named_parameters():
                if n.endswith(("ls1.weight", "ls2.weight")) or "norm" in n:
                    p.fill_(1.0 if "norm" in n else 0.1)
                else:
                    p.normal_(0, 0.02, generator=g)
        vb = vb.bfloat16().eval()
        clip = rn(256, 1, 3, 224, 224)
        ms = timeit(lambda: vb(clip, is_video=True), iters=3, warmup=1)
        print(f"InternVideo2-1B tower, 256 frames (64 clips x 1025 tokens) {ms:9.2f} ms  "
              f"{256/ms*1e3:8.1f} frames/s")


if __name__ == "__main__":
    main()
