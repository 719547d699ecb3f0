"""FP8 (e4m3) MFMA attention variant — BASELINE config 5, `tv_flash_attn_fp8_fwd`.

The reference's attention arithmetic is bf16, so this opt-in variant is judged by stated tolerances:
  (a) against the oracle attention evaluated ON THE QUANTISED INPUTS (oracle.ops.fp8_quantise_ref:
      the same per-(batch, head) e4m3 quantiser): what is left is the e4m3 rounding of P (2^-4
      relative per element, averaged over the keys of a row), the fp32 accumulation order and the
      bf16 output rounding;
  (b) against the bf16 kernel on the original inputs: adds the e4m3 rounding of q, k, v.
Both bounds are written below as relative L2 error of the whole output and as a per-element bound."""
import math

import pytest
import torch

from oracle import ops as R

pytestmark = pytest.mark.gpu
DEV = "cuda"

# Error model (random data, the worst case: nothing for the rounding errors to average against).
# e4m3 keeps 3 mantissa bits: relative rounding error uniform in +-2^-4, rms 2^-4 / sqrt(3) = 3.6 %.
# Rounding P perturbs every term of sum_k p_k v_k independently -> 3.6 % of the output's norm; rounding
# V the same again; rounding q and k perturbs the logits (a further ~1-2 %); the matrix pipe's own fp8
# accumulation is good to ~1e-3 relative (measured through the log-sum-exp below).
#   (a) vs the oracle on the quantised inputs (P rounding + pipe):        rel L2 < 4.5e-2
#   (b) vs the bf16 kernel on the original inputs (P, V, q, k rounding):  rel L2 < 7e-2
# per element: |err| <= atol * max|v| + rtol * |ref|
TOL_VS_QUANTISED_ORACLE = dict(rel_l2=4.5e-2, rtol=6e-2, atol=5e-2)
TOL_VS_BF16_KERNEL = dict(rel_l2=7e-2, rtol=1e-1, atol=8e-2)


@pytest.fixture(scope="module")
def K():
    from timeviper_amd import kernels
    return kernels


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()


def check(o, ref, tol, what, vmax=4.5):
    o, ref = o.double().cpu(), ref.double().cpu()
    assert o.shape == ref.shape
    e = rel_l2(o, ref)
    assert e < tol["rel_l2"], f"{what}: relative L2 error {e:.3e}"
    bad = (o - ref).abs() > tol["atol"] * vmax + tol["rtol"] * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())} / {bad.numel()} elements out of tolerance, max |err| {(o - ref).abs().max():.3e}"


SHAPES = [
    (1, 300, 300, 4, 2, 128, True),
    (1, 1000, 1000, 14, 2, 128, True),      # Qwen2.5-7B head ratio 7:1
    (2, 77, 500, 4, 4, 64, False),          # few queries (4-wave workgroups), DINOv2-like head_dim
    (3, 729, 729, 2, 2, 72, False),         # SigLIP ViT heads
    (2, 1025, 1025, 2, 2, 88, False),       # InternVideo2 ViT heads
    (1, 50, 700, 8, 2, 128, True),          # bottom-right aligned causal (Lk > Lq)
    (1, 257, 129, 2, 1, 96, False),
    (1, 640, 640, 5, 1, 128, True),         # Nemotron head ratio 5:1, whole stages
]


@pytest.mark.parametrize("B,Lq,Lk,Hq,Hkv,D,causal", SHAPES)
def test_fp8_attention_vs_oracle_on_quantised_inputs_and_vs_bf16(K, B, Lq, Lk, Hq, Hkv, D, causal):
    g = torch.Generator().manual_seed(Lq * 5 + Lk + D)
    q = torch.randn(B, Lq, Hq, D, generator=g).bfloat16()
    k = torch.randn(B, Lk, Hkv, D, generator=g).bfloat16()
    v = torch.randn(B, Lk, Hkv, D, generator=g).bfloat16()
    o8, lse8 = K.flash_attn_fp8_func(q.to(DEV), k.to(DEV), v.to(DEV), causal=causal, return_lse=True)
    assert torch.isfinite(o8.float()).all()
    # (a) the kernel's arithmetic: oracle on the same quantised values
    qd, kd, vd = R.fp8_quantise_ref(q), R.fp8_quantise_ref(k), R.fp8_quantise_ref(v)
    o_ref, lse_ref = R.attention_ref(qd, kd, vd, causal)
    check(o8, o_ref, TOL_VS_QUANTISED_ORACLE, "fp8 vs oracle on quantised inputs")
    # the row sum is taken over the unrounded P, so the log-sum-exp sees only the matrix pipe's fp8
    # accumulation of q~ . k~ (not a plain fp32 fmaf chain: ~1e-3 relative on a logit)
    assert (lse8.cpu() - lse_ref).abs().max() < 2e-2, "lse"
    # (b) end to end: the bf16 kernel on the original inputs
    o16 = K.flash_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), causal=causal)
    check(o8, o16.float(), TOL_VS_BF16_KERNEL, "fp8 vs bf16 kernel")


def test_fp8_attention_spiked_max_and_scale_spread(K):
    """Forces large running-max jumps at chosen tiles (the rescale branch) and heads whose magnitudes
    differ by 100x (per-head scales)."""
    g = torch.Generator().manual_seed(0)
    B, L, H, D = 1, 700, 2, 128
    q = torch.randn(B, L, H, D, generator=g)
    k = torch.randn(B, L, H, D, generator=g)
    v = torch.randn(B, L, H, D, generator=g)
    for pos in (70, 300, 650):
        k[0, pos, :, :] = q[0, 699, :, :] * 3.0
    q[:, :, 1] *= 0.05
    v[:, :, 1] *= 20.0
    q, k, v = (t.bfloat16() for t in (q, k, v))
    o_ref, _ = R.attention_ref(R.fp8_quantise_ref(q), R.fp8_quantise_ref(k), R.fp8_quantise_ref(v), True)
    o8 = K.flash_attn_fp8_func(q.to(DEV), k.to(DEV), v.to(DEV), causal=True)
    for h in range(H):
        check(o8[:, :, h], o_ref[:, :, h], TOL_VS_QUANTISED_ORACLE, f"head {h}", vmax=float(v[:, :, h].float().abs().max()))


def test_fp8_switch_routes_flash_attn_func(K):
    """`kernels.fp8_attention()` sends flash_attn_func (and the model code built on it) to the fp8
    kernel; short key sequences (decode) and the default state stay on bf16."""
    g = torch.Generator(device=DEV).manual_seed(3)
    q = torch.randn(1, 512, 4, 128, device=DEV, generator=g).bfloat16()
    k = torch.randn(1, 512, 2, 128, device=DEV, generator=g).bfloat16()
    v = torch.randn(1, 512, 2, 128, device=DEV, generator=g).bfloat16()
    o16 = K.flash_attn_func(q, k, v, causal=True)
    with K.fp8_attention(min_keys=256):
        o8 = K.flash_attn_func(q, k, v, causal=True)
        assert torch.equal(o8, K.flash_attn_fp8_func(q, k, v, causal=True))
        short = K.flash_attn_func(q[:, :1], k[:, :100], v[:, :100], causal=True)     # 100 keys < min_keys: bf16
        decode = K.flash_attn_func(q[:, -1:], k, v, causal=True)                     # 1 query against 512 >= min_keys keys: bf16 too
    assert torch.equal(short, K.flash_attn_func(q[:, :1], k[:, :100], v[:, :100], causal=True))
    assert torch.equal(decode, K.flash_attn_func(q[:, -1:], k, v, causal=True)), "decode step must stay on the bf16 kernel, bit for bit"
    assert not torch.equal(o8, o16) and torch.equal(K.flash_attn_func(q, k, v, causal=True), o16)
