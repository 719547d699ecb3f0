"""Seeded random-shape sweeps of the three hot operators against the CPU oracle: ragged lengths
around the tile sizes (64-token chunks, 64-key tiles, 256-row query blocks), every supported
head_dim, GQA ratios, both head->group maps — the cases a fixed parametrisation misses."""
import random

import pytest
import torch

from oracle import ops as R
from test_ops_gpu import TOL, close, run_scan, scan_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def K():
    from timeviper_amd import kernels
    return kernels


def test_scan_random_shapes(K):
    rng = random.Random(1234)
    for it in range(24):
        slice_ok = it % 2 == 0
        if slice_ok:      # shapes the MFMA march kernels accept (bf16, d_state 128, P % 8 == 0)
            dtype, N = torch.bfloat16, 128
            P = rng.choice([24, 40, 48, 64, 80, 96, 128])
        else:
            dtype, N = rng.choice([torch.float32, torch.bfloat16]), rng.choice([16, 40, 64, 128])
            P = rng.choice([8, 16, 24, 64, 80])
        G = rng.choice([1, 2, 4])
        H = G * rng.choice([1, 2, 4])
        B = rng.choice([1, 1, 2])
        L = rng.choice([1, 2, 63, 64, 65, 127, 128, 129, rng.randint(1, 700)])
        gmap = rng.choice(["block", "tile"])
        with_init = rng.random() < 0.5
        ins = scan_inputs(B, L, H, P, G, N, 1000 + it, dtype)
        init = torch.randn(B, H, P, N, generator=torch.Generator().manual_seed(it)) if with_init else None
        f = [t.float() for t in ins]
        y_ref, fin_ref, dec_ref = R.ssd_recurrence_ref(*f[:5], D=f[5], dt_bias=f[6], initial_states=init,
                                                       group_map=gmap)
        y, fin, dec = run_scan(K, *ins, initial_states=None if init is None else init.to(DEV), group_map=gmap)
        rt, at = TOL[dtype]
        tag = f"#{it} B{B} L{L} H{H} P{P} G{G} N{N} {dtype} {gmap} init={with_init}"
        close(y, y_ref, rt, at * 2, "y " + tag)
        close(fin, fin_ref, rt, at, "final " + tag)
        close(dec, dec_ref, 1e-4, 1e-4, "decay " + tag)


def test_attention_random_shapes(K):
    rng = random.Random(99)
    for it in range(28):
        D = rng.choice([16, 32, 64, 72, 80, 88, 96, 128])
        Hkv = rng.choice([1, 2, 3])
        Hq = Hkv * rng.choice([1, 2, 5])
        B = rng.choice([1, 2])
        causal = rng.random() < 0.5
        Lq = rng.choice([1, 31, 32, 33, 127, 128, 129, 255, 256, 257, rng.randint(1, 600)])
        Lk = Lq + rng.choice([0, 0, 1, 63, 64, 65, rng.randint(0, 300)]) if causal else rng.randint(1, 700)
        dtype = rng.choice([torch.bfloat16, torch.float16])
        g = torch.Generator().manual_seed(500 + it)
        q = torch.randn(B, Lq, Hq, D, generator=g).to(dtype)
        k = torch.randn(B, Lk, Hkv, D, generator=g).to(dtype)
        v = torch.randn(B, Lk, Hkv, D, generator=g).to(dtype)
        o_ref, lse_ref = R.attention_ref(q.float(), k.float(), v.float(), causal)
        o, lse = K.flash_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), causal=causal, return_lse=True)
        rt, at = (2e-2, 1e-2) if dtype == torch.bfloat16 else (4e-3, 2e-3)
        tag = f"#{it} B{B} Lq{Lq} Lk{Lk} Hq{Hq} Hkv{Hkv} D{D} causal={causal} {dtype}"
        close(o, o_ref, rt, at, "o " + tag)
        close(lse, lse_ref, 1e-3, 2e-3, "lse " + tag)


def test_conv_random_shapes(K):
    rng = random.Random(7)
    for it in range(16):
        B, L = rng.choice([1, 2]), rng.choice([1, 2, 3, 4, 63, 64, 65, rng.randint(1, 500)])
        C = 8 * rng.randint(1, 200)
        dtype = rng.choice([torch.float32, torch.bfloat16])
        g = torch.Generator().manual_seed(it)
        x = torch.randn(B, L, C, generator=g).to(dtype)
        w = (torch.randn(C, 4, generator=g) * 0.5).to(dtype)
        b = torch.randn(C, generator=g).to(dtype)
        halo = torch.randn(B, 3, C, generator=g).to(dtype) if rng.random() < 0.5 else None
        y_ref = R.causal_conv1d_ref(x.float(), w.float(), b.float(), "silu", None if halo is None else halo.float())
        y = K.causal_conv1d_fn(x.to(DEV).transpose(1, 2), w.to(DEV), b.to(DEV), activation="silu",
                               halo=None if halo is None else halo.to(DEV)).transpose(1, 2)
        close(y, y_ref, *TOL[dtype], f"#{it} B{B} L{L} C{C} {dtype} halo={halo is not None}")
