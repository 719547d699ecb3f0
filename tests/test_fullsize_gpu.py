"""BASELINE.json's full sizes (10 240 frames -> 163 940 tokens, Nemotron-Nano-9B-v2 dims) through
size-independent properties: the oracle cannot run here in seconds, so the HIP kernels are checked
against themselves (shard chaining, exact power-of-two linearity, two independent kernels) and
against the definition on sampled rows."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
L_FULL = 10240 * 16 + 100
H, P, G, N = 128, 80, 8, 128


@pytest.fixture(scope="module")
def K():
    from timeviper_amd import kernels
    return kernels


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def scan_inputs(L, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    x = rn(1, L, H, P).bfloat16()
    dt = (rn(1, L, H) * 0.5).bfloat16()
    Bm = (rn(1, G, L, N) * 0.5).bfloat16().transpose(1, 2)     # group-major storage like the conv kernel's
    Cm = (rn(1, G, L, N) * 0.5).bfloat16().transpose(1, 2)
    A = -(torch.rand(H, device=DEV, generator=g) * 15 + 1)
    D = torch.rand(H, device=DEV, generator=g) + 0.5
    dtv = torch.exp(torch.rand(H, device=DEV, generator=g) * (math.log(0.1) - math.log(1e-3)) + math.log(1e-3))
    return x, dt, A, Bm, Cm, D, dtv + torch.log(-torch.expm1(-dtv))


def run(K, x, dt, A, Bm, Cm, D, bias, **kw):
    return K.mamba_chunk_scan_combined(x, dt, A, Bm, Cm, chunk_size=128, D=D, dt_bias=bias, dt_softplus=True,
                                       return_final_states=True, return_total_decay=True, **kw)


def test_scan_full_length_properties(K):
    x, dt, A, Bm, Cm, D, bias = scan_inputs(L_FULL)
    y, fin, dec = run(K, x, dt, A, Bm, Cm, D, bias)
    assert torch.isfinite(y.float()).all() and torch.isfinite(fin).all()
    # (1) y and the final state are linear in x: scaling by 2 is exact in bf16 and fp32 — up to the contributions that a
    # reset step of the head-per-wave march builds at the bottom of the exponent range (a chunk that decays by more than
    # 2^-64 starts its state in the frame E = 100: weights below 2^-126, i.e. tokens whose true weight is < 2^-26 of the
    # chunk's last, are flushed, and whether a product lands above or below that line depends on the scale of x): the last
    # bit of a sum may differ, nothing more
    y2, fin2, _ = run(K, x * 2, dt, A, Bm, Cm, D, bias)
    ne = (y2 != y * 2)
    assert ne.float().mean() < 1e-3, ne.float().mean()
    assert ((y2.float() - 2 * y.float()).abs() <= 2.0 ** -7 * (2 * y.float()).abs() + 1e-30).all()      # one bf16 step
    assert rel(fin2, fin * 2) < 1e-6
    del y2, fin2, ne
    # (2) shard chaining: two shards with the state handed over == one pass
    s = 81 * 1000 + 37                                   # not a multiple of the chunk length
    ya, fa, da = run(K, x[:, :s], dt[:, :s], A, Bm[:, :s], Cm[:, :s], D, bias)
    yb, fb, db = run(K, x[:, s:], dt[:, s:], A, Bm[:, s:], Cm[:, s:], D, bias, initial_states=fa)
    assert rel(torch.cat([ya, yb], 1), y) < 5e-3
    # (the march rounds x~ = w_s x to bf16 before the state update, ~2e-3 on the state — check (4); in the head-per-wave
    # kernel w_s carries the floating frame's scale, which depends on where a march started, so two marches over the same
    # tokens round differently: the states agree to that rounding, not to fp32)
    assert rel(fb, fin) < 5e-3
    assert torch.allclose(da + db, dec, rtol=1e-4, atol=1e-2)
    del ya, yb
    # (3) the two MFMA kernels (head-per-wave march / whole-head slice march) are independent implementations
    K.ssd_scan_set_impl(4)
    try:
        ym, fm, dm = run(K, x, dt, A, Bm, Cm, D, bias)
    finally:
        K.ssd_scan_set_impl(0)
    assert rel(ym, y) < 5e-3 and rel(fm, fin) < 5e-3
    assert torch.allclose(dm, dec, rtol=1e-5, atol=1e-3)
    del ym
    # (4) the definition (fp32 token recurrence, generic kernel) on the last 3 000 tokens,
    #     started from the state the full pass had there
    t0 = L_FULL - 3000
    _, f0, _ = run(K, x[:, :t0], dt[:, :t0], A, Bm[:, :t0], Cm[:, :t0], D, bias)
    K.ssd_scan_set_impl(1)
    try:
        yr, fr, _ = run(K, x[:, t0:], dt[:, t0:], A, Bm[:, t0:], Cm[:, t0:], D, bias, initial_states=f0)
    finally:
        K.ssd_scan_set_impl(0)
    # (the MFMA kernels round x~ = w_t x to bf16 before the state update: ~2e-3 on the state)
    assert rel(y[:, t0:], yr) < 5e-3 and rel(fin, fr) < 5e-3


def test_conv_full_length_shard_halo(K):
    conv_dim = H * P + 2 * G * N
    g = torch.Generator(device=DEV).manual_seed(1)
    xBC = torch.randn(1, L_FULL, conv_dim, device=DEV, generator=g).bfloat16()
    w = torch.randn(conv_dim, 4, device=DEV, generator=g).bfloat16()
    b = torch.randn(conv_dim, device=DEV, generator=g).bfloat16()
    x, Bm, Cm = K.causal_conv1d_xbc(xBC, w, b, H * P, G, N)
    s = 70001
    xa, Ba, Ca = K.causal_conv1d_xbc(xBC[:, :s], w, b, H * P, G, N)
    xb, Bb, Cb = K.causal_conv1d_xbc(xBC[:, s:], w, b, H * P, G, N, halo=xBC[:, s - 3:s].contiguous())
    assert torch.equal(torch.cat([xa, xb], 1), x)
    assert torch.equal(torch.cat([Ba, Bb], 1), Bm) and torch.equal(torch.cat([Ca, Cb], 1), Cm)
    # against the definition on a window
    t = slice(s - 8, s + 8)
    ref = torch.nn.functional.conv1d(xBC[:, s - 11:s + 8].float().transpose(1, 2), w.float()[:, None], b.float(),
                                     groups=conv_dim)
    ref = torch.nn.functional.silu(ref).transpose(1, 2)
    assert rel(x[:, t], ref[..., :H * P]) < 1e-2


def test_attention_full_length_rows_and_shards(K):
    L = L_FULL
    g = torch.Generator(device=DEV).manual_seed(2)
    q = torch.randn(1, L, 40, 128, device=DEV, generator=g).bfloat16()
    k = torch.randn(1, L, 8, 128, device=DEV, generator=g).bfloat16()
    v = torch.randn(1, L, 8, 128, device=DEV, generator=g).bfloat16()
    o = K.flash_attn_func(q, k, v, causal=True)
    assert torch.isfinite(o.float()).all()
    # shard property: the second half's queries against all keys, bottom-right aligned
    s = 90000 + 13
    ob = K.flash_attn_func(q[:, s:], k, v, causal=True)
    assert torch.equal(ob, o[:, s:])
    # definition on sampled rows (fp32 softmax over the visible keys)
    for i in (0, 1, 63, 64, 12345, s, L - 1):
        qi = q[0, i].float().view(8, 5, 128)                       # GQA: 5 query heads per kv head
        sc = torch.einsum("ghd,lgd->ghl", qi, k[0, :i + 1].float()) / math.sqrt(128)
        ref = torch.einsum("ghl,lgd->ghd", torch.softmax(sc, -1), v[0, :i + 1].float()).reshape(40, 128)
        assert rel(o[0, i], ref) < 2e-2, i


def test_gated_norm_and_rmsnorm_full_length_rows(K):
    g = torch.Generator(device=DEV).manual_seed(3)
    L = L_FULL
    x = torch.randn(L, H * P, device=DEV, generator=g).bfloat16()
    z = torch.randn(L, H * P, device=DEV, generator=g).bfloat16()
    w = (1 + 0.1 * torch.randn(H * P, device=DEV, generator=g)).bfloat16()
    y = K.rmsnorm_fn(x, w, None, z, 1e-5, H * P // G, norm_before_gate=False)
    rows = torch.tensor([0, 1, 77777, L - 1], device=DEV)
    xs = (x[rows].float() * torch.nn.functional.silu(z[rows].float())).view(4, G, -1)
    ref = (xs * torch.rsqrt(xs.pow(2).mean(-1, keepdim=True) + 1e-5)).view(4, -1) * w.float()
    assert rel(y[rows], ref) < 1e-2
    # rows are independent: any permutation of the rows permutes the output
    perm = torch.randperm(4096, device=DEV, generator=g)
    y2 = K.rmsnorm_fn(x[:4096][perm].contiguous(), w, None, z[:4096][perm].contiguous(), 1e-5, H * P // G,
                      norm_before_gate=False)
    assert torch.equal(y2, y[:4096][perm])


def test_vit_fused_clips_equal_reference_clips(K):
    """SigLIP-so400m widths at 384 px, 2 048 frames in ONE call (what `encode_vision` does for
    frame-independent towers) against the reference's 256-frame clips (generic_vlm.py:274-281):
    same rows for every frame — launches past 2^31 elements (fc1 output: 6.4e9) index correctly."""
    from timeviper_amd.model import build_synthetic_timeviper
    from timeviper_amd.model.llm.nano import NemotronHConfig
    cfg = NemotronHConfig(vocab_size=128, hidden_size=64, intermediate_size=96, num_hidden_layers=2,
                          hybrid_override_pattern="M-", num_attention_heads=4, head_dim=16,
                          num_key_value_heads=2, ssm_state_size=16, mamba_num_heads=8,
                          mamba_n_groups=2, mamba_head_dim=8, mamba_chunk_size=16)
    vlm = build_synthetic_timeviper(cfg, "siglip-vit-so400m-384px", vit_depth=3)   # 2 blocks run
    assert vlm.vit_clip_frames == 256 and vlm.vit_clip_fuse == 8
    T = 2048 + 40
    pix = torch.randn(T, 3, 384, 384, device=DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        feats = vlm.vision_backbone(pix[:2048], is_video=True)           # one 2 048-frame launch
        vis = vlm.projector_forward(feats, is_video=True)
        assert vlm.encode_vision(pix, True).shape == (T, 16, 64)
        for lo in (0, 1024, 1792):
            ref = vlm.vision_backbone(pix[lo:lo + 256], is_video=True)   # the reference's clip
            got = feats[lo:lo + 256]
            err = (got.float() - ref.float()).norm() / ref.float().norm()
            assert err < 5e-3, (lo, float(err))          # stream-K GEMMs are not bit-reproducible
            # ToMe + MLP on the SAME features: frames are independent, so the clip size cannot matter
            # beyond the GEMM's rounding (matching and merge are exact per frame)
            pv = vlm.projector_forward(got, is_video=True)
            errp = (vis[lo:lo + 256].float() - pv.float()).norm() / pv.float().norm()
            assert errp < 5e-3, (lo, float(errp))


def test_pdrop_token_ops_full_length(K):
    """The four evaluate.py stages (1 -> .8 -> .6 -> .4 -> .2 of 163 840 vision tokens, generic_vlm /
    evaluate.py:170) at full size, integer work checked exactly: keep indices == CPU
    `torch.linspace(dtype=long)`, dropped = exact complement, gather == torch indexing (bit-exact),
    and the rank scores of the last prompt token == a direct softmax on the same rows."""
    NV, D = 10240 * 16, 4480
    hidden = torch.randn(NV + 100, D, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)).bfloat16()
    n = NV
    for r in (0.8, 0.6, 0.4, 0.2):
        keep = int(NV * r)
        idx = K.uniform_keep_indices(n, keep, offset=20)
        ref = torch.linspace(0, n - 1, keep, dtype=torch.long) + 20                 # CPU semantics
        assert torch.equal(idx.cpu(), ref), r
        drop = K.dropped_indices(idx, 20, n)
        assert drop.numel() == n - keep
        both = torch.cat([idx, drop]).sort().values
        assert torch.equal(both, torch.arange(20, 20 + n, device=DEV)), r
        rows = K.gather_rows(hidden, idx)
        assert torch.equal(rows, hidden[idx]), r
        n = keep
    # "attn" ranking: 40 query heads / 8 kv heads x 128 over 163 940 keys
    Hq, Hkv, Dh, Lk = 40, 8, 128, NV + 100
    g = torch.Generator(device=DEV).manual_seed(5)
    q = torch.randn(Hq, Dh, device=DEV, generator=g).bfloat16()
    k = torch.randn(Lk, Hkv, Dh, device=DEV, generator=g).bfloat16()
    sc = K.attn_rank_scores(q, k, Lk, 20, NV)
    assert sc.shape == (NV,)
    logit = torch.einsum("hd,khd->hk", q.float(), k.float().repeat_interleave(Hq // Hkv, 1)).bfloat16()
    logit = (logit.float() / math.sqrt(Dh)).bfloat16().float()
    p = torch.softmax(logit, dim=-1).bfloat16().float().mean(0).bfloat16().float()[20:20 + NV]
    top = lambda t: set(torch.topk(t, 2048).indices.tolist())
    assert len(top(sc.float()) & top(p)) >= 2048 * 0.98            # bf16 ties at the cut may differ
    assert rel(sc, p) < 2e-2
    # The kernel rounds where the reference rounds (bf16 logits after the fp32-accumulated dot, bf16 after "/ sqrt(d)",
    # fp32 softmax -> bf16, fp32 head sum / H -> bf16): the bf16 SCORE VECTORS are equal except where an fp32
    # difference that no two implementations share — exp's last bit, the order of the 163 940-term fp32 sum — meets a
    # bf16 rounding boundary: a handful of scores, each exactly one bf16 step away
    sb, pb = sc.bfloat16(), p.bfloat16()
    assert torch.equal(sb.float(), sc.float()) and torch.equal(pb.float(), p)        # both ARE bf16 values
    steps = (sb.view(torch.int16).int() - pb.view(torch.int16).int()).abs()
    n_diff = int((steps != 0).sum())
    assert int(steps.max()) <= 1 and n_diff <= 2e-3 * NV, (int(steps.max()), n_diff)
    # ... so the keep-sets under the defined tie-break (lower index first) can differ only on those tokens
    order = lambda t, kk: set(torch.sort(t.float(), descending=True, stable=True).indices[:kk].tolist())
    moved = set(torch.nonzero(steps).flatten().tolist())
    for r in (0.8, 0.6, 0.4, 0.2):
        kk = int(NV * r)
        a, b = order(sb, kk), order(pb, kk)
        if n_diff == 0:
            assert a == b
        else:
            tau = torch.sort(pb.float(), descending=True).values[kk - 1].item()
            for i in a ^ b:       # a token kept by one only: its own score moved, or it ties with the threshold a moved score crossed
                assert i in moved or abs(pb[i].float().item() - tau) <= 2.0 ** -8 * abs(tau), (r, i)
    # ... and they differ ONLY inside the bf16 rounding band around the k-th score: every token above it is kept by both,
    # every token below it by neither (tests/keepsets.py), at the four keep counts of the evaluate.py schedule
    from keepsets import assert_keepsets_agree_outside_rounding_band
    for r in (0.8, 0.6, 0.4, 0.2):
        info = assert_keepsets_agree_outside_rounding_band(sc, p, int(NV * r))
        assert info["disagree"] <= info["in_band"]
