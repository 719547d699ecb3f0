"""BASELINE config 5 at its named sizes: "TimeViper-Qwen2.5 backbone, DINOv2+InternVideo2 dual encoder,
4096 frames, fp8 MFMA attention path" = 4 096 frames x 32 tokens + 100 text tokens = 131 172 tokens of a
28 q / 4 kv x 128 causal RoPE attention, and a 4 096-frame dual-encoder vision path.

The oracle cannot run these sizes, so (like tests/test_fullsize_gpu.py) the checks are the definition on
sampled rows and size-independent properties: RoPE on sampled positions, softmax attention of sampled
query rows over all visible keys in fp32, the dual encoder against its members run one by one, the
interleave rule of projector/tome.py:214-231."""
import math

import pytest
import torch

from oracle import qwen2 as oq

pytestmark = pytest.mark.gpu
DEV = "cuda"
FRAMES, TOK_PER_FRAME = 4096, 32
L5 = FRAMES * TOK_PER_FRAME + 100          # 131 172
HQ, HKV, D = 28, 4, 128                    # Qwen2.5-7B attention geometry (llm_registry.py:74-75)
THETA = 1e6


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def test_config5_rope_causal_attention_131k_tokens_fp8_and_bf16():
    """modeling_qwen2.py:196-244 at full length: rotary embedding of q and k in place (tv_rope_fwd),
    then causal GQA attention — the bf16 kernel and the fp8 MFMA variant — against the definition on
    sampled rows.  Tolerances: bf16 2e-2; fp8 7e-2 (the e4m3 error model of test_attention_fp8_gpu.py,
    relative to rows of random data; a row's error is dominated by the 3.6 % rms of P and of V)."""
    from timeviper_amd import kernels as K
    from timeviper_amd.model.llm.qwen2 import Qwen2Config, Qwen2RotaryEmbedding
    g = torch.Generator(device=DEV).manual_seed(5)
    q = torch.randn(1, L5, HQ, D, device=DEV, generator=g).bfloat16()
    k = torch.randn(1, L5, HKV, D, device=DEV, generator=g).bfloat16()
    v = torch.randn(1, L5, HKV, D, device=DEV, generator=g).bfloat16()
    q0, k0 = q.clone(), k.clone()
    rot = Qwen2RotaryEmbedding(Qwen2Config.qwen2_5_7b())
    pos = torch.arange(L5, device=DEV)[None]
    cos, sin = rot(q, pos)
    K.apply_rotary_pos_emb_(q, k, cos, sin)
    rows = (0, 1, 63, 64, 4097, 77777, L5 - 1)
    # RoPE itself on the sampled positions: x cos + rotate_half(x) sin with the oracle's tables
    for i in rows:
        c, s_ = oq.rotary_tables(torch.tensor([[i]]), D, THETA, torch.bfloat16)
        c, s_ = c[0, 0].float().to(DEV), s_[0, 0].float().to(DEV)
        for x0, x1 in ((q0, q), (k0, k)):
            ref = x0[0, i].float() * c + oq.rotate_half(x0[0, i].float()) * s_
            assert rel(x1[0, i].float(), ref) < 1e-2, i
    o16 = K.flash_attn_func(q, k, v, causal=True)
    o8 = K.flash_attn_fp8_func(q, k, v, causal=True)
    assert torch.isfinite(o16.float()).all() and torch.isfinite(o8.float()).all()
    rep = HQ // HKV
    for i in rows:
        qi = q[0, i].float().view(HKV, rep, D)
        sc = torch.einsum("ghd,lgd->ghl", qi, k[0, :i + 1].float()) / math.sqrt(D)
        ref = torch.einsum("ghl,lgd->ghd", torch.softmax(sc, -1), v[0, :i + 1].float()).reshape(HQ, D)
        assert rel(o16[0, i].float(), ref) < 2e-2, i
        assert rel(o8[0, i].float(), ref) < 7e-2, i
    assert rel(o8.float(), o16.float()) < 7e-2
    # shard property (bottom-right aligned causal mask): the last third's queries against all keys;
    # the bf16 kernel is bit-identical, the fp8 one re-derives q's per-head scale from the slice
    s0 = 2 * L5 // 3 + 5
    assert torch.equal(K.flash_attn_func(q[:, s0:], k, v, causal=True), o16[:, s0:])
    assert rel(K.flash_attn_fp8_func(q[:, s0:], k, v, causal=True).float(), o16[:, s0:].float()) < 7e-2


def test_config5_dual_encoder_4096_frames_equals_members_run_alone():
    """DINOv2-L (frame-wise, 256 patches) + InternVideo2-1B (4-frame tubes, 39 blocks) at full widths
    over 4 096 frames of 224 px: the dual encoder + MultiToMe projector must equal its members run one
    by one (clip by clip, as generic_vlm.py:274-281 would call them), interleaved token-wise
    (projector/tome.py:214-231: 16 + 16 tokens per frame).  A small Qwen2 stack stands behind it
    (the LM at full size is covered by the attention test above and by devtools/run_config5.py)."""
    from timeviper_amd import kernels as K
    from timeviper_amd.model import build_synthetic_timeviper
    from timeviper_amd.model.llm.qwen2 import Qwen2Config
    cfg = Qwen2Config(vocab_size=128, hidden_size=256, intermediate_size=512, num_hidden_layers=4,
                      num_attention_heads=2, num_key_value_heads=1, rope_theta=THETA)
    bid_d, bid_v = "dinov2-vit-l", "internvideo2-1b-16-224px"
    vlm = build_synthetic_timeviper(cfg, f"{bid_d}+{bid_v}", pdrop_type="uni_1_0.75-uni_2_0.5",
                                    merge_module="CrossAttention", llm_backbone_id="qwen2.5-7b-instruct")
    T = FRAMES
    g = torch.Generator(device=DEV).manual_seed(7)
    pix = torch.randn(T, 3, 224, 224, device=DEV, dtype=torch.bfloat16, generator=g)
    with torch.no_grad():
        vis = vlm.encode_vision(pix, is_video=True)
        assert vis.shape == (T, TOK_PER_FRAME, 256) and torch.isfinite(vis.float()).all()
        vb, pj = vlm.vision_backbone, vlm.projector.projectors
        md, mv = vb.backbones[bid_d.replace("-", "_")], vb.backbones[bid_v.replace("-", "_")]
        for lo in (0, 1792, T - 256):                     # three of the sixteen 256-frame clips
            clip = pix[lo:lo + 256]
            feats = vb(clip, is_video=True, clip_frames=256)
            # the members alone on the same clip: same kernels, same shapes -> the same bits
            assert torch.equal(feats[bid_d], md(clip)) and feats[bid_d].shape == (256, 256, 1024)
            assert torch.equal(feats[bid_v], mv(clip.unsqueeze(1), is_video=True)) and feats[bid_v].shape == (64, 1024, 1408)
            # projector: frame-wise ToMe beside 4-frame-tube ToMe, interleaved token-wise
            vis_clip = vlm.projector_forward(feats, is_video=True)
            a = pj[bid_d](feats[bid_d], compress=True, local_num_frames=1)
            b = pj[bid_v](feats[bid_v], compress=True, local_num_frames=4)
            assert torch.equal(vis_clip[:, 0::2], a) and torch.equal(vis_clip[:, 1::2], b.reshape(256, 16, -1))
        # eight clips per launch (what encode_vision feeds the towers) against the clips on their own, at the
        # FEATURE level: the GEMM library picks kernels by shape, so the rows agree to bf16 noise, not bit for
        # bit (after the projector the comparison would be dominated by ToMe's discrete matching, where a
        # last-bit difference in a similarity merges another pair of tokens)
        fused = vb(pix[:2048], is_video=True, clip_frames=256)
        for cidx in (0, 7):
            one = vb(pix[256 * cidx:256 * cidx + 256], is_video=True, clip_frames=256)
            assert rel(fused[bid_d][256 * cidx:256 * cidx + 256].float(), one[bid_d].float()) < 2e-2
            assert rel(fused[bid_v][64 * cidx:64 * cidx + 64].float(), one[bid_v].float()) < 2e-2
        del fused, one
        tok = vlm.default_token_id
        ids = torch.cat([torch.arange(3, 23, device=DEV), torch.full((T,), tok, device=DEV),
                         torch.arange(30, 110, device=DEV)])[None]
        out16 = vlm(input_ids=ids, visual_embeddings=vis).logits
        with K.fp8_attention():
            out8 = vlm(input_ids=ids, visual_embeddings=vis).logits
    assert out16.shape == (1, 1, 128) and torch.isfinite(out16).all() and torch.isfinite(out8).all()
    # (random-init weights: a smoke bound on the toy stack — fp8 against bf16, two noisy paths; the bound against the
    # fp32 oracle on well-conditioned weights is tests/test_model_gpu.py::test_qwen2_bf16_vs_oracle_well_conditioned)
    assert rel(out8.float(), out16.float()) < 0.15
