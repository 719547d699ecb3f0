"""Host logic of the model mirror that needs no GPU: config / state-dict contract,
fused-embedding layout, ToMe projector (plain torch) against reference golden vectors."""
import numpy as np
import pytest
import torch

from conftest import golden_state_dict, load_golden
from timeviper_amd.model.llm.nano import NemotronHConfig, NemotronHForCausalLM
from timeviper_amd.model.projector.tome import ToMe16_mlp_hd64, merge_schedule

PD = "uni_2_0.75-attn_3_0.5-attn_6_0.25"


def toy_config(**kw):
    base = dict(vocab_size=64, hidden_size=64, intermediate_size=96, num_hidden_layers=8,
                hybrid_override_pattern="M-M*M-*M", num_attention_heads=4, head_dim=16,
                num_key_value_heads=2, ssm_state_size=16, mamba_num_heads=8, mamba_n_groups=1,
                mamba_head_dim=8, mamba_chunk_size=16)
    base.update(kw)
    return NemotronHConfig(**base)


@pytest.mark.parametrize("tag,kw", [("plain", {}),
                                    ("pdrop_nomerge", dict(use_pdrop=True, pdrop_type=PD)),
                                    ("pdrop_transv", dict(use_pdrop=True, pdrop_type=PD,
                                                          merge_module="CrossAttention"))])
def test_state_dict_is_reference_compatible(tag, kw):
    g = load_golden(f"toy_{tag}")
    model = NemotronHForCausalLM(toy_config(**kw))
    missing = model.load_state_dict(golden_state_dict(g), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys


def test_nano_9b_config_matches_survey():
    c = NemotronHConfig.nemotron_nano_9b_v2()
    bt = c.layers_block_type
    assert len(bt) == 56 and bt.count("mamba") == 27 and bt.count("mlp") == 25
    assert [i for i, t in enumerate(bt) if t == "attention"] == [14, 21, 30, 39]
    assert (c.hidden_size, c.mamba_num_heads, c.mamba_head_dim, c.ssm_state_size, c.n_groups) == \
        (4480, 128, 80, 128, 8)


def test_embedding_key_rename_hook():
    g = load_golden("toy_plain")
    sd = golden_state_dict(g)
    sd["backbone.embedding.weight"] = sd.pop("backbone.embeddings.weight")
    NemotronHForCausalLM(toy_config()).load_state_dict(sd, strict=True)


def test_tome_schedule():
    assert merge_schedule(729, 16) == [364, 182, 91, 46, 23, 7]
    assert merge_schedule(400, 64) == [200, 100, 36]


def test_tome_projector_matches_reference():
    """module logic (schedule, clip reshape, MLP) with the round operator bound to the oracle"""
    from cpu_kernel_shim import cpu_kernels
    g = load_golden("tome")
    proj = ToMe16_mlp_hd64(64, 48, num_compressed_tokens=16).eval()
    proj.load_state_dict(golden_state_dict(g), strict=True)
    with pytest.raises(Exception):          # no host-side path in the product: CPU tensors raise
        proj.merge_tokens(torch.from_numpy(g["x"]), 16, "raw")
    with torch.no_grad(), cpu_kernels():
        merged = proj.merge_tokens(torch.from_numpy(g["x"]), 16, "raw")
        y = proj(torch.from_numpy(g["x"]), compress=True, local_num_frames=1)
        y2 = proj(torch.from_numpy(g["x2"]), compress=True, local_num_frames=4)
    assert torch.allclose(merged, torch.from_numpy(g["merged"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(y, torch.from_numpy(g["y"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(y2, torch.from_numpy(g["y2"]), rtol=1e-5, atol=1e-6)


def test_fused_embedding_layout_matches_reference():
    from types import SimpleNamespace
    from timeviper_amd.model.generic_vlm import GenericTimeViperVLM
    g = load_golden("fused_embeddings")
    emb_w = torch.from_numpy(g["emb_w"])
    embed = lambda ids: torch.nn.functional.embedding(ids, emb_w)
    fake = SimpleNamespace(default_token_id=int(g["image_token_id"]),
                           llm_backbone=SimpleNamespace(embed_input_ids=embed),
                           dtype_of=lambda e: torch.float32)
    vis = torch.from_numpy(g["vis"])
    f1, _ = GenericTimeViperVLM.get_fused_data_nopacked(fake, vis, torch.from_numpy(g["ids"]))
    f2, _ = GenericTimeViperVLM.get_fused_data_nopacked(fake, vis, torch.from_numpy(g["ids2"]))
    assert torch.equal(f1, torch.from_numpy(g["fused"]))      # contiguous-run fast path
    assert torch.equal(f2, torch.from_numpy(g["fused2"]))     # interleaved text: general walk


def _toy_internvideo2():
    from timeviper_amd.model.vit.internvideo2 import InternVideo2VisionConfig, InternVideo2VisionTower
    cfg = InternVideo2VisionConfig(num_frames=4, hidden_size=64, num_hidden_layers=5,
                                   num_attention_heads=2, image_size=28, patch_size=14)
    return InternVideo2VisionTower(cfg).eval()


def test_internvideo2_sincos_tables_match_reference():
    g = load_golden("internvideo2")
    vt = _toy_internvideo2().vision_tower
    np.testing.assert_allclose(vt.pos_embed.numpy(), g["pos_embed_init"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(vt.img_pos_embed.numpy(), g["img_pos_embed_init"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("case,is_video", [("video", True), ("video_b2", True), ("images", False)])
def test_internvideo2_mirror_matches_reference(case, is_video):
    """Module wiring (clip regrouping, cls/pos tables, fused residual stream) with the
    operators replaced by their CPU restatements; the HIP operators are checked on the GPU."""
    from cpu_kernel_shim import cpu_kernels
    g = load_golden("internvideo2")
    tower = _toy_internvideo2()
    tower.vision_tower.load_state_dict(golden_state_dict(g), strict=True)
    with cpu_kernels():
        out = tower(torch.from_numpy(g[case]), is_video=is_video)
    np.testing.assert_allclose(out.numpy(), g[case + "_out"], rtol=1e-4, atol=3e-5)


@pytest.mark.parametrize("tag,extra", [("plain", {}), ("pdrop_nomerge", dict(use_pdrop=True)),
                                       ("pdrop_transv", dict(use_pdrop=True, merge_module="CrossAttention")),
                                       ("wc_plain", {}), ("wc_pdrop_nomerge", dict(use_pdrop=True)),
                                       ("wc_pdrop_transv", dict(use_pdrop=True, merge_module="CrossAttention"))])
def test_qwen2_mirror_matches_reference(tag, extra):
    """Qwen2 mirror: reference state dict loads strict; forward (operators replaced by their CPU
    restatements) reproduces the reference logits, including pdrop + TransV."""
    from cpu_kernel_shim import cpu_kernels
    from timeviper_amd.model.llm.qwen2 import Qwen2Config, Qwen2ForCausalLM
    g = load_golden(f"qwen2_{tag}")
    pd = "uni_1_0.75-uni_3_0.5-uni_4_0.25"
    cfg = Qwen2Config(vocab_size=64, hidden_size=64, intermediate_size=96, num_hidden_layers=6,
                      num_attention_heads=4, num_key_value_heads=2, rope_theta=10000.0,
                      pdrop_type=pd if extra else None, **extra)
    model = Qwen2ForCausalLM(cfg).eval()
    model.load_state_dict(golden_state_dict(g), strict=True)
    args = {"train_pdrop_args": {"first_vision_token_positions": [3], "num_vision_tokens": [24],
                                 "text_prompt_lens": [19]}} if extra else {}
    with cpu_kernels(), torch.no_grad():
        out = model(input_ids=torch.from_numpy(g["ids"]).long(), use_cache=False, **args)
    np.testing.assert_allclose(out.logits.numpy(), g["logits"], rtol=1e-4, atol=8e-5)


def test_llm_factory_families():
    """llm_registry.py:64-97: both families resolve by id; a config of the wrong family or an
    unknown id is an error, not a silent default."""
    from timeviper_amd.model.llm.llm_factory import GenericLLMBackbone, get_llm_config, llm_family_of
    from timeviper_amd.model.llm.nano import NemotronHConfig
    from timeviper_amd.model.llm.qwen2 import Qwen2Config, Qwen2ForCausalLM
    assert llm_family_of("nanov2-9b") == "nano" and llm_family_of("qwen2.5-7b-instruct") == "qwen2"
    c = get_llm_config("qwen2.5-7b-instruct")
    assert (c.hidden_size, c.num_hidden_layers, c.num_key_value_heads, c.head_dim) == (3584, 28, 4, 128)
    with pytest.raises(ValueError):
        llm_family_of("llama-9000")
    with pytest.raises(ValueError):
        get_llm_config("qwen2.5-3b-instruct")
    small = Qwen2Config(vocab_size=64, hidden_size=32, intermediate_size=48, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1)
    bb = GenericLLMBackbone("qwen2.5-3b-instruct", config=small, use_pdrop=True, pdrop_type="uni_1_0.5")
    assert bb.llm_family == "qwen2" and isinstance(bb.llm, Qwen2ForCausalLM) and bb.embed_dim == 32
    assert bb.llm.backbone.pdrop_layers == [1]
    assert bb.llm.new_cache().get_seq_length() == 0
    with pytest.raises(TypeError):
        GenericLLMBackbone("nanov2-9b", config=small)
    with pytest.raises(TypeError):
        GenericLLMBackbone("qwen2-7b", config=NemotronHConfig(vocab_size=64, hidden_size=32, num_hidden_layers=2,
                                                              hybrid_override_pattern="M*"))


def test_multi_projectors_match_reference():
    """Two-encoder projectors on dict inputs (projector/tome.py:180-231, mlp.py:37-68): frame-wise
    ToMe beside 4-frame-tube ToMe with the reshape + token interleave, and the concatenation case."""
    from timeviper_amd.model.projector import MultiMLPProjector, MultiToMe16_mlp_hd64
    g = load_golden("multi_projector")
    t = lambda k: torch.from_numpy(g[k])
    keys = {"dinov2-vit-l": 48, "internvideo2-1b-16-224px": 64}
    from cpu_kernel_shim import cpu_kernels
    proj = MultiToMe16_mlp_hd64(keys, 40, mlp_type="tome_mlp", num_compressed_tokens=16).eval()
    proj.load_state_dict(golden_state_dict(g), strict=True)
    with torch.no_grad(), cpu_kernels():
        yv = proj({"dinov2-vit-l": t("v_dino"), "internvideo2-1b-16-224px": t("v_iv2")}, compress=True,
                  local_num_frames={"dinov2-vit-l": 1, "internvideo2-1b-16-224px": 4})
        yi = proj({"dinov2-vit-l": t("i_dino"), "internvideo2-1b-16-224px": t("i_iv2")}, compress=True,
                  local_num_frames={"dinov2-vit-l": 1, "internvideo2-1b-16-224px": 1})
    assert yv.shape == g["y_video"].shape == (8, 32, 40)
    assert torch.allclose(yv, t("y_video"), rtol=1e-5, atol=1e-6)
    assert torch.allclose(yi, t("y_image"), rtol=1e-5, atol=1e-6)
    mp = MultiMLPProjector({"a": 48, "b": 64}, 40).eval()
    mp.load_state_dict(golden_state_dict(g, prefix="mw."), strict=True)
    with torch.no_grad():
        assert torch.allclose(mp({"a": t("m_same_a"), "b": t("m_same_b")}), t("m_same_y"), rtol=1e-5, atol=1e-6)
        assert torch.allclose(mp({"a": t("m_diff_a"), "b": t("m_diff_b")}), t("m_diff_y"), rtol=1e-5, atol=1e-6)


def test_multivit_registry_and_contract():
    """registry.py:74-99 ('+' ids, named variant) and the attributes generic_vlm.py:180-186,:415
    read from a multi-encoder backbone."""
    from timeviper_amd.model.vit import MultiViTBackbone, get_vision_backbone_config
    from timeviper_amd.model.vit.internvideo2 import InternVideo2VisionConfig
    c = get_vision_backbone_config("dinov2-vit-l+internvideo2-1b-16-224px")
    assert c["type"] == "multi" and c["default_image_size"] == 224
    assert c["backbones"] == ["dinov2-vit-l", "internvideo2-1b-16-224px"]
    c = get_vision_backbone_config("dinosiglip-vit-so-384px")
    assert c["backbones"] == ["dinov2-vit-l", "siglip-vit-so400m-384px"] and c["default_image_size"] == 384
    with pytest.raises(ValueError):
        get_vision_backbone_config("dinov2-vit-l+nope")
    vcfg = InternVideo2VisionConfig(num_frames=4, hidden_size=64, num_hidden_layers=3,
                                    num_attention_heads=2, image_size=28, patch_size=14)
    with torch.device("meta"):
        vb = MultiViTBackbone("dinov2-vit-l+internvideo2-1b-16-224px", member_kwargs={
            "dinov2-vit-l": dict(depth_override=2, default_image_size=28),
            "internvideo2-1b-16-224px": dict(default_image_size=28, vision_config=vcfg)})
    assert vb.get_identifier == "multivit"
    assert vb.backbone_ids == ["dinov2-vit-l", "internvideo2-1b-16-224px"]
    assert list(vb.backbones.keys()) == ["dinov2_vit_l", "internvideo2_1b_16_224px"]
    assert vb.backbones["dinov2_vit_l"].embed_dim == 1024 and vb.backbones["internvideo2_1b_16_224px"].embed_dim == 64


def test_evaluate_py_call_sequence_on_a_saved_checkpoint(tmp_path):
    """The outer drop-in surface as `evaluate.py` drives it (reference evaluate.py:183-214 build_model,
    :507-530 generate -> extract_answer): factories -> `HybridTimeViperVLM.from_pretrained(checkpoint,
    ...)` with `strict=True` loading -> `model.generate(input_ids, pixel_values=..., pixel_values_videos=...,
    attention_mask=..., max_new_tokens=..., use_cache=True, do_sample=False, temperature=0,
    answer_prompt=...)` returning the decoded, prompt-stripped TEXT (generic_vlm.py:743-760).  Kernels
    are the oracle-backed shims (no GPU here): this is the host logic above the C ABI."""
    from cpu_kernel_shim import cpu_kernels
    from timeviper_amd.model import (HybridTimeViperVLM, get_llm_backbone_and_tokenizer)
    from timeviper_amd.model.llm import NemotronHConfig
    from timeviper_amd.model.vit import TimmViTBackbone

    def backbones():
        cfg = NemotronHConfig(vocab_size=128, hidden_size=64, intermediate_size=96, num_hidden_layers=8,
                              hybrid_override_pattern="M-M*M-*M", num_attention_heads=4, head_dim=16,
                              num_key_value_heads=2, ssm_state_size=16, mamba_num_heads=8,
                              mamba_n_groups=2, mamba_head_dim=8, mamba_chunk_size=16)
        vb = TimmViTBackbone("siglip-vit-b16-224px", depth_override=2, default_image_size=96)
        llm, tokenizer = get_llm_backbone_and_tokenizer(
            "nanov2-9b", llm_max_length=None, attn_implementation="flash_attention_2",
            merge_module="CrossAttention", use_pdrop=True, pdrop_type="uni_2_0.75-attn_3_0.5", config=cfg)
        return vb, llm, tokenizer

    # a "pretrained checkpoint": the state dict of a randomly initialised model, as torch.save writes it
    torch.manual_seed(3)
    vb, llm, _ = backbones()
    src = HybridTimeViperVLM("src", vb, llm, arch_specifier="tome_mlp-16")
    ckpt = tmp_path / "timeviper.pt"
    torch.save(src.state_dict(), ckpt)

    torch.manual_seed(4)                       # a different init: only the checkpoint can make them agree
    vb, llm, tokenizer = backbones()
    model = HybridTimeViperVLM.from_pretrained(pretrained_checkpoint=ckpt, model_id="cobra-siglip+3b",
                                               vision_backbone=vb, llm_backbone=llm,
                                               arch_specifier="tome_mlp-16", visual_token_order="raw")
    assert not model.training and all(not p.requires_grad for p in model.parameters())
    assert next(model.parameters()).dtype == torch.bfloat16
    for (k, a), (_, b) in zip(sorted(src.state_dict().items()), sorted(model.state_dict().items())):
        assert torch.equal(a.to(torch.bfloat16), b), k
    # a checkpoint with a missing / unexpected key must be refused (strict=True, :896-899)
    bad = dict(src.state_dict())
    bad.pop(next(iter(bad)))
    torch.save(bad, tmp_path / "bad.pt")
    vb2, llm2, _ = backbones()
    with pytest.raises(RuntimeError):
        HybridTimeViperVLM.from_pretrained(tmp_path / "bad.pt", "m", vb2, llm2, arch_specifier="tome_mlp-16")

    # attributes evaluate.py / vllm_infer.py read
    assert model.llm_tokenizer is tokenizer and model.config is model.llm_backbone.llm.config
    assert model.arch_specifier == "tome_mlp-16" and model.llm_backbone.half_precision_dtype == torch.bfloat16
    T, tok = 4, model.default_token_id
    batch_itm = {"input_ids": torch.tensor([[5, 6] + [tok] * T + [7, 8, 9]]),
                 "pixel_values": None,
                 "pixel_values_videos": torch.randn(T, 3, 96, 96).to(torch.bfloat16),
                 "attention_mask": torch.ones(1, T + 5, dtype=torch.long)}
    with cpu_kernels():
        kw = dict(pixel_values=batch_itm["pixel_values"], pixel_values_videos=batch_itm["pixel_values_videos"],
                  attention_mask=batch_itm["attention_mask"], max_new_tokens=5, use_cache=True,
                  do_sample=False, temperature=0)
        text = model.generate(batch_itm["input_ids"], answer_prompt="Best Options: (", **kw)
        ids = model.generate(batch_itm["input_ids"], answer_prompt="Best Options: (", return_ids=True, **kw)
        text_plain = model.generate(batch_itm["input_ids"], answer_prompt=None, **kw)
        # the prefill behind it: prompt + tokenised answer prompt, greedy first token
        ap = tokenizer("Best Options: (", add_special_tokens=False, return_tensors="pt").input_ids
        out = model(input_ids=torch.cat([batch_itm["input_ids"], ap], dim=1),
                    pixel_values_videos=batch_itm["pixel_values_videos"])
    assert isinstance(text, str) and isinstance(text_plain, str)
    assert text == tokenizer.decode(ids[0], skip_special_tokens=False).strip()
    assert 1 <= ids.shape[1] <= 5 and int(ids[0, 0]) == int(out.logits[0, -1].argmax())
    assert f"<{tok}>" not in text                      # the prompt (and its <image> run) is stripped
    # padded batches are refused loudly rather than mis-decoded
    with cpu_kernels(), pytest.raises(NotImplementedError):
        model.generate(batch_itm["input_ids"], pixel_values_videos=batch_itm["pixel_values_videos"],
                       attention_mask=torch.tensor([[0] + [1] * (T + 4)]), max_new_tokens=1)
