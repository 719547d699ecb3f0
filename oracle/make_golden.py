"""Generate golden fixtures by RUNNING THE REFERENCE's own Python (build container only).

    python oracle/make_golden.py            # writes tests/golden/*.npz

/root/reference (xiaomi-research/timeviper) is imported with the stub recipe of
SURVEY.md Appendix B: the third-party wheels it needs (mamba_ssm, timm,
flash_attn, causal_conv1d) are absent here, so `is_fast_path_available` is False
and every module runs the reference's own pure-PyTorch path (`torch_forward`,
SDPA).  The only stand-in arithmetic is mamba_ssm's `rmsnorm_fn` (gated RMSNorm),
which the reference does not vendor: that operator stays "parity unpinned".
Nothing under /root/reference is copied; the fixtures hold tensors only
(inputs, weights of toy modules, outputs), all fp32, fixed seeds.
"""
from __future__ import annotations

import contextlib
import importlib
import importlib.machinery
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def import_reference():
    sys.path.insert(0, REF)
    import transformers.utils.import_utils as iu
    iu.is_mamba_2_ssm_available()

    def _stub(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        m.__path__ = []
        sys.modules[name] = m
        return m

    for n in ["mamba_ssm", "mamba_ssm.ops", "mamba_ssm.ops.triton",
              "mamba_ssm.ops.triton.layernorm_gated"]:
        _stub(n)
    from oracle.ops import rmsnorm_gated_ref

    def rmsnorm_fn(x, weight, bias=None, z=None, eps=1e-6, group_size=None,
                   norm_before_gate=True):
        assert bias is None and not norm_before_gate
        return rmsnorm_gated_ref(x, weight, z, eps, group_size).to(x.dtype)

    sys.modules["mamba_ssm.ops.triton.layernorm_gated"].rmsnorm_fn = rmsnorm_fn

    def _pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m

    _pkg("timeviper", f"{REF}/timeviper")
    _pkg("timeviper.model", f"{REF}/timeviper/model")
    _pkg("timeviper.model.llm", f"{REF}/timeviper/model/llm")
    _pkg("timeviper.model.projector", f"{REF}/timeviper/model/projector")
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.default_stream = lambda d=None: None
    nano = importlib.import_module("timeviper.model.llm.llm_repo.nano.modeling_nano")
    return nano


def npz(name, **arrs):
    OUT.mkdir(parents=True, exist_ok=True)
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(f"wrote {name}.npz  ({sum(a.nbytes for a in out.values()) / 1024:.0f} KiB raw)")


def sd_arrays(module, prefix="w."):
    return {prefix + k: v for k, v in module.state_dict().items()}


def tiny_config(nano, **over):
    kw = dict(vocab_size=64, hidden_size=64, intermediate_size=96, num_hidden_layers=8,
              hybrid_override_pattern="M-M*M-*M", num_attention_heads=4, head_dim=16,
              num_key_value_heads=2, ssm_state_size=16, mamba_num_heads=8, mamba_n_groups=1,
              mamba_head_dim=8, mamba_chunk_size=16, rescale_prenorm_residual=False)
    kw.update(over)
    cfg = nano.NemotronHConfig(**kw)
    cfg._attn_implementation = "sdpa"
    return cfg


def randomize(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith("A_log"):
                p.copy_(torch.log(torch.rand(p.shape, generator=g) * 15 + 1))
            elif n.endswith("dt_bias"):
                dt = torch.exp(torch.rand(p.shape, generator=g) * (np.log(0.1) - np.log(1e-3))
                               + np.log(1e-3))
                p.copy_(dt + torch.log(-torch.expm1(-dt)))
            elif n.endswith(".D"):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif "norm" in n and n.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith("alpha"):
                p.copy_(torch.randn(p.shape, generator=g))
            elif p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / np.sqrt(p.shape[-1])) * 2)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)


@torch.no_grad()
def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    nano = import_reference()
    assert nano.is_fast_path_available is False

    # ---- G1/G2: Mamba2 mixer (reference torch_forward) + scan-level tensors ----
    for tag, G, L in [("g1", 1, 45), ("g2_tile", 2, 37), ("g4_tile", 4, 64)]:
        cfg = tiny_config(nano, mamba_n_groups=G, mamba_num_heads=8, mamba_head_dim=8,
                          ssm_state_size=16, mamba_chunk_size=16)
        mixer = nano.NemotronHMamba2Mixer(cfg, layer_idx=0).eval()
        randomize(mixer, 1)
        hidden = torch.randn(2, L, cfg.hidden_size)
        cap = {}
        orig_norm = mixer.norm.forward
        mixer.norm.forward = lambda y, gate=None: (cap.__setitem__("y", y.clone()),
                                                   orig_norm(y, gate))[1]
        cache = nano.HybridMambaAttentionDynamicCache(cfg, 2, dtype=torch.float32)
        out = mixer(hidden, cache_params=cache, cache_position=torch.arange(L))
        # scan inputs, re-derived with the reference module's own submodules (:677-712)
        d_in, H, P, N = mixer.intermediate_size, mixer.num_heads, mixer.head_dim, mixer.ssm_state_size
        proj = mixer.in_proj(hidden)
        gate, xBC, dt = proj.split([d_in, mixer.conv_dim, H], dim=-1)
        xBC_c = mixer.act(mixer.conv1d(xBC.transpose(1, 2))[..., :L].transpose(1, 2))
        x, Bm, Cm = xBC_c.split([d_in, G * N, G * N], dim=-1)
        npz(f"mixer_{tag}", hidden=hidden, out=out, gate=gate, xBC_pre=xBC, xBC_conv=xBC_c,
            scan_x=x.reshape(2, L, H, P), scan_dt=dt, scan_B=Bm.reshape(2, L, G, N),
            scan_C=Cm.reshape(2, L, G, N), scan_y=cap["y"].reshape(2, L, H, P),
            scan_final=cache.ssm_states[0], conv_state=cache.conv_states[0],
            meta=np.array([G, H, P, N, cfg.chunk_size, cfg.conv_kernel]), **sd_arrays(mixer))

    # ---- G5: NemotronHRMSNorm ----
    norm = nano.NemotronHRMSNorm(48, eps=1e-5)
    norm.weight.data = 1 + 0.2 * torch.randn(48)
    xs = torch.randn(3, 7, 48) * 3
    npz("rmsnorm", x=xs, w=norm.weight, y=norm(xs), eps=np.array(1e-5))

    # ---- G6: attention (SDPA causal GQA, no positional encoding) + TransV cross-attention ----
    cfg = tiny_config(nano)
    attn = nano.NemotronHSdpaAttention(cfg, layer_idx=3).eval()
    randomize(attn, 2)
    h = torch.randn(2, 29, cfg.hidden_size)
    o = attn(h)[0]
    npz("attention", hidden=h, out=o, meta=np.array([4, 2, 16]), **sd_arrays(attn))
    ca_mod = importlib.import_module(
        "timeviper.model.llm.llm_repo.nano.merge_modules.cross_attention")
    ca = ca_mod.Qwen2VLSdpaCrossAttention(cfg, layer_idx=3).eval()
    randomize(ca, 3)
    text, drop = torch.randn(1, 9, cfg.hidden_size), torch.randn(1, 21, cfg.hidden_size)
    npz("cross_attention", text=text, dropped=drop, out=ca(text, drop)[0], **sd_arrays(ca))

    # ---- G7: uniform keep indices (the reference's own expression, :1947-1953) ----
    cases, arrs = [], {}
    for n in [16, 100, 4096, 3276, 32768, 26214, 160000, 128000, 163840]:
        for r in [0.8, 0.75, 0.5, 0.2]:
            keep = int(n * r)
            idx = torch.linspace(0, n - 1, keep, dtype=torch.long)
            cases.append((n, keep))
            if n <= 4096:
                arrs[f"idx_{n}_{keep}"] = idx
            else:  # big cases: position-weighted checksum + strided sample
                w = torch.arange(1, keep + 1, dtype=torch.long)
                arrs[f"sum_{n}_{keep}"] = np.array([(idx * w).sum().item() % (2 ** 61 - 1),
                                                    idx.sum().item()])
                arrs[f"smp_{n}_{keep}"] = idx[::997]
    npz("uniform_indices", cases=np.array(cases), **arrs)

    # ---- G9: 8-layer hybrid toy, with and without pdrop + TransV ----
    for tag, pd, merge in [("plain", None, "no_merge"),
                           ("pdrop_nomerge", "uni_2_0.75-attn_3_0.5-attn_6_0.25", "no_merge"),
                           ("pdrop_transv", "uni_2_0.75-attn_3_0.5-attn_6_0.25", "CrossAttention")]:
        cfg = tiny_config(nano, use_pdrop=pd is not None, pdrop_type=pd, merge_module=merge)
        model = nano.NemotronHForCausalLM(cfg).eval()
        randomize(model, 4)
        model.config._attn_implementation = "flash_attention_2"  # mask-free path
        n_vis, t_before, t_after = 40, 5, 9
        Ltot = t_before + n_vis + t_after
        emb = torch.randn(1, Ltot, cfg.hidden_size)
        kw = {}
        lens = [Ltot]
        if pd is not None:
            ptypes = [t.split("_") for t in pd.split("-")]
            model.set_pdrop_args(pdrop_compress_types=[t[0] for t in ptypes],
                                 pdrop_layers=[int(t[1]) for t in ptypes],
                                 pdrop_ratios=[1] + [float(t[2]) for t in ptypes])
            kw["train_pdrop_args"] = {"first_vision_token_positions": torch.tensor([t_before]),
                                      "text_prompt_lens": [t_before + t_after],
                                      "num_vision_tokens": [n_vis], "is_interleaved": False}
        hs = []
        hooks = [blk.register_forward_hook(lambda m, i, o: hs.append(o.clone()))
                 for blk in model.backbone.layers]
        out = model(inputs_embeds=emb, **kw)
        for hk in hooks:
            hk.remove()
        lens = [h_.shape[1] for h_ in hs]
        npz(f"toy_{tag}", embeds=emb, logits=out.logits, layer_lens=np.array(lens),
            hidden_last=hs[-1], hidden_l3=hs[3],
            meta=np.array([t_before, n_vis, t_after]), **sd_arrays(model))

    # ---- G8: fused embedding layout (generic_vlm.py:517-564) ----
    try:
        for n in ["timm", "timm.models", "timm.models.vision_transformer", "torchvision",
                  "torchvision.transforms", "PIL", "PIL.Image"]:
            if n not in sys.modules:
                m = types.ModuleType(n)
                m.__spec__ = importlib.machinery.ModuleSpec(n, None, is_package=True)
                m.__path__ = []
                sys.modules[n] = m
        gv_src = Path(f"{REF}/timeviper/model/generic_vlm.py").read_text()
        # only the method is needed; execute the class body with inert imports
        fake = types.ModuleType("fake_generic_vlm")
        ns = fake.__dict__
        for name in ["timeviper.model.llm", "timeviper.model.vit", "timeviper.utils",
                     "timeviper.utils.overwatch"]:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
        sys.modules["timeviper.model.llm"].GenericLLMBackbone = object
        sys.modules["timeviper.model.vit"].VisionBackbone = object
        for nm in ["MLPProjector", "MultiMLPProjector", "MultiToMe16_mlp_hd64", "ToMe16_mlp_hd64"]:
            setattr(sys.modules["timeviper.model.projector"], nm, object)
        import logging
        sys.modules["timeviper.utils.overwatch"].initialize_overwatch = logging.getLogger
        exec(compile(gv_src, f"{REF}/timeviper/model/generic_vlm.py", "exec"), ns)
        VLM = ns["GenericTimeViperVLM"]
        emb_w = torch.randn(50, 12)
        fake_self = types.SimpleNamespace(
            default_token_id=49,
            llm_backbone=types.SimpleNamespace(embed_input_ids=lambda ids: F.embedding(ids, emb_w)))
        ids = torch.tensor([[3, 7, 1, 49, 49, 49, 5, 6, 2, 8]])
        vis = torch.randn(3, 4, 12)
        fused, _ = VLM.get_fused_data_nopacked(fake_self, vis, ids, None)
        ids2 = torch.tensor([[3, 49, 4, 49, 49, 5]])
        fused2, _ = VLM.get_fused_data_nopacked(fake_self, vis, ids2, None)
        npz("fused_embeddings", emb_w=emb_w, ids=ids, vis=vis, fused=fused, ids2=ids2,
            fused2=fused2, image_token_id=np.array(49))
    except Exception as e:  # pragma: no cover
        print("G8 fused layout fixture skipped:", repr(e))


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") is None:
    main()


@torch.no_grad()
def make_tome_golden():
    """G11: ToMe projector of the reference (timeviper/model/projector/tome.py), fp32."""
    import_reference()
    tome = importlib.import_module("timeviper.model.projector.tome")
    torch.manual_seed(5)
    proj = tome.ToMe16_mlp_hd64(64, 48, num_compressed_tokens=16).eval()
    x = torch.randn(3, 729, 64)
    y = proj(x, compress=True, local_num_frames=1)
    merged = proj.merge_tokens(x, 16, "raw")
    x2 = torch.randn(2, 4 * 100, 64)
    y2 = proj(x2, compress=True, local_num_frames=4)
    npz("tome", x=x, y=y, merged=merged, x2=x2, y2=y2,
        **{"w." + k: v for k, v in proj.state_dict().items()})


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") in (None, "tome"):
    make_tome_golden()


@torch.no_grad()
def make_internvideo2_golden():
    """G10: the reference's InternVideo2 tower (timeviper/model/vit/internvideo2/model.py,
    vit_scale_clean.py) at toy width, fp32.  timm and flash_attn are absent, so the names the
    file imports from them are stubbed (DropPath is inactive in eval, to_2tuple/trunc_normal_
    are torch one-liners) and every block is switched to the file's own `_naive_attn`
    branch — the arithmetic that runs is the reference's."""
    sys.path.insert(0, REF)

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        m.__dict__.update(attrs)
        sys.modules[name] = m

    import transformers.image_processing_utils, transformers.image_transforms  # noqa: F401 (before the timm stub)
    import transformers.image_utils  # noqa: F401
    _mod("timm"); _mod("timm.models")
    _mod("timm.models.layers", DropPath=lambda p=0.0: torch.nn.Identity(),
         to_2tuple=lambda v: v if isinstance(v, tuple) else (v, v),
         trunc_normal_=torch.nn.init.trunc_normal_)
    _mod("flash_attn"); _mod("flash_attn.bert_padding", pad_input=None, unpad_input=None)
    _mod("flash_attn.flash_attn_interface", flash_attn_varlen_qkvpacked_func=None)
    for name, path in [("timeviper", f"{REF}/timeviper"), ("timeviper.model", f"{REF}/timeviper/model"),
                       ("timeviper.model.vit", f"{REF}/timeviper/model/vit"),
                       ("timeviper.model.vit.internvideo2", f"{REF}/timeviper/model/vit/internvideo2")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    iv = importlib.import_module("timeviper.model.vit.internvideo2.model")
    cfg = iv.InternVideo2VisionConfig(num_frames=4, hidden_size=64, num_hidden_layers=5,
                                      num_attention_heads=2, image_size=28, patch_size=14)
    torch.manual_seed(21)
    tower = iv.InternVideo2VisionTower(cfg).eval().float()
    vt = tower.vision_tower
    init_pos = {"pos_embed_init": vt.pos_embed.clone(), "img_pos_embed_init": vt.img_pos_embed.clone()}
    g = torch.Generator().manual_seed(22)
    for name, p in vt.named_parameters():
        if name.endswith(("ls1.weight", "ls2.weight")):
            p.copy_(torch.rand(p.shape, generator=g) + 0.5)
        elif "norm" in name:
            p.copy_(torch.rand(p.shape, generator=g) + 0.5)
        elif name in ("pos_embed", "img_pos_embed"):
            p.add_(torch.randn(p.shape, generator=g) * 0.05)
        elif name.endswith("bias"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        else:
            p.copy_(torch.randn(p.shape, generator=g) * 0.08)
    for blk in vt.blocks:
        blk.attn.use_flash_attn = False
    video = torch.randn(8, 1, 3, 28, 28, generator=g)
    video_b2 = torch.randn(4, 2, 3, 28, 28, generator=g)
    images = torch.randn(2, 1, 3, 28, 28, generator=g)
    npz("internvideo2", video=video, video_out=tower(video, is_video=True),
        video_b2=video_b2, video_b2_out=tower(video_b2, is_video=True),
        images=images, images_out=tower(images, is_video=False),
        depth=np.array(vt.depth), num_heads=np.array(2), **init_pos,
        **{"w." + k: v for k, v in vt.state_dict().items()})


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") in (None, "internvideo2"):
    make_internvideo2_golden()


@torch.no_grad()
def make_qwen2_golden():
    """G12: the reference's Qwen2ForCausalLM (llm_repo/qwen2/modeling_qwen2.py) at toy width, fp32,
    eager attention: plain, pdrop without merge, pdrop + TransV CrossAttention."""
    sys.path.insert(0, REF)
    for name, path in [("timeviper", f"{REF}/timeviper"), ("timeviper.model", f"{REF}/timeviper/model"),
                       ("timeviper.model.llm", f"{REF}/timeviper/model/llm"),
                       ("timeviper.model.llm.llm_repo", f"{REF}/timeviper/model/llm/llm_repo"),
                       ("timeviper.model.llm.llm_repo.qwen2", f"{REF}/timeviper/model/llm/llm_repo/qwen2")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    # The reference targets transformers 4.56; the 5.x in this image no longer registers the
    # "default" rotary initialiser its Qwen2RotaryEmbedding looks up (:354).  Restated from the
    # published 4.56 `_compute_default_rope_parameters`: inv_freq = base^(-2i/d), scaling 1.
    import transformers.modeling_rope_utils as ru

    def default_rope(config, device=None, seq_len=None, **kw):
        d = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        base = getattr(config, "rope_theta", None) or config.rope_parameters["rope_theta"]
        inv = 1.0 / (base ** (torch.arange(0, d, 2, dtype=torch.int64).to(device=device, dtype=torch.float) / d))
        return inv, 1.0
    ru.ROPE_INIT_FUNCTIONS.setdefault("default", default_rope)
    q = importlib.import_module("timeviper.model.llm.llm_repo.qwen2.modeling_qwen2")
    # every parameter is overwritten below: skip the 5.x weight-init pass, which expects rotary
    # modules of its own vintage
    q.Qwen2PreTrainedModel._init_weights = lambda self, module: None
    # 5.x renamed create_causal_mask's `input_embeds` argument
    _ccm = q.create_causal_mask

    def create_causal_mask(**kw):
        if "input_embeds" in kw:
            kw["inputs_embeds"] = kw.pop("input_embeds")
        kw.pop("cache_position", None)            # dropped from the 5.x signature
        return _ccm(**kw)
    q.create_causal_mask = create_causal_mask
    # "uni" stages only: the reference's Qwen2 "attn" ranking cannot run — its mask row is taken
    # from a (1,1,L,L) tensor (:560-569) and broadcasts the scores to 4-D, after which the vision
    # slice (:651-653) is empty and torch.cat (:692) raises.  (The nano file keeps the mask 2-D.)
    pd = "uni_1_0.75-uni_3_0.5-uni_4_0.25"
    # "wc_*": the same three models with well-conditioned projections (q / k weights N(0, 0.08) like the rest) and every
    # parameter rounded to bf16 before the reference runs in fp32: a bf16 implementation then differs from these logits by
    # its activations' rounding only, so the GPU test can hold it to 3e-2 (the sharp toys above amplify any bf16 run to
    # ~ 14 % and only pin the fp32 oracle).
    for tag, kw in [("plain", {}), ("pdrop_nomerge", dict(use_pdrop=True, pdrop_type=pd)),
                    ("pdrop_transv", dict(use_pdrop=True, pdrop_type=pd, merge_module="CrossAttention")),
                    ("wc_plain", {}), ("wc_pdrop_nomerge", dict(use_pdrop=True, pdrop_type=pd)),
                    ("wc_pdrop_transv", dict(use_pdrop=True, pdrop_type=pd, merge_module="CrossAttention"))]:
        wc = tag.startswith("wc_")
        cfg = q.Qwen2Config(vocab_size=64, hidden_size=64, intermediate_size=96, num_hidden_layers=6,
                            num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=512,
                            rope_theta=10000.0, rms_norm_eps=1e-6, pad_token_id=None, **kw)
        cfg._attn_implementation = "eager"
        torch.manual_seed(31)
        model = q.Qwen2ForCausalLM(cfg).eval().float()
        g = torch.Generator().manual_seed(33 if wc else 32)
        for n, p in model.named_parameters():
            if n.endswith("alpha"):
                p.fill_(0.7)
            elif "norm" in n:
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif n.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (0.4 if ("q_proj" in n or "k_proj" in n) and not wc else 0.08))
            if wc:
                p.copy_(p.bfloat16().float())
        ids = torch.randint(0, 64, (1, 43), generator=g)
        args = {}
        if kw:
            model.set_pdrop_args(**model.model.pdrop_args)
            args["train_pdrop_args"] = {"first_vision_token_positions": [3], "num_vision_tokens": [24],
                                        "text_prompt_lens": [19]}
        out = model(input_ids=ids, use_cache=False, **args)
        npz(f"qwen2_{tag}", ids=ids, logits=out.logits,
            **{"w." + k: v for k, v in model.state_dict().items()})


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") in (None, "qwen2"):
    make_qwen2_golden()


@torch.no_grad()
def make_multi_projector_golden():
    """G13: the two-encoder projectors of the reference (projector/tome.py:180-231,
    projector/mlp.py:37-68) on dict inputs keyed by backbone id: an image encoder (frame-wise ToMe)
    beside a 4-frame-tube video encoder (local_num_frames=4) — the reshape + token interleave of
    BASELINE config 4 — and the unequal-token-count case (concatenation)."""
    import_reference()
    tome = importlib.import_module("timeviper.model.projector.tome")
    mlp = importlib.import_module("timeviper.model.projector.mlp")
    torch.manual_seed(9)
    keys = {"dinov2-vit-l": 48, "internvideo2-1b-16-224px": 64}
    proj = tome.MultiToMe16_mlp_hd64(keys, 40, mlp_type="tome_mlp", num_compressed_tokens=16).eval()
    T = 8
    feats = {"dinov2-vit-l": torch.randn(T, 100, 48), "internvideo2-1b-16-224px": torch.randn(T // 4, 4 * 100, 64)}
    y_video = proj(feats, compress=True, local_num_frames={"dinov2-vit-l": 1, "internvideo2-1b-16-224px": 4})
    imgs = {"dinov2-vit-l": torch.randn(3, 100, 48), "internvideo2-1b-16-224px": torch.randn(3, 100, 64)}
    y_image = proj(imgs, compress=True, local_num_frames={"dinov2-vit-l": 1, "internvideo2-1b-16-224px": 1})
    mproj = mlp.MultiMLPProjector({"a": 48, "b": 64}, 40).eval()
    same = {"a": torch.randn(2, 9, 48), "b": torch.randn(2, 9, 64)}
    diff = {"a": torch.randn(2, 9, 48), "b": torch.randn(2, 5, 64)}
    npz("multi_projector", v_dino=feats["dinov2-vit-l"], v_iv2=feats["internvideo2-1b-16-224px"], y_video=y_video,
        i_dino=imgs["dinov2-vit-l"], i_iv2=imgs["internvideo2-1b-16-224px"], y_image=y_image,
        m_same_a=same["a"], m_same_b=same["b"], m_same_y=mproj(same),
        m_diff_a=diff["a"], m_diff_b=diff["b"], m_diff_y=mproj(diff),
        **{"w." + k: v for k, v in proj.state_dict().items()},
        **{"mw." + k: v for k, v in mproj.state_dict().items()})


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") in (None, "multi_projector"):
    make_multi_projector_golden()


@torch.no_grad()
def make_realshape_golden():
    """Two fixtures at the REAL head shapes of Nemotron-Nano-9B-v2 (mamba_head_dim 80, ssm_state_size 128, attention
    head_dim 128; 8 Mamba heads, hidden 128, layers `M*-`): the toy goldens above run the generic scan kernel and the
    small attention instance; these reach ssd_head_kernel<5,4,2>, the d-128 attention kernels and the decode kernels.
      toy_realshape_g2: 2 B/C groups, the reference's prefill over 300 tokens.  (Its CPU prefill maps head h to group
                        h % G — `repeat`, modeling_nano.py:781-782 — which the test mirrors with group_map = "tile".)
      toy_realshape_g1: 1 group (both maps agree), the reference's prefill over 300 + 4 tokens: positions 300 .. 303 are
                        what a prefill of 300 tokens followed by four decode steps must produce.  (The reference's own
                        CPU decode branch cannot be run: `cache_params.ssm_states.device` on a list, :718.)"""
    torch.manual_seed(0)
    torch.set_num_threads(4)
    nano = import_reference()
    L, ndec = 300, 4
    for tag, G in (("g2", 2), ("g1", 1)):
        cfg = tiny_config(nano, vocab_size=96, hidden_size=128, intermediate_size=192, num_hidden_layers=3,
                          hybrid_override_pattern="M*-", num_attention_heads=4, head_dim=128, num_key_value_heads=2,
                          ssm_state_size=128, mamba_num_heads=8, mamba_n_groups=G, mamba_head_dim=80, mamba_chunk_size=64)
        model = nano.NemotronHForCausalLM(cfg).eval()
        randomize(model, 21 + G)
        model.config._attn_implementation = "flash_attention_2"   # mask-free path, SDPA modules stay
        emb = torch.randn(1, L + ndec, cfg.hidden_size)
        n = L if G == 2 else L + ndec
        out = model(inputs_embeds=emb[:, :n])
        npz(f"toy_realshape_{tag}", embeds=emb[:, :n], logits=out.logits, meta=np.array([L, ndec, G]), **sd_arrays(model))


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") in (None, "realshape"):
    make_realshape_golden()
