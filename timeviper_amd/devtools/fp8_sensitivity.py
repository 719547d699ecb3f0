"""Dev tool: how much of the fp8-vs-bf16 difference of the config-5 logits is the model's own sensitivity
(random-init weights, 28 layers) and how much each quantisation step contributes.
usage: python timeviper_amd/devtools/fp8_sensitivity.py [frames=1024]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from timeviper_amd import kernels as K  # noqa: E402
from timeviper_amd.model import build_synthetic_timeviper  # noqa: E402
from timeviper_amd.model.llm.qwen2 import Qwen2Config  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
vlm = build_synthetic_timeviper(Qwen2Config.qwen2_5_7b(), "dinov2-vit-l+internvideo2-1b-16-224px",
                                pdrop_type="uni_7_0.8-uni_14_0.6-uni_21_0.4", merge_module="CrossAttention",
                                device=dev, llm_backbone_id="qwen2.5-7b-instruct")
g = torch.Generator(device=dev).manual_seed(1)
tok = vlm.default_token_id
ids = torch.cat([torch.randint(3, 1000, (20,), device=dev, generator=g), torch.full((T,), tok, device=dev),
                 torch.randint(3, 1000, (80,), device=dev, generator=g)])[None]
pix = torch.randn(T, 3, 224, 224, device=dev, dtype=torch.bfloat16, generator=g)
rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()


def qdq(x):        # e4m3 quantise / dequantise with one scale per (batch, head), as the fp8 pre-pass does
    amax = x.float().abs().amax(dim=(1, 3), keepdim=True).clamp_min(1e-30)
    s = 440.0 / amax
    return ((x.float() * s).to(torch.float8_e4m3fn).float() / s).to(x.dtype)


orig = K.flash_attn_func
with torch.inference_mode():
    vis = vlm.encode_vision(pix, True)
    run = lambda: vlm(input_ids=ids, visual_embeddings=vis).logits
    a = run()
    a2 = run()
    with K.fp8_attention(min_keys=4096):
        b = run()

    def qdq_attn(q, k, v, *args, **kw):
        if k.shape[1] >= 4096:
            q, k, v = qdq(q), qdq(k), qdq(v)
        return orig(q, k, v, *args, **kw)
    K.flash_attn_func = qdq_attn
    try:
        c = run()
    finally:
        K.flash_attn_func = orig

    def noisy_attn(q, k, v, *args, **kw):       # a 2^-9 relative perturbation of the attention OUTPUT (one bf16 ulp)
        o = orig(q, k, v, *args, **kw)
        return o * (1 + (torch.rand_like(o, dtype=torch.float32) - 0.5) * 2 ** -7).to(o.dtype) if k.shape[1] >= 4096 else o
    K.flash_attn_func = noisy_attn
    try:
        d = run()
    finally:
        K.flash_attn_func = orig
print(f"frames {T}: bf16 run-to-run {rel(a2, a):.3e}; fp8 kernel vs bf16 {rel(b, a):.3e}; bf16 kernel on e4m3-rounded q/k/v "
      f"vs bf16 {rel(c, a):.3e}; bf16 kernel with +-0.4 % output noise vs bf16 {rel(d, a):.3e}; "
      f"argmax equal: fp8 {bool(b.argmax() == a.argmax())}, qdq {bool(c.argmax() == a.argmax())}, noise {bool(d.argmax() == a.argmax())}")
