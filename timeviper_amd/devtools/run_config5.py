"""BASELINE config 4 at its named size on one GPU (dev run, not the bench line): Qwen2.5-7B geometry
+ DINOv2-L / InternVideo2-1B dual encoder, 224 px, 32 tokens per frame.
usage: python timeviper_amd/devtools/run_config4.py [frames=4096] [steps=2]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from timeviper_amd.model import build_synthetic_timeviper  # noqa: E402
from timeviper_amd.model.llm.qwen2 import Qwen2Config  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
pd = "uni_7_0.8-uni_14_0.6-uni_21_0.4"
vlm = build_synthetic_timeviper(Qwen2Config.qwen2_5_7b(), "dinov2-vit-l+internvideo2-1b-16-224px",
                                pdrop_type=pd, merge_module="CrossAttention", device=dev,
                                llm_backbone_id="qwen2.5-7b-instruct")
g = torch.Generator(device=dev).manual_seed(1)
tok = vlm.default_token_id
ids = torch.cat([torch.randint(3, 1000, (20,), device=dev, generator=g), torch.full((T,), tok, device=dev),
                 torch.randint(3, 1000, (80,), device=dev, generator=g)])[None]
pix = torch.randn(T, 3, 224, 224, device=dev, dtype=torch.bfloat16, generator=g)
with torch.inference_mode():
    out = vlm(input_ids=ids, pixel_values_videos=pix).logits
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = vlm(input_ids=ids, pixel_values_videos=pix).logits
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    vis = vlm.encode_vision(pix, True)
    ev[1].record()
    fused, _ = vlm.get_fused_data_nopacked(vis, ids)
    vlm.llm_backbone(inputs_embeds=fused, train_pdrop_args={**vlm.pdrop_bookkeeping(ids, vis)}, logits_to_keep=1)
    ev[2].record()
    torch.cuda.synchronize()
assert torch.isfinite(out.float()).all()
print(f"config4: {T} frames, {T * 32 + 100} tokens, {dt * 1e3:.0f} ms/forward = {T / dt:.0f} frames/s; "
      f"vision {ev[0].elapsed_time(ev[1]):.0f} ms, LM {ev[1].elapsed_time(ev[2]):.0f} ms; "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
