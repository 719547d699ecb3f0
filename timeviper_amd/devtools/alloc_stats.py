"""Dev tool: does the steady-state forward still call hipMalloc / hipFree (caching-allocator misses)?"""
import sys
import time

import torch

sys.path.insert(0, ".")
from timeviper_amd.model import build_synthetic_timeviper  # noqa: E402
from timeviper_amd.model.llm.nano import NemotronHConfig  # noqa: E402

PD = "uni_14_0.8-attn_21_0.6-attn_30_0.4-attn_39_0.2"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
dev = torch.device("cuda", 0)
vlm = build_synthetic_timeviper(NemotronHConfig.nemotron_nano_9b_v2(), "siglip-vit-so400m-384px", pdrop_type=PD,
                                merge_module="CrossAttention", device=dev)
tok = vlm.default_token_id
ids = torch.cat([torch.randint(3, 1000, (20,), device=dev), torch.full((T,), tok, device=dev),
                 torch.randint(3, 1000, (80,), device=dev)])[None]
pix = torch.randn(T, 3, 384, 384, device=dev, dtype=torch.bfloat16)
keys = ["num_device_alloc", "num_device_free", "num_alloc_retries", "reserved_bytes.all.peak", "allocated_bytes.all.peak"]
with torch.inference_mode():
    vlm(input_ids=ids, pixel_values_videos=pix)
    torch.cuda.synchronize()
    for it in range(3):
        s0 = torch.cuda.memory_stats()
        t0 = time.perf_counter()
        vlm(input_ids=ids, pixel_values_videos=pix)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s1 = torch.cuda.memory_stats()
        print(f"step {it}: {dt * 1e3:.0f} ms  " + "  ".join(
            f"{k}={s1.get(k, 0) - s0.get(k, 0) if 'peak' not in k else round(s1.get(k, 0) / 2**30, 1)}" for k in keys))
