import sys, torch, time
sys.path.insert(0, ".")
from timeviper_amd.model import build_synthetic_timeviper
from timeviper_amd.model.llm.nano import NemotronHConfig
cfg = NemotronHConfig(vocab_size=128, hidden_size=4480, intermediate_size=96, num_hidden_layers=1,
                      hybrid_override_pattern="-", num_attention_heads=4, head_dim=16,
                      num_key_value_heads=2, ssm_state_size=16, mamba_num_heads=8,
                      mamba_n_groups=2, mamba_head_dim=8, mamba_chunk_size=16)
vlm = build_synthetic_timeviper(cfg, "siglip-vit-so400m-384px", vit_depth=2)
feats = torch.randn(256, 729, 1152, device="cuda").bfloat16()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
with torch.no_grad():
    print("projector_forward (ToMe 729->16 + MLP), 256 frames: %.2f ms" % timeit(lambda: vlm.projector_forward(feats, is_video=True)))
    print("merge_tokens only: %.2f ms" % timeit(lambda: vlm.projector.merge_tokens(feats, 16, "raw")))
