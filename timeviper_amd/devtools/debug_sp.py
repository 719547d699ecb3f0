"""dev: 2 ranks on one GPU, per-layer comparison of the sharded hidden states."""
import os, sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from pathlib import Path
_ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(_ROOT)); sys.path.insert(0, str(_ROOT / "tests"))
from test_distributed_cpu import free_port
from test_distributed_gpu import stage_collectives_through_host


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    stage_collectives_through_host()
    from timeviper_amd.distributed import SequenceParallelTimeViper, all_gather_varlen
    from timeviper_amd.model import build_synthetic_timeviper
    from timeviper_amd.model.llm.nano import NemotronHConfig
    cfg = NemotronHConfig(vocab_size=128, hidden_size=256, intermediate_size=384, num_hidden_layers=8,
                          hybrid_override_pattern="M-M*M-*M", num_attention_heads=4, head_dim=64,
                          num_key_value_heads=2, ssm_state_size=128, mamba_num_heads=8,
                          mamba_n_groups=2, mamba_head_dim=40, mamba_chunk_size=64)
    vlm = build_synthetic_timeviper(cfg, "siglip-vit-b16-224px", pdrop_type=None, merge_module="no_merge",
                                    vit_depth=3, image_size=96, seed=3)
    T = 21
    tok = vlm.default_token_id
    g = torch.Generator().manual_seed(1)
    ids = torch.tensor([[5, 6, 7] + [tok] * T + [8, 9, 10, 11, 12]], device="cuda")
    pix = torch.randn(T, 3, 96, 96, generator=g).cuda().bfloat16()
    with torch.no_grad():
        r = SequenceParallelTimeViper(vlm, rank, world)
        lo, hi = r.frame_range(T)
        vis = vlm.encode_vision(pix[lo:hi], is_video=True)
        vis_full = vlm.encode_vision(pix, is_video=True)
        print(rank, "vision shard vs full slice", (vis.float() - vis_full[lo:hi].float()).abs().max().item(), flush=True)
        fused, _ = vlm.get_fused_data_nopacked(vis_full, ids)
        bb = r.bb
        # reference per-layer stream
        ref_h = [fused]
        x = fused
        for blk in bb.layers:
            x = blk(x) if not isinstance(blk(x), tuple) else blk(x)[0]
            ref_h.append(x)
        tpf = vis.shape[1]
        bounds, nb, na = r.shard_layout(ids, T, tpf)
        s, e = bounds[rank]
        embed = vlm.llm_backbone.embed_input_ids
        parts = []
        if rank == 0 and nb:
            parts.append(embed(ids[:, :nb]))
        parts.append(vis.reshape(1, -1, vis.shape[-1]).to(fused.dtype))
        if rank == world - 1 and na:
            parts.append(embed(ids[:, ids.shape[1] - na:]))
        hidden = torch.cat(parts, 1)
        print(rank, "embed", (hidden.float() - fused[:, s:e].float()).abs().max().item(), flush=True)
        for i, block in enumerate(bb.layers):
            normed = block.norm(hidden)
            if block.block_type == "mamba":
                d = r._mamba(block.mixer, normed)
            elif block.block_type == "attention":
                d = r._attention(block.mixer, normed)
            else:
                d = block.mixer(normed)
            hidden = hidden + d
            ref = ref_h[i + 1][:, s:e]
            lt = ((hidden[:, -1].float() - ref[:, -1].float()).norm() / ref[:, -1].float().norm()).item()
            print(rank, "layer", i, block.block_type, "rel", ((hidden.float() - ref.float()).norm() / ref.float().norm()).item(), "last-token rel", lt, "last-token norm", ref[:, -1].float().norm().item(), flush=True)
        if rank == world - 1:
            la = r.llm.lm_head(bb.norm_f(hidden)[:, -1:]).float()
            lb = r.llm.lm_head(bb.norm_f(ref_h[-1])[:, -1:]).float()
            print("logits rel", ((la - lb).norm() / lb.norm()).item(), "logits norm", lb.norm().item(), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
