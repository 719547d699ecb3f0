"""Sequence-parallel runner with the REAL HIP kernels: two ranks share the one GPU of the test
box (RCCL refuses two ranks on one device, so the collectives are staged through gloo on the
host); the sharded forward must reproduce the single-process forward on the same GPU."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_distributed_cpu import PD, free_port

pytestmark = pytest.mark.gpu


def stage_collectives_through_host():
    """gloo has no CUDA all_gather: run every collective on CPU copies."""
    ag, bc = dist.all_gather, dist.broadcast

    def all_gather(out, t, group=None, **kw):
        if not t.is_cuda:
            return ag(out, t, group=group, **kw)
        host = [torch.empty(o.shape, dtype=o.dtype) for o in out]
        ag(host, t.cpu(), group=group)
        for o, h in zip(out, host):
            o.copy_(h)

    def broadcast(t, src=0, group=None, **kw):
        if not t.is_cuda:
            return bc(t, src=src, group=group, **kw)
        h = t.cpu()
        bc(h, src=src, group=group)
        t.copy_(h)

    agt = dist.all_gather_into_tensor

    class _Done:
        def wait(self):
            return True

    def all_gather_into_tensor(out, t, group=None, async_op=False):
        if not t.is_cuda:
            return agt(out, t, group=group, async_op=async_op)
        host = torch.empty(out.shape, dtype=out.dtype)
        agt(host, t.cpu(), group=group)
        out.copy_(host)
        return _Done() if async_op else None

    dist.all_gather, dist.broadcast, dist.all_gather_into_tensor = all_gather, broadcast, all_gather_into_tensor


def worker(rank, world, port, merge, T, q, pd=PD, family="nano"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stage_collectives_through_host()
        if os.environ.get("TV_SSD_IMPL"):
            from timeviper_amd import kernels as K
            K.ssd_scan_set_impl(int(os.environ["TV_SSD_IMPL"]))
        from timeviper_amd.distributed import SequenceParallelTimeViper
        from timeviper_amd.model import build_synthetic_timeviper
        from timeviper_amd.model.llm.nano import NemotronHConfig
        cfg = NemotronHConfig(vocab_size=128, hidden_size=256, intermediate_size=384, num_hidden_layers=8,
                              hybrid_override_pattern="M-M*M-*M", num_attention_heads=4, head_dim=64,
                              num_key_value_heads=2, ssm_state_size=128, mamba_num_heads=8,
                              mamba_n_groups=2, mamba_head_dim=40, mamba_chunk_size=64)
        kw = {}
        if family == "qwen2":
            from timeviper_amd.model.llm.qwen2 import Qwen2Config
            cfg = Qwen2Config(vocab_size=128, hidden_size=256, intermediate_size=512, num_hidden_layers=6,
                              num_attention_heads=4, num_key_value_heads=2, rope_theta=10000.0)
            kw = {"llm_backbone_id": "qwen2.5-7b-instruct"}
        vlm = build_synthetic_timeviper(cfg, "siglip-vit-b16-224px", pdrop_type=pd, merge_module=merge,
                                        vit_depth=3, image_size=96, seed=3, **kw)
        tok = vlm.default_token_id
        g = torch.Generator().manual_seed(1)
        ids = torch.tensor([[5, 6, 7] + [tok] * T + [8, 9, 10, 11, 12]], device="cuda")
        pix = torch.randn(T, 3, 96, 96, generator=g).cuda().bfloat16()
        with torch.no_grad():
            runner = SequenceParallelTimeViper(vlm, rank, world)
            lo, hi = runner.frame_range(T)
            logits = runner.forward(ids, pix[lo:hi], T)
            trace = [t.cpu().numpy() for t in runner.trace]
            if rank == 0:
                ref = vlm(input_ids=ids, pixel_values_videos=pix).logits
                ref_trace = [t["kept"].cpu().numpy() for t in (vlm.llm_backbone.llm.backbone.last_pdrop_trace or [])]
                q.put((logits.float().cpu().numpy(), ref.float().cpu().numpy(), trace, ref_trace))
        dist.barrier()
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)
    finally:
        dist.destroy_process_group()


UNI3 = "uni_2_0.75-uni_3_0.5-uni_6_0.25"


@pytest.mark.parametrize("merge,pd,family", [("no_merge", None, "nano"), ("CrossAttention", UNI3, "nano"),
                                             ("CrossAttention", PD, "nano"), ("CrossAttention", UNI3, "qwen2")])
def test_sequence_parallel_hip_matches_single_process(merge, pd, family):
    world, T = 2, 21           # 21 frames x 16 tokens: shards of 11 / 10 frames, ragged scan chunks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, merge, T, q, pd, family)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        logits, ref, trace, ref_trace = q.get(timeout=600)
    finally:
        for p in procs:
            p.join(120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), "worker failed (see its traceback above)"
    logits, ref = torch.from_numpy(logits), torch.from_numpy(ref)
    assert logits.shape == ref.shape and torch.isfinite(logits).all()
    assert len(trace) == len(ref_trace) and all(a.shape == b.shape for a, b in zip(trace, ref_trace))
    rel = ((logits - ref).norm() / ref.norm()).item()
    if pd != PD:
        # uniform stages keep exactly the same tokens; logits agree to bf16 model-level noise
        # (the hipBLASLt stream-K GEMMs alone move them by ~1 % from run to run)
        assert all((a == b).all() for a, b in zip(trace, ref_trace))
        assert rel < 5e-2, rel
    else:
        # "attn" stages rank near-uniform random-init attention in bf16: near-ties may be
        # kept differently by the sharded softmax, so only the uniform stage is compared
        # exactly here (the fp32 gloo test compares every stage exactly)
        assert (trace[0] == ref_trace[0]).all()


def stage_worker(rank, world, port, q):
    """One "attn" pdrop stage on IDENTICAL inputs: the full hidden states cut into two shards
    against the unsharded `pdrop_no_pack`."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stage_collectives_through_host()
        from timeviper_amd.distributed import SequenceParallelTimeViper
        from timeviper_amd.model import build_synthetic_timeviper
        from timeviper_amd.model.llm.nano import NemotronHConfig
        cfg = NemotronHConfig(vocab_size=128, hidden_size=256, intermediate_size=384, num_hidden_layers=8,
                              hybrid_override_pattern="M-M*M-*M", num_attention_heads=8, head_dim=64,
                              num_key_value_heads=2, ssm_state_size=128, mamba_num_heads=8,
                              mamba_n_groups=2, mamba_head_dim=40, mamba_chunk_size=64)
        vlm = build_synthetic_timeviper(cfg, "siglip-vit-b16-224px", pdrop_type="attn_3_0.5-attn_6_0.25",
                                        merge_module="CrossAttention", vit_depth=1, image_size=96, seed=5)
        bb = vlm.llm_backbone.llm.backbone
        n_before, n_vis, n_after = 7, 4096, 9
        L = n_before + n_vis + n_after
        g = torch.Generator(device="cuda").manual_seed(11)
        hidden = torch.randn(1, L, 256, device="cuda", generator=g).bfloat16()
        cut = n_before + 2309                                 # an uneven split inside the vision span
        bounds = [(0, cut), (cut, L)]
        runner = SequenceParallelTimeViper(vlm, rank, world)
        runner.shard_lens = [e - s0 for s0, e in bounds]
        meta = {"num_vision_tokens": n_vis, "vision_index": n_before, "text_prompt_len": n_before + n_after}
        s0, e0 = bounds[rank]
        with torch.no_grad():
            new, new_start, top = runner._pdrop(0, 3, hidden[:, s0:e0].contiguous(), s0, meta)
            if rank == 0:
                bb.last_pdrop_trace = []
                pa = {"first_vision_token_positions": torch.tensor([n_before]), "num_vision_tokens": [n_vis],
                      "text_prompt_lens": [n_before + n_after], "is_interleaved": False}
                _, _, ref_new, _, _ = bb.flash_rank_drop(0, 3, hidden, None, None, None, train_pdrop_args=pa)
                ref_top = bb.last_pdrop_trace[0]["kept"]
                q.put((top.cpu().numpy(), ref_top.cpu().numpy(), new.float().cpu().numpy(),
                       ref_new[:, :new.shape[1]].float().cpu().numpy(), runner.shard_lens))
        dist.barrier()
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)
    finally:
        dist.destroy_process_group()


def test_sharded_attn_rank_keeps_exactly_the_single_gpu_tokens():
    """north_star: bit-exact token-drop masks.  The sharded stage runs the SAME kernels as the
    unsharded one (tv_attn_rank_logits on each rank's keys -> all-gather -> tv_attn_rank_scores_from_logits
    + the same stable sort), so on identical inputs two ranks keep exactly the tokens one GPU keeps."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=stage_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        top, ref_top, new, ref_new, lens = q.get(timeout=600)
    finally:
        for p in procs:
            p.join(120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), "worker failed (see its traceback above)"
    assert top.shape == ref_top.shape == (2048,) and (top == ref_top).all(), "kept indices differ"
    assert sum(lens) == 7 + 2048 + 9
    assert (new == ref_new).all()                 # rank 0's rows: pre-vision text + its kept vision rows
