"""Sequence-parallel runner (timeviper_amd.distributed) with world_size 2 over gloo on the
CPU: the collectives, shard bookkeeping, state chaining, KV gathering and sharded pdrop
must reproduce the single-process forward.  Kernels are replaced by oracle-backed shims
(tests/cpu_kernel_shim.py) — this exercises the host logic above the C ABI only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cpu_kernel_shim import cpu_kernels
from timeviper_amd.distributed import chain_states, split_frames

PD = "uni_2_0.75-attn_3_0.5-attn_6_0.25"


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def build(merge, family="nano"):
    from timeviper_amd.model import GenericTimeViperVLM, HybridTimeViperVLM
    from timeviper_amd.model.llm import GenericLLMBackbone, NemotronHConfig
    from timeviper_amd.model.vit import TimmViTBackbone
    torch.manual_seed(0)
    if family == "qwen2":
        from timeviper_amd.model.llm.qwen2 import Qwen2Config
        cfg = Qwen2Config(vocab_size=128, hidden_size=64, intermediate_size=96, num_hidden_layers=8,
                          num_attention_heads=4, num_key_value_heads=2, rope_theta=10000.0)
        vb = TimmViTBackbone("siglip-vit-b16-224px", depth_override=2, default_image_size=96)
        llm = GenericLLMBackbone("qwen2.5-7b-instruct", config=cfg, merge_module=merge, use_pdrop=True, pdrop_type=PD)
        vlm = GenericTimeViperVLM("t", vb, llm, arch_specifier="tome_mlp-16").eval()
        with torch.no_grad():
            for n, p in vlm.named_parameters():
                if n.endswith("alpha"):
                    p.fill_(0.7)
                elif ("q_proj" in n or "k_proj" in n) and n.endswith("weight"):
                    p.mul_(20.0)
        return vlm
    cfg = NemotronHConfig(vocab_size=128, hidden_size=64, intermediate_size=96, num_hidden_layers=8,
                          hybrid_override_pattern="M-M*M-*M", num_attention_heads=4, head_dim=16,
                          num_key_value_heads=2, ssm_state_size=16, mamba_num_heads=8,
                          mamba_n_groups=2, mamba_head_dim=8, mamba_chunk_size=16)
    vb = TimmViTBackbone("siglip-vit-b16-224px", depth_override=2, default_image_size=96)
    llm = GenericLLMBackbone("nanov2-9b", config=cfg, merge_module=merge, use_pdrop=True, pdrop_type=PD)
    vlm = HybridTimeViperVLM("t", vb, llm, arch_specifier="tome_mlp-16").eval()
    with torch.no_grad():
        for n, p in vlm.named_parameters():
            if n.endswith("alpha"):
                p.fill_(0.7)
            elif "q_proj" in n or "k_proj" in n:
                p.mul_(20.0)
    return vlm


class HostReadCounter:
    """Counts the Tensor methods that copy a value to the host (each one is a device
    synchronisation when the tensor lives on a GPU); `paused()` exempts a region."""
    NAMES = ("item", "tolist", "__int__", "__float__", "__bool__", "__index__", "nonzero", "cpu", "numpy")

    def __init__(self):
        self.count, self.on, self.saved = 0, False, {}

    def __enter__(self):
        for n in self.NAMES:
            orig = getattr(torch.Tensor, n)
            self.saved[n] = orig

            def wrapped(t, *a, __orig=orig, **kw):
                if self.on:
                    self.count += 1
                return __orig(t, *a, **kw)
            setattr(torch.Tensor, n, wrapped)
        self.on = True
        return self

    def __exit__(self, *exc):
        self.on = False
        for n, f in self.saved.items():
            setattr(torch.Tensor, n, f)


def worker(rank, world, port, merge, T, q, family="nano", rebalance=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2 if world <= 4 else 1)
    try:
        from timeviper_amd.distributed import SequenceParallelTimeViper
        vlm = build(merge, family)
        tok = vlm.default_token_id
        g = torch.Generator().manual_seed(1)
        ids = torch.tensor([[5, 6, 7] + [tok] * T + [8, 9, 10, 11, 12]])
        pix = torch.randn(T, 3, 96, 96, generator=g)
        with cpu_kernels(), torch.no_grad():
            # one case with the model-derived frame split, one with a strongly skewed split (4 + 1)
            runner = SequenceParallelTimeViper(vlm, rank, world,
                                               causal_skew=10.0 if merge == "CrossAttention" else None,
                                               rebalance=rebalance)
            lo, hi = runner.frame_range(T)
            if merge == "CrossAttention" and world == 2 and family == "nano":
                assert runner.frame_split(T) == [(0, 4), (4, 5)]
            # host reads: none in the layer loop outside the pdrop stages; inside a stage at most the
            # one copy of world + 1 positions (a .tolist()) — counted here on every rank
            counter, in_stage = HostReadCounter(), [0]
            run_layers, pdrop = runner.run_layers, runner._pdrop

            def counted_layers(*a, **kw):
                with counter:
                    return run_layers(*a, **kw)

            def exempt_pdrop(*a, **kw):
                before = counter.count
                out = pdrop(*a, **kw)
                in_stage[0] += counter.count - before
                counter.count = before
                return out
            runner.run_layers, runner._pdrop = counted_layers, exempt_pdrop
            logits = runner.forward(ids, pix[lo:hi], T)
            n_stages = len(runner.trace)
            assert counter.count == 0, f"{counter.count} host reads in the layer loop"
            assert in_stage[0] <= 2 * n_stages, (in_stage[0], n_stages)
            trace = [t.clone() for t in runner.trace]
            if rank == 0:
                ref = vlm(input_ids=ids, pixel_values_videos=pix).logits
                ref_trace = [t["kept"] for t in vlm.llm_backbone.llm.backbone.last_pdrop_trace]
                q.put((logits.numpy(), ref.numpy(), [t.numpy() for t in trace],
                       [t.numpy() for t in ref_trace], runner.rebalanced, list(runner.final_lens)))
        dist.barrier()
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("merge,world,T,family,rebalance", [
    ("no_merge", 2, 5, "nano", None), ("CrossAttention", 2, 5, "nano", None), ("CrossAttention", 8, 21, "nano", None),
    ("CrossAttention", 2, 7, "qwen2", None), ("no_merge", 4, 9, "qwen2", None),
    # re-balancing after every token-drop stage that leaves the shards uneven (threshold 1.0 = any imbalance; the 4 + 1
    # frame split and the ranking's clustered keep-sets make every stage uneven): rows move between the ranks, the
    # trailing text may end up on two of them, kept tokens and logits stay those of one process
    ("CrossAttention", 2, 5, "nano", 1.0), ("CrossAttention", 4, 9, "nano", 1.0), ("CrossAttention", 4, 9, "qwen2", 1.0)])
def test_sequence_parallel_matches_single_process(merge, world, T, family, rebalance):
    """world 8 = the node size the driver scales to: eight ranks, unequal frame ranges; qwen2 = the
    decoder-only family of BASELINE config 5 (every layer gathers rotated K and V)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, merge, T, q, family, rebalance)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        logits, ref, trace, ref_trace, moved, lens = q.get(timeout=240)   # plain numpy: no shm handles to lose
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), "worker failed (see its traceback above)"
    logits, ref = torch.from_numpy(logits), torch.from_numpy(ref)
    trace, ref_trace = [torch.from_numpy(t) for t in trace], [torch.from_numpy(t) for t in ref_trace]
    assert all(torch.equal(a, b) for a, b in zip(trace, ref_trace)), "kept indices differ"
    assert torch.allclose(logits, ref, rtol=1e-4, atol=1e-5), (logits - ref).abs().max()
    if rebalance is not None:
        assert moved >= 1, "no stage moved rows: the case does not exercise rebalance_rows"
        assert max(lens) - min(lens) <= 1, lens


def rebalance_worker(rank, world, port, lens, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from timeviper_amd.distributed import rebalance_rows
        total = sum(lens)
        full = torch.arange(total * 3, dtype=torch.float32).view(total, 3)
        lo = sum(lens[:rank])
        x, new = rebalance_rows(full[lo:lo + lens[rank]].clone(), lens)
        x2, tgt = rebalance_rows(x, new, target=lens)              # and back to where the rows came from
        q.put((rank, x.numpy(), new, x2.numpy()))
        dist.barrier()
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lens", [[50, 3, 0, 27], [0, 0, 9], [7, 7], [1, 0, 0, 0, 0, 0, 0, 30]])
def test_rebalance_rows_keeps_the_order(lens):
    """a deliberately clustered shard layout (what a top-k token drop can leave): one all-to-all of row ranges gives
    every rank its slice of the even split, in the original order; empty senders / receivers included"""
    from timeviper_amd.distributed import balanced_lens
    world = len(lens)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=rebalance_worker, args=(r, world, port, lens, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = {r: (x, new, x2) for r, x, new, x2 in (q.get(timeout=120) for _ in range(world))}
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), "worker failed (see its traceback above)"
    total = sum(lens)
    full = np.arange(total * 3, dtype=np.float32).reshape(total, 3)
    tgt = balanced_lens(total, world)
    assert sum(tgt) == total and max(tgt) - min(tgt) <= 1 and (total == 0 or tgt[-1] > 0)
    for r in range(world):
        x, new, x2 = got[r]
        assert list(new) == tgt
        lo = sum(tgt[:r])
        assert np.array_equal(x, full[lo:lo + tgt[r]])
        lo0 = sum(lens[:r])
        assert np.array_equal(x2, full[lo0:lo0 + lens[r]])


def test_chain_states_and_split():
    assert split_frames(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    g = torch.Generator().manual_seed(0)
    S = torch.randn(3, 1, 2, 4, 5, generator=g)
    d = -torch.rand(3, 1, 2, generator=g)
    inc2 = chain_states(S, d, 2)
    assert torch.allclose(inc2, torch.exp(d[1])[..., None, None] * S[0] + S[1])
    assert torch.equal(chain_states(S, d, 0), torch.zeros_like(S[0]))


def ragged_worker(rank, world, port, lens, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        from timeviper_amd.distributed import SequenceParallelTimeViper
        vlm = build("no_merge")
        bb = vlm.llm_backbone.llm.backbone
        g = torch.Generator().manual_seed(3)
        L = sum(lens)
        hidden = torch.randn(1, L, 64, generator=g)
        lo = sum(lens[:rank])
        with cpu_kernels(), torch.no_grad():
            runner = SequenceParallelTimeViper(vlm, rank, world)
            outs = []
            for blk in (bb.layers[0], bb.layers[3]):          # a Mamba and an attention layer
                normed = blk.norm(hidden)
                mine = normed[:, lo:lo + lens[rank]]
                part = runner._mamba(blk.mixer, mine) if blk.block_type == "mamba" \
                    else runner._attention(blk.mixer, mine)
                full = blk.mixer(normed)
                full = full[0] if isinstance(full, tuple) else full
                outs.append((part.numpy(), full[:, lo:lo + lens[rank]].numpy()))
            q.put((rank, outs))
        dist.barrier()
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)
    finally:
        dist.destroy_process_group()


def test_ragged_and_empty_shards():
    """Shards shorter than the conv halo (1 token) and empty shards (0 tokens) — what a
    top-k pdrop stage can leave on a rank — still reproduce the unsharded mixers."""
    lens = [5, 1, 0, 6]
    world = len(lens)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=ragged_worker, args=(r, world, port, lens, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = dict(q.get(timeout=240) for _ in range(world))
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), "worker failed (see its traceback above)"
    for r in range(world):
        for part, ref in got[r]:
            assert part.shape == ref.shape
            assert np.allclose(part, ref, rtol=1e-4, atol=1e-5), (r, np.abs(part - ref).max())


def test_split_frames_balances_causal_cost():
    """k = 0 is the even split; k > 0 equalises f_r (1 + k (F_before + f_r / 2)) over the ranks with
    contiguous, exhaustive, monotonically shrinking ranges; degenerate inputs stay valid."""
    from timeviper_amd.distributed import split_frames
    assert split_frames(10240, 8) == [(1280 * r, 1280 * (r + 1)) for r in range(8)]
    k = 9e-6
    sp = split_frames(10240, 8, k)
    assert sp[0][0] == 0 and sp[-1][1] == 10240 and all(a[1] == b[0] for a, b in zip(sp, sp[1:]))
    sizes = [b - a for a, b in sp]
    assert sizes == sorted(sizes, reverse=True) and sizes[0] > 1280 > sizes[-1]
    cost = [(b - a) * (1 + k * (a + (b - a) / 2)) for a, b in sp]
    assert max(cost) / min(cost) < 1.002
    even = [(b - a) * (1 + k * (a + (b - a) / 2)) for a, b in split_frames(10240, 8)]
    assert max(even) > 1.025 * max(cost)                     # what the skew buys: ~3 % of the step
    # clip-aligned boundaries (towers that regroup the frames of a 256-frame clip into tubes)
    al = split_frames(4096, 8, k, align=256)
    assert al[0][0] == 0 and al[-1][1] == 4096 and all(a[1] == b[0] and a[1] % 256 == 0 for a, b in zip(al, al[1:]))
    assert split_frames(4096, 8, 0.0, align=256) == [(512 * r, 512 * (r + 1)) for r in range(8)]
    for n, w in ((3, 8), (17, 8), (1, 2), (0, 4)):
        sp = split_frames(n, w, 1e-2)
        assert len(sp) == w and sp[0][0] == 0 and sp[-1][1] == n
        assert all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(sp, sp[1:]))


def test_causal_skew_estimate_for_the_default_model():
    """k from the 9B model's own shapes (built on the meta device): later ranks get fewer frames, by
    the few percent the causal-attention imbalance is worth."""
    from timeviper_amd.distributed import estimate_causal_skew, split_frames
    from timeviper_amd.model import GenericLLMBackbone, HybridTimeViperVLM
    from timeviper_amd.model.llm.nano import NemotronHConfig
    from timeviper_amd.model.vit import TimmViTBackbone
    with torch.device("meta"):
        vb = TimmViTBackbone("siglip-vit-so400m-384px")
        llm = GenericLLMBackbone("nanov2-9b", config=NemotronHConfig.nemotron_nano_9b_v2(), merge_module="CrossAttention",
                                 use_pdrop=True, pdrop_type="uni_14_0.8-attn_21_0.6-attn_30_0.4-attn_39_0.2")
        vlm = HybridTimeViperVLM("x", vb, llm, arch_specifier="tome_mlp-16")
    k = estimate_causal_skew(vlm, 16)
    assert 5e-6 < k < 1.5e-5
    sizes = [b - a for a, b in split_frames(10240, 8, k)]
    assert sum(sizes) == 10240 and 1300 < sizes[0] < 1360 and 1200 < sizes[-1] < 1260
    with torch.device("meta"):
        llm0 = GenericLLMBackbone("nanov2-9b", config=NemotronHConfig.nemotron_nano_9b_v2())
        vlm0 = HybridTimeViperVLM("x", vb, llm0, arch_specifier="tome_mlp-16")
    assert estimate_causal_skew(vlm0, 16) > k            # without pdrop attention weighs more
