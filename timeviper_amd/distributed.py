"""Sequence-sharded TimeViper forward over the GPUs of one node.

The reference has no collective anywhere (multi-GPU is delegated to DeepSpeed / vLLM,
SURVEY §2a); this is new design for the north star's "frame/token sequence shards across
the 8 GPUs with RCCL all-gather of the SSM states over xGMI".  One process per GPU,
`torch.distributed` backend "nccl" (= RCCL on ROCm); every rank holds the full weights.

  * ViT + ToMe + projector: frames are independent -> rank r encodes its contiguous frame
    range, which IS its token shard (16 tokens per frame); the text before the video goes
    to rank 0's shard, the text after it to the last rank's.  No collective.
  * RMSNorm / MLP / all GEMMs: row-local, no communication.
  * Mamba-2 mixer: the causal conv needs the K-1 = 3 pre-conv rows of the previous shard
    (74 KB halo); the scan runs on the shard from a zero state, the ranks all-gather their
    final states S_r (H,P,N fp32 = 5.2 MB) and total log-decays L_r (H), every rank chains
      In_0 = 0,  In_r = exp(L_{r-1}) In_{r-1} + S_{r-1}
    locally and ranks > 0 add the carried-in term exp(cs_t) C_t . In_r to their outputs in place
    (tv_ssd_state_correction; it stops at each head's decay horizon).
  * attention (4 layers): K/V all-gather (variable shard lengths), causal attention of the
    local queries against the keys up to the shard's end (bottom-right aligned mask).
  * TransV / pdrop: "uni" indices are computed identically on every rank; "attn" scores need one
    softmax over all keys -> every rank computes the logits of ITS keys (tv_attn_rank_logits), the
    (keys, heads) pieces are all-gathered, and every rank runs tv_attn_rank_scores_from_logits + the
    same stable sort on the identical array — the single-GPU kernels on the same numbers, so one GPU
    and N GPUs keep the same tokens bit for bit; every rank keeps its own rows; the dropped rows' K/V
    for the TransV cross-attention are gathered to the rank that owns the trailing text.
    After a token-drop stage the shards can be moved back to an even split (`rebalance_rows`: ONE
    all_to_all_single of contiguous row ranges, order kept; option `rebalance` / env TV_SP_REBALANCE,
    off by default: uniform and synthetic "attn" stages leave the shards within a few percent of even;
    the kept tokens of a real checkpoint may cluster).
Collectives per Mamba layer: two — the conv halo (needed BEFORE the conv, whose output feeds the scan)
and ONE gather of [S_r | L_r] packed in a single fp32 buffer (after the scan); the dependency
in_proj -> halo -> conv -> scan -> state leaves no way to merge the two.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
import torch.distributed as dist

from . import kernels as K


# ------------------------------------------------------------------ collectives
def all_gather_varlen(t: torch.Tensor, group=None, lens: Optional[List[int]] = None) -> List[torch.Tensor]:
    """all-gather of tensors that differ in dim 0 (padded to the longest).  `lens`: every rank's
    dim-0 length when the caller already knows them on the host (saves the size exchange and its
    device-to-host synchronisation)."""
    world = dist.get_world_size(group)
    if lens is None:
        n = torch.tensor([t.shape[0]], device=t.device, dtype=torch.int64)
        ns = [int(v) for v in all_gather_stack(n, group).view(-1).tolist()]
    else:
        ns = [int(v) for v in lens]
        assert len(ns) == world and ns[dist.get_rank(group)] == t.shape[0], (ns, t.shape)
    mx = max(ns)
    if mx == 0:
        return [t[:0] for _ in range(world)]
    pad = torch.empty((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    if t.shape[0] < mx:
        pad[t.shape[0]:].zero_()
    out = all_gather_stack(pad, group)          # one tensor (world, mx, ...): the returned pieces are views of it
    return [out[r, :k] for r, k in enumerate(ns)]


def all_gather_stack(t: torch.Tensor, group=None) -> torch.Tensor:
    """(world, *t.shape): every rank's equally-shaped `t`, gathered straight into one tensor
    (no per-rank temporaries, unlike the list form of all_gather)."""
    world = dist.get_world_size(group)
    t = t.contiguous()
    flat = t.reshape(1, -1)                     # (world * 1, numel): the concatenating form every backend takes
    out = torch.empty((world, flat.shape[1]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    return out.view((world,) + tuple(t.shape))


def all_gather_stack_async(t: torch.Tensor, group=None):
    """The same gather started asynchronously (RCCL runs it on its own stream): returns (out, work); `out` —
    (world, *t.shape) — holds every rank's `t` once `work.wait()` has returned."""
    world = dist.get_world_size(group)
    t = t.contiguous()
    flat = t.reshape(1, -1)
    out = torch.empty((world, flat.shape[1]), dtype=t.dtype, device=t.device)
    work = dist.all_gather_into_tensor(out, flat, group=group, async_op=True)
    return out.view((world,) + tuple(t.shape)), work


def balanced_lens(total: int, world: int) -> List[int]:
    """contiguous shards of `total` rows as even as possible (the LAST total % world ranks hold one more: the last
    rank, which owns the final token, is never the empty one)"""
    return [total // world + (1 if r >= world - total % world else 0) for r in range(world)]


def rebalance_rows(x: torch.Tensor, lens: List[int], group=None, target: Optional[List[int]] = None):
    """Move the shard boundaries of a row-sharded sequence without changing its order: rank r holds rows
    [sum(lens[:r]), sum(lens[:r+1])) in `x` (dim 0) and ends up with [sum(target[:r]), sum(target[:r+1]))
    (default: `balanced_lens`).  One all-to-all of row ranges — a rank only exchanges rows with the ranks whose new
    range overlaps its old one (after an "attn" token-drop stage whose kept tokens cluster on a few ranks, §6 of
    DESIGN.md); every split size is a host integer derived from `lens`, no size exchange.  Returns (x', target)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    total = sum(lens)
    target = balanced_lens(total, world) if target is None else [int(t) for t in target]
    assert sum(target) == total and len(lens) == world == len(target) and lens[rank] == x.shape[0], (lens, target, x.shape)
    if list(lens) == list(target):
        return x, target
    old_lo = [sum(lens[:r]) for r in range(world)]
    new_lo = [sum(target[:r]) for r in range(world)]

    def overlap(a_lo, a_n, b_lo, b_n):
        return max(0, min(a_lo + a_n, b_lo + b_n) - max(a_lo, b_lo))
    send = [overlap(old_lo[rank], lens[rank], new_lo[d], target[d]) for d in range(world)]     # in rank order = row order
    recv = [overlap(old_lo[s], lens[s], new_lo[rank], target[rank]) for s in range(world)]
    out = torch.empty((target[rank],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_to_all_single(out, x.contiguous(), output_split_sizes=recv, input_split_sizes=send, group=group)
    return out, target


def chain_states(states: torch.Tensor, decays: torch.Tensor, rank: int) -> torch.Tensor:
    """Incoming SSM state of shard `rank` from the per-shard (zero-init) final states
    (R,B,H,P,N) and total log-decays (R,B,H):  In_r = exp(L_{r-1}) In_{r-1} + S_{r-1}."""
    inc = torch.zeros_like(states[0])
    for j in range(rank):
        inc = inc * torch.exp(decays[j])[..., None, None] + states[j]
    return inc


def split_frames(n_frames: int, world: int, causal_skew: float = 0.0, align: int = 1) -> List[Tuple[int, int]]:
    """Contiguous frame ranges, one per rank.  `causal_skew` = k >= 0 models a rank's step time as
    f_r (1 + k (F_before_r + f_r / 2)): work linear in its frames (ViT, GEMMs, scans) plus causal
    attention of its frames against everything before them.  k = 0 gives the even split; k > 0
    hands later ranks fewer frames so that all ranks finish together (k from
    `estimate_causal_skew`; at 10 240 frames over 8 ranks the last rank's attention is otherwise
    1.9x the mean: ~3 % of the step).  `align`: shard boundaries fall on multiples of `align` frames."""
    if causal_skew <= 0.0 or world == 1 or n_frames < 2 * world:
        base, rem = divmod(n_frames, world)
        sizes = [base + (1 if r < rem else 0) for r in range(world)]
    else:
        k = float(causal_skew)

        def sizes_for(cost):
            out, before = [], 0.0
            for _ in range(world):
                a = 1.0 + k * before
                f = (-a + math.sqrt(a * a + 2.0 * k * cost)) / k
                out.append(f)
                before += f
            return out
        lo, hi = 0.0, n_frames * (1.0 + k * n_frames)
        for _ in range(80):                       # bisection on the common per-rank cost
            mid = 0.5 * (lo + hi)
            lo, hi = (mid, hi) if sum(sizes_for(mid)) < n_frames else (lo, mid)
        real = sizes_for(hi)
        sizes = [int(v) for v in real]            # largest remainders take the frames left over
        order = sorted(range(world), key=lambda r: real[r] - sizes[r], reverse=True)
        for r in order[: n_frames - sum(sizes)]:
            sizes[r] += 1
    if align > 1:      # boundaries on multiples of `align` frames (towers that regroup frames per clip)
        cuts, acc = [], 0
        for n in sizes[:-1]:
            acc += n
            cuts.append(min(n_frames, int(round(acc / align)) * align))
        cuts = [0] + [max(c, 0) for c in cuts] + [n_frames]
        for i in range(1, len(cuts)):
            cuts[i] = max(cuts[i], cuts[i - 1])
        if n_frames >= align * world and any(cuts[i + 1] == cuts[i] for i in range(world)):
            # rounding emptied a shard although every rank could have a whole clip: hand out whole clips evenly
            clips = n_frames // align
            per, rem = divmod(clips, world)
            cuts = [0]
            for r in range(world):
                cuts.append(cuts[-1] + (per + (1 if r < rem else 0)) * align)
            cuts[-1] = n_frames
        return [(cuts[i], cuts[i + 1]) for i in range(world)]
    out, lo_f = [], 0
    for n in sizes:
        out.append((lo_f, lo_f + n))
        lo_f += n
    return out


# effective rates behind `estimate_causal_skew` (measured on MI355X): the whole
# forward without the LLM attention sustains ~0.95 PFLOP/s of its linear-layer FLOPs, the causal
# attention kernel 0.86 PFLOP/s
_LINEAR_RATE, _ATTN_RATE = 0.95e15, 0.86e15


def estimate_causal_skew(vlm, tokens_per_frame: int) -> float:
    """k of `split_frames` from the model's own shapes: seconds of causal attention per
    (query frame, key frame) pair over seconds of everything else per frame."""
    import torch.nn as nn
    lin = lambda m: sum(p.weight.numel() for p in m.modules() if isinstance(p, nn.Linear))
    bb = vlm.llm_backbone.llm.backbone
    vb = vlm.vision_backbone
    patches = getattr(vb, "num_patches", 0)
    flops_lin = 2.0 * lin(vb) * patches
    keep, attn_pair = 1.0, 0.0
    ratios = list(getattr(bb, "pdrop_ratios", [1])) if getattr(bb, "use_pdrop", False) else [1]
    layers = list(getattr(bb, "pdrop_layers", [])) if getattr(bb, "use_pdrop", False) else []
    for i, block in enumerate(bb.layers):
        if i in layers:
            keep = float(ratios[layers.index(i) + 1])
        flops_lin += 2.0 * lin(block) * tokens_per_frame * keep
        mx = block.mixer if getattr(block, "block_type", "") == "attention" else getattr(block, "self_attn", None)
        if mx is not None:          # a hybrid stack's attention layers / every Qwen2 layer
            attn_pair += 4.0 * (tokens_per_frame * keep) ** 2 * mx.head_dim * mx.num_heads
    if flops_lin <= 0.0:
        return 0.0
    # TV_SP_RATES="<linear TFLOP/s>,<causal-attention TFLOP/s>": the two kernel rates of the GPU at hand (the module
    # constants are this repo's MI355X measurements); TV_SP_CAUSAL_SKEW=<k> overrides the estimate
    if os.environ.get("TV_SP_CAUSAL_SKEW"):
        return float(os.environ["TV_SP_CAUSAL_SKEW"])
    lin_rate, attn_rate = _LINEAR_RATE, _ATTN_RATE
    if os.environ.get("TV_SP_RATES"):
        lin_rate, attn_rate = (float(v) * 1e12 for v in os.environ["TV_SP_RATES"].split(","))
    return (attn_pair / attn_rate) / (flops_lin / lin_rate)


def _global_rank(group, group_rank: int) -> int:
    """torch's broadcast `src` is a GLOBAL rank; a rank index inside `group` must be translated."""
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


class SequenceParallelTimeViper:
    def __init__(self, vlm, rank: int, world: int, group=None, causal_skew: Optional[float] = None,
                 rebalance: Optional[float] = None):
        """`causal_skew`: see `split_frames`; None = estimate it from the model (every rank computes
        the same number from the same shapes), 0 = even frame split.
        `rebalance`: after a token-drop stage, move the shard boundaries back to an even split (`rebalance_rows`)
        when the longest shard exceeds `rebalance` x the mean (e.g. 1.25); None (default, env TV_SP_REBALANCE) =
        keep the shards as the stage leaves them."""
        self.vlm, self.rank, self.world, self.group = vlm, rank, world, group
        self.causal_skew = causal_skew
        if rebalance is None and os.environ.get("TV_SP_REBALANCE"):
            rebalance = float(os.environ["TV_SP_REBALANCE"])
        self.rebalance = rebalance
        self.rebalanced = 0                             # stages after which rows moved (tests, logs)
        self.shard_lens: Optional[List[int]] = None     # host-side shard lengths (set by forward)
        self.llm = vlm.llm_backbone.llm
        self.family = vlm.llm_backbone.llm_family
        if self.family not in ("nano", "qwen2"):
            raise NotImplementedError(f"sequence parallelism: unknown LLM family `{self.family}`")
        self.bb = self.llm.backbone            # NemotronHModel / Qwen2Model
        self.cfg = self.llm.config

    # ---------------------------------------------------------------- layout
    def frame_range(self, n_frames: int) -> Tuple[int, int]:
        return self.frame_split(n_frames)[self.rank]

    def frame_split(self, n_frames: int) -> List[Tuple[int, int]]:
        if self.causal_skew is None:
            tpf = getattr(self.vlm, "num_compressed_tokens", 16)
            if hasattr(self.vlm.vision_backbone, "backbone_ids"):
                tpf *= len(self.vlm.vision_backbone.backbone_ids)
            k = estimate_causal_skew(self.vlm, tpf)
            # the estimate can be overridden from the environment (TV_SP_CAUSAL_SKEW / TV_SP_RATES), and a launcher may
            # export different values to different ranks: rank 0's number is the one every rank uses — ranks that
            # derived different splits would enter the host-sized collectives below with mismatched lengths
            if self.world > 1 and dist.is_available() and dist.is_initialized():
                box = [k]
                dist.broadcast_object_list(box, src=_global_rank(self.group, 0), group=self.group)
                k = float(box[0])
            self.causal_skew = k
        # towers that regroup the frames of a clip into tubes (InternVideo2, alone or inside a dual
        # encoder) see the same clips as an unsharded run only if the shards start on clip boundaries
        vb = self.vlm.vision_backbone
        align = int(getattr(self.vlm, "vit_clip_frames", 256)) if getattr(vb, "batched_clips", False) else 1
        return split_frames(n_frames, self.world, self.causal_skew, align=align)

    def shard_layout(self, input_ids: torch.Tensor, n_frames: int, tok_per_frame: int):
        """Global token ranges [start, end) of every rank's shard, for a prompt of the form
        [text_before | <image> x T | text_after] (the benchmark / evaluate.py layout)."""
        ids = input_ids[0].cpu()            # one device-to-host copy; everything below is host arithmetic
        is_img = ids == self.vlm.default_token_id
        pos = is_img.nonzero().flatten()
        first, last = int(pos[0]), int(pos[-1])
        assert last - first + 1 == n_frames == int(is_img.sum()), "one contiguous <image> run expected"
        n_before, n_after = first, ids.numel() - last - 1
        bounds = []
        for r, (lo, hi) in enumerate(self.frame_split(n_frames)):
            s = n_before + lo * tok_per_frame if r > 0 else 0
            e = n_before + hi * tok_per_frame + (n_after if r == self.world - 1 else 0)
            bounds.append((s, e))
        return bounds, n_before, n_after

    def _lens(self, mine: int, device) -> List[int]:
        """Every rank's current shard length.  Inside `forward` they are tracked on the host (layout
        + the counts `_pdrop` exchanges anyway); a mixer called on its own asks the other ranks."""
        if self.shard_lens is not None:
            assert self.shard_lens[self.rank] == mine, (self.shard_lens, mine)
            return self.shard_lens
        n = torch.tensor([mine], device=device, dtype=torch.int64)
        return [int(v) for v in all_gather_stack(n, self.group).view(-1).tolist()]

    # ---------------------------------------------------------------- mixers
    def _mamba(self, mixer, normed):
        Bsz, L, _ = normed.shape
        d_in = mixer.intermediate_size
        Kw = mixer.conv_kernel_size
        # halo: the K-1 pre-conv rows that precede this shard.  Every rank contributes its last
        # K-1 rows (front-padded with zeros when its shard is shorter) and their count, so a
        # shard shorter than K-1 — or empty — still hands the right rows to its successors.
        # Those rows come FIRST, out of a product of their own (K-1 rows x conv_dim columns of in_proj: nothing beside the
        # projection of the whole shard), and travel while that projection runs: the layer then has one exposed
        # collective (the shard states behind the scan) instead of two.
        n_tail = min(L, Kw - 1)
        w_in, b_in = mixer.in_proj.weight, mixer.in_proj.bias
        tail = normed.new_zeros((Bsz, Kw - 1, mixer.conv_dim))
        if n_tail:
            tail[:, Kw - 1 - n_tail:] = F.linear(normed[:, L - n_tail:], w_in[d_in:d_in + mixer.conv_dim],
                                                 None if b_in is None else b_in[d_in:d_in + mixer.conv_dim])
        tails, tails_work = all_gather_stack_async(tail, self.group)
        proj = mixer.in_proj(normed)
        gate, xBC, dt = proj.split([d_in, mixer.conv_dim, mixer.num_heads], dim=-1)
        tails_work.wait()
        cnts = [min(n, Kw - 1) for n in self._lens(L, xBC.device)]   # host-side inside forward(): no sync
        halo = None
        if self.rank > 0:
            rows, need = [], Kw - 1
            for j in range(self.rank - 1, -1, -1):
                take = min(cnts[j], need)
                if take:
                    rows.insert(0, tails[j][:, Kw - 1 - take:])
                    need -= take
                if need == 0:
                    break
            if need < Kw - 1:
                halo = torch.cat(rows, dim=1)
                if need:        # sequence start reached: zeros in front, like the unsharded conv
                    halo = torch.cat([halo.new_zeros((Bsz, need, halo.shape[-1])), halo], dim=1)
                halo = halo.contiguous()
        x, Bm, Cm, cb = K.causal_conv1d_xbc(xBC, mixer.conv1d.weight.squeeze(1), mixer.conv1d.bias,
                                            d_in, mixer.n_groups, mixer.ssm_state_size,
                                            activation=mixer.activation, halo=halo, return_cb=True)
        xh = x.view(Bsz, L, mixer.num_heads, mixer.head_dim)
        A, D32, dtb32 = mixer._consts()          # fp32, derived once per parameter version
        kw = dict(chunk_size=mixer.chunk_size, D=D32, dt_bias=dtb32, dt_softplus=True,
                  return_final_states=True, group_map=mixer.group_map)
        if mixer.time_step_limit != (0.0, float("inf")):
            kw["dt_limit"] = mixer.time_step_limit
        y, S, dec = K.mamba_chunk_scan_combined(xh, dt, A, Bm, Cm, return_total_decay=True, cb=cb, **kw)
        # final state and total log-decay of the shard travel in ONE collective (5.2 MB + 512 B at Nano dims)
        nS = S.numel()
        packed = torch.empty(nS + dec.numel(), dtype=torch.float32, device=S.device)
        packed[:nS] = S.reshape(-1)
        packed[nS:] = dec.reshape(-1)
        both = all_gather_stack(packed, self.group)
        S_all = both[:, :nS].view((self.world,) + tuple(S.shape))
        d_all = both[:, nS:].view((self.world,) + tuple(dec.shape))
        if self.rank > 0 and L > 0:
            # the state entering this shard, then the carried-in term y_t += exp(cs_t) C_t . In_r added
            # in place — not a second scan: the term dies out after each head's
            # decay horizon and the kernel stops there
            inc = chain_states(S_all, d_all, self.rank)
            dtl = {} if "dt_limit" not in kw else {"dt_limit": kw["dt_limit"]}
            if y.dtype == torch.bfloat16 and mixer.ssm_state_size == 128 and mixer.head_dim % 8 == 0 \
                    or not y.is_cuda:
                y = K.ssd_state_correction(y, dt, A, Cm, inc, dt_bias=dtb32, dt_softplus=True,
                                           group_map=mixer.group_map, **dtl)
            else:       # shapes outside the correction kernel (fp32 / other d_state): scan again from In_r
                y, _ = K.mamba_chunk_scan_combined(xh, dt, A, Bm, Cm, initial_states=inc, **kw)
        y = mixer.norm(y.view(Bsz, L, d_in), gate)
        return mixer.out_proj(y)

    def _attention(self, attn, normed, rope=None):
        """`rope` = (cos, sin) of this shard's GLOBAL positions (Qwen2): keys are rotated before they are
        gathered (every rank rotates its own rows once), queries after the gather has been started."""
        Bsz, L, _ = normed.shape
        assert Bsz == 1, "the sequence-sharded runner is batch 1 (the evaluation path)"
        kvd = attn.num_key_value_heads * attn.head_dim
        lens = self._lens(L, normed.device)
        mx = max(lens)
        # K and V of this shard in ONE padded buffer, rows interleaved (row i = [K_i | V_i]), gathered asynchronously
        # (RCCL runs the collective on its own stream) while q_proj — the largest of the three GEMMs — computes.
        # In the gathered buffer (world, mx, 2, kvd) the K rows of ALL ranks then have ONE row stride (2 kvd): when the
        # shards before this one are full (balanced split: the usual case) the causal prefix "ranks <= r" is a strided
        # VIEW of it — no compaction copy; tv_flash_attn_fwd takes the row stride.
        Hkv, Dh = attn.num_key_value_heads, attn.head_dim
        kv = torch.empty((mx, 2, kvd), dtype=normed.dtype, device=normed.device)
        kv[:L, 0] = attn.k_proj(normed).view(L, kvd)
        kv[:L, 1] = attn.v_proj(normed).view(L, kvd)
        if L < mx:
            kv[L:].zero_()
        q = None
        if rope is not None and L > 0:       # rotary embedding in place on q and on the k rows of the buffer
            q = attn.q_proj(normed).view(Bsz, L, attn.num_heads, attn.head_dim)
            K.apply_rotary_pos_emb_(q, kv[:L, 0].unflatten(-1, (Hkv, Dh)).unsqueeze(0), *rope)
        gathered = torch.empty((self.world, kv.numel()), dtype=kv.dtype, device=kv.device)
        work = dist.all_gather_into_tensor(gathered, kv.view(1, -1), group=self.group, async_op=True)
        if q is None:
            q = attn.q_proj(normed).view(Bsz, L, attn.num_heads, attn.head_dim)
        work.wait()
        if all(lens[r] == mx for r in range(self.rank)):
            rows = gathered.view(self.world * mx, 2, kvd)[:self.rank * mx + L]
            kf = rows[:, 0].unflatten(-1, (Hkv, Dh)).unsqueeze(0)
            vf = rows[:, 1].unflatten(-1, (Hkv, Dh)).unsqueeze(0)
        else:                                # a ragged shard in front of this one: compact the prefix
            g4 = gathered.view(self.world, mx, 2, kvd)
            upto = self.rank + 1
            kf = torch.cat([g4[r, :lens[r], 0] for r in range(upto)]).view(1, -1, Hkv, Dh)
            vf = torch.cat([g4[r, :lens[r], 1] for r in range(upto)]).view(1, -1, Hkv, Dh)
        scale = getattr(attn, "scaling", None)
        o = K.flash_attn_func(q, kf, vf, softmax_scale=scale, causal=True)      # bottom-right aligned: Lk >= Lq
        return attn.o_proj(o.reshape(Bsz, L, attn.num_heads * attn.head_dim))

    # ---------------------------------------------------------------- pdrop
    def _pdrop(self, stage: int, layer_idx: int, hidden, start: int, meta):
        """Sharded pdrop_no_pack (eval, batch 1).  `start` = global index of hidden[0, 0].
        Returns the new shard and its new global start."""
        bb = self.bb
        nv, vis0, txt_len = meta["num_vision_tokens"], meta["vision_index"], meta["text_prompt_len"]
        image_tokens = int(nv * bb.pdrop_ratios[stage])
        keep = int(nv * bb.pdrop_ratios[stage + 1])
        ctype = bb.pdrop_compress_types[stage]
        feats = hidden[0]
        L = feats.shape[0]
        dev = feats.device
        vis_end = vis0 + image_tokens
        if "attn" in ctype:
            sa = bb._rank_attention(layer_idx)
            row = txt_len + image_tokens - 1                      # global index of the query token
            # the query row lives on the last rank unless the shards were re-balanced and the trailing text spans more
            # than one of them: its owner (host integers) broadcasts its projected queries
            s_all = self._starts()
            owner = max(r for r in range(self.world) if s_all[r] <= row)
            q_row = torch.empty((sa.num_heads, sa.head_dim), dtype=feats.dtype, device=dev)
            if self.rank == owner:
                q_row = sa.q_proj(feats[row - start: row - start + 1]).view(sa.num_heads, sa.head_dim).contiguous()
            dist.broadcast(q_row, src=_global_rank(self.group, owner), group=self.group)
            n_local = max(0, min(L, row + 1 - start))             # local keys that take part
            k_loc = sa.k_proj(feats[:n_local]).view(n_local, sa.num_key_value_heads, sa.head_dim)
            # the SAME two kernels the unsharded model runs (tv_attn_rank_scores = logits of the keys,
            # then statistics + head mean over all of them): each rank computes the logits of its own
            # keys, the (keys, heads) pieces are all-gathered, and every rank ranks the identical array
            # — one GPU and N GPUs keep the same tokens, bit for bit
            lg = K.attn_rank_logits(q_row, k_loc)
            n_all = [max(0, min(n, row + 1 - s0)) for n, s0 in zip(self.shard_lens, self._starts())]
            logits = torch.cat(all_gather_varlen(lg, self.group, n_all))
            assert logits.shape[0] == row + 1
            scores = K.attn_rank_scores_from_logits(logits, vis0, image_tokens, feats.dtype)
            order = torch.sort(scores, descending=True, stable=True).indices
            top = (order[:keep] + vis0).sort().values
        elif "uni" in ctype:
            top = None               # located on the host below (one source for boundaries and rows)
        else:
            raise NotImplementedError(ctype)
        # Where the (identical, sorted) kept indices fall in every rank's old range.  "uni" indices are
        # a pure function of (image_tokens, keep): located on the host from the CPU reference formula
        # (what the kernel reproduces bit for bit) without touching the device; "attn" needs the
        # ranking's outcome: ONE device-to-host copy of world + 1 positions per stage — no collective,
        # and every later shape in this function is a host integer
        starts = self._starts()
        ends = [s0 + n for s0, n in zip(starts, self.shard_lens)]
        edges = starts + [ends[-1]]
        if "uni" in ctype:
            top_h = torch.linspace(0, image_tokens - 1, keep, dtype=torch.long) + vis0
            pos = torch.searchsorted(top_h, torch.tensor(edges)).tolist()
            top = top_h.to(dev)      # ONE source for the boundaries and the gathered rows (the kernel reproduces it bit for bit,
                                     # tests/test_ops_gpu.py; a disagreement would index rows outside the shard)
        else:
            pos = torch.searchsorted(top, torch.tensor(edges, device=dev)).tolist()
        kept_per_rank = [pos[r + 1] - pos[r] for r in range(self.world)]
        # rows of this shard that survive: [pre-vision text | kept vision | trailing text]
        end = start + L
        assert (start, end) == (starts[self.rank], ends[self.rank])
        mine = top[pos[self.rank]: pos[self.rank + 1]]
        pre_hi, post_lo = min(end, vis0), max(start, vis_end)
        parts = []
        if pre_hi > start:
            parts.append(torch.arange(0, pre_hi - start, device=dev))
        parts.append(mine - start)
        n_text = max(0, end - post_lo)
        if n_text:
            parts.append(torch.arange(post_lo - start, L, device=dev))
        new = K.gather_rows(feats, torch.cat(parts))
        # TransV merge: trailing text (last rank) attends to ALL dropped vision rows
        if bb.merge_modules is not None and bb.merge_module_names[stage] != "none":
            v_lo = [max(s0, vis0) for s0 in starts]
            v_n = [max(0, min(e0, vis_end) - lo) for lo, e0 in zip(v_lo, ends)]       # vision rows per rank
            n_drop = [n - k for n, k in zip(v_n, kept_per_rank)]
            mod = bb.merge_modules[stage]
            kvd = mod.num_key_value_heads * mod.head_dim
            if n_drop[self.rank]:
                didx = K.dropped_indices(mine, v_lo[self.rank], v_n[self.rank]) - start
                dropped_local = K.gather_rows(feats, didx)
                kd, vd = mod.k_proj(dropped_local), mod.v_proj(dropped_local)
            else:
                kd = vd = feats.new_empty((0, kvd))
            kd = torch.cat(all_gather_varlen(kd, self.group, n_drop))
            vd = torch.cat(all_gather_varlen(vd, self.group, n_drop))
            if n_text:               # (the last rank; after a re-balance possibly its neighbour too)
                text = new[new.shape[0] - n_text:]
                qd = mod.q_proj(text).view(1, n_text, mod.num_heads, mod.head_dim)
                o = K.flash_attn_func(qd, kd.view(1, -1, mod.num_key_value_heads, mod.head_dim),
                                      vd.view(1, -1, mod.num_key_value_heads, mod.head_dim), causal=False)
                merged = mod.o_proj(o.reshape(n_text, mod.num_heads * mod.head_dim))
                new[new.shape[0] - n_text:] = text + bb.alpha[stage].tanh() * merged
        new_lens = [max(0, min(e0, vis0) - s0) + kept_per_rank[r] + max(0, e0 - max(s0, vis_end))
                    for r, (s0, e0) in enumerate(zip(starts, ends))]
        assert new_lens[self.rank] == new.shape[0], (new_lens, new.shape)
        self.shard_lens = new_lens
        if self.rebalance is not None and self.world > 1:
            total = sum(new_lens)
            if total and max(new_lens) * self.world > self.rebalance * total:
                new, self.shard_lens = rebalance_rows(new, new_lens, self.group)
                self.rebalanced += 1
        new_start = sum(self.shard_lens[: self.rank])
        return new.unsqueeze(0), new_start, top

    def _starts(self) -> List[int]:
        out, acc = [], 0
        for n in self.shard_lens:
            out.append(acc)
            acc += n
        return out

    # ---------------------------------------------------------------- forward
    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, pixel_values_local: torch.Tensor, n_frames: int,
                visual_embeddings_local: Optional[torch.Tensor] = None):
        """input_ids: the FULL prompt (identical on every rank); pixel_values_local: this
        rank's frames (frame_range).  Returns the last-token logits on every rank."""
        vlm, bb = self.vlm, self.bb
        vis = visual_embeddings_local if visual_embeddings_local is not None \
            else vlm.encode_vision(pixel_values_local, is_video=True)
        tpf = vis.shape[1]
        bounds, n_before, n_after = self.shard_layout(input_ids, n_frames, tpf)
        start, end = bounds[self.rank]
        self.shard_lens = [e - s0 for s0, e in bounds]         # host-side shard lengths, kept current by _pdrop
        embed = vlm.llm_backbone.embed_input_ids
        parts = []
        if self.rank == 0 and n_before:
            parts.append(embed(input_ids[:, :n_before]))
        parts.append(vis.reshape(1, -1, vis.shape[-1]).to(vlm.dtype_of(embed)))
        if self.rank == self.world - 1 and n_after:
            parts.append(embed(input_ids[:, input_ids.shape[1] - n_after:]))
        hidden = torch.cat(parts, dim=1)
        assert hidden.shape[1] == end - start
        meta = {"num_vision_tokens": n_frames * tpf, "vision_index": n_before,
                "text_prompt_len": n_before + n_after}
        logits = self.run_layers(hidden, start, meta)
        self.final_lens, self.shard_lens = list(self.shard_lens), None     # (final_lens: tests, logs)
        return logits

    def run_layers(self, hidden, start: int, meta):
        """The 56-layer loop on this rank's shard.  Outside `_pdrop` it reads nothing back from the
        device: shard lengths live on the host, every collective has host-known sizes
        (tests/test_distributed_cpu.py counts the host reads)."""
        bb = self.bb
        self.trace = []
        if self.family == "qwen2":
            return self._run_layers_qwen2(hidden, start, meta)
        delta = None
        for i, block in enumerate(bb.layers):
            if bb.use_pdrop and i in bb.pdrop_layers:
                if delta is not None:
                    hidden, delta = hidden + delta, None
                stage = bb.pdrop_layers.index(i)
                hidden, start, top = self._pdrop(stage, i, hidden, start, meta)
                self.trace.append(top)
            if delta is None:
                normed = block.norm(hidden)
            else:
                normed, hidden = block.norm(hidden, residual=delta, return_sum=True)
            if block.block_type == "mamba":
                delta = self._mamba(block.mixer, normed)
            elif block.block_type == "attention":
                delta = self._attention(block.mixer, normed)
            else:
                delta = block.mixer(normed)
        hidden = bb.norm_f(hidden, residual=delta) if delta is not None else bb.norm_f(hidden)
        logits = torch.empty((1, 1, self.cfg.vocab_size), dtype=torch.float32, device=hidden.device)
        if self.rank == self.world - 1:
            logits = self.llm.lm_head(hidden[:, -1:]).float()
        dist.broadcast(logits, src=_global_rank(self.group, self.world - 1), group=self.group)
        return logits

    def _run_layers_qwen2(self, hidden, start: int, meta):
        """Qwen2 decoder stack (modeling_qwen2.py:878-1038) on this rank's shard: every layer is attention
        (K/V all-gather, rotary embedding at the shard's global positions) + a row-local SwiGLU MLP;
        positions restart from 0 after each pdrop stage (:918-966), i.e. they are the global row indices
        of the shortened sequence."""
        bb = self.bb

        def rope_of(h, s0):
            pos = torch.arange(s0, s0 + h.shape[1], device=h.device)[None]
            return bb.rotary_emb(h, pos)
        rope = rope_of(hidden, start)
        delta = None
        for i, layer in enumerate(bb.layers):
            if bb.use_pdrop and i in bb.pdrop_layers:
                if delta is not None:
                    hidden, delta = hidden + delta, None
                stage = bb.pdrop_layers.index(i)
                hidden, start, top = self._pdrop(stage, i, hidden, start, meta)
                self.trace.append(top)
                rope = rope_of(hidden, start)
            if delta is None:
                h = layer.input_layernorm(hidden)
            else:
                h, hidden = layer.input_layernorm(hidden, residual=delta, return_sum=True)
            a = self._attention(layer.self_attn, h, rope=rope)
            h, hidden = layer.post_attention_layernorm(hidden, residual=a, return_sum=True)
            delta = layer.mlp(h)
        hidden = bb.norm(hidden, residual=delta) if delta is not None else bb.norm(hidden)
        logits = torch.empty((1, 1, self.cfg.vocab_size), dtype=torch.float32, device=hidden.device)
        if self.rank == self.world - 1:
            logits = self.llm.lm_head(hidden[:, -1:]).float()
        dist.broadcast(logits, src=_global_rank(self.group, self.world - 1), group=self.group)
        return logits
