"""CPU oracle for the composite layers of the hot path — TEST INFRASTRUCTURE ONLY.

Functional (state_dict + config in, tensors out) restatement of the reference's
NemotronH hybrid stack with pdrop / TransV, so the same weights can be pushed
through the reference (golden fixtures), this oracle (CPU) and the product
(`timeviper_amd`, GPU).  Line references are to
`timeviper/model/llm/llm_repo/nano/modeling_nano.py` unless stated otherwise.
See `oracle/ops.py` for the pinning status.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import ops


@dataclass
class OracleConfig:
    hidden_size: int
    num_hidden_layers: int
    hybrid_override_pattern: str
    mamba_num_heads: int
    mamba_head_dim: int
    ssm_state_size: int
    n_groups: int
    conv_kernel: int
    chunk_size: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int
    intermediate_size: int
    layer_norm_epsilon: float = 1e-5
    time_step_limit: Tuple[float, float] = (0.0, float("inf"))
    pdrop_type: Optional[str] = None            # "type_layer_ratio-..."  (:1469-1479)
    merge_module: str = "no_merge"
    group_map: str = "block"                    # "tile" = reference CPU quirk (:781-782)

    @property
    def block_types(self) -> List[str]:
        m = {"M": "mamba", "*": "attention", "-": "mlp"}
        return [m[c] for c in self.hybrid_override_pattern]

    @classmethod
    def from_hf(cls, cfg, **over):
        keys = ["hidden_size", "num_hidden_layers", "hybrid_override_pattern", "mamba_num_heads",
                "mamba_head_dim", "ssm_state_size", "n_groups", "conv_kernel", "chunk_size",
                "num_attention_heads", "num_key_value_heads", "head_dim", "intermediate_size",
                "layer_norm_epsilon", "time_step_limit", "pdrop_type", "merge_module"]
        d = {k: getattr(cfg, k) for k in keys}
        d["time_step_limit"] = tuple(d["time_step_limit"])
        d.update(over)
        return cls(**d)


def _lin(x, sd, name):
    w = sd[name + ".weight"]
    b = sd.get(name + ".bias")
    return F.linear(x, w, b)


# ------------------------------------------------------------------------ S1
def mamba_mixer_ref(sd: Dict[str, torch.Tensor], pfx: str, cfg: OracleConfig, hidden,
                    initial_states=None, conv_halo=None, return_states=False):
    """NemotronHMamba2Mixer inference branch (cuda_kernels_forward :582-667 ==
    torch_forward :671-859): in_proj -> [gate | xBC | dt] -> conv+SiLU -> [x|B|C]
    -> SSD scan -> gated RMSNorm -> out_proj."""
    H, P, N, G = cfg.mamba_num_heads, cfg.mamba_head_dim, cfg.ssm_state_size, cfg.n_groups
    d_inner = H * P
    conv_dim = d_inner + 2 * G * N
    Bsz, L, _ = hidden.shape
    proj = _lin(hidden, sd, pfx + "in_proj")
    gate, xBC, dt = proj.split([d_inner, conv_dim, H], dim=-1)                       # :583-592
    xBC_pre = xBC
    xBC = ops.causal_conv1d_ref(xBC, sd[pfx + "conv1d.weight"].squeeze(1),
                                sd.get(pfx + "conv1d.bias"), "silu", conv_halo).to(hidden.dtype)
    x, Bm, Cm = xBC.split([d_inner, G * N, G * N], dim=-1)                           # :628-636
    A = -torch.exp(sd[pfx + "A_log"].float())                                        # :550
    y, final, decay = ops.ssd_chunk_scan_ref(
        x.reshape(Bsz, L, H, P), dt, A, Bm.reshape(Bsz, L, G, N), Cm.reshape(Bsz, L, G, N),
        cfg.chunk_size, D=sd[pfx + "D"], dt_bias=sd[pfx + "dt_bias"], dt_softplus=True,
        dt_limit=cfg.time_step_limit, initial_states=initial_states, group_map=cfg.group_map)
    y = y.reshape(Bsz, L, d_inner)
    y = ops.rmsnorm_gated_ref(y, sd[pfx + "norm.weight"], gate, cfg.layer_norm_epsilon,
                              d_inner // G)                                          # :664
    out = _lin(y.to(hidden.dtype), sd, pfx + "out_proj")                             # :667
    if return_states:
        K = cfg.conv_kernel
        conv_state = F.pad(xBC_pre.transpose(1, 2), (K - L, 0)) if L < K \
            else xBC_pre.transpose(1, 2)[..., -K:]                                   # :596-610
        return out, final, conv_state, decay
    return out


# ------------------------------------------------------------------------ A1
def attention_mixer_ref(sd, pfx, cfg: OracleConfig, hidden, return_kv=False):
    """NemotronHFlashAttention2 / Sdpa forward (:1134-1220 / :1233-1314): q/k/v
    projections, causal GQA attention without positional encoding, o_proj."""
    Bsz, L, _ = hidden.shape
    Hq, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    q = _lin(hidden, sd, pfx + "q_proj").view(Bsz, L, Hq, D)
    k = _lin(hidden, sd, pfx + "k_proj").view(Bsz, L, Hkv, D)
    v = _lin(hidden, sd, pfx + "v_proj").view(Bsz, L, Hkv, D)
    o, _ = ops.attention_ref(q, k, v, causal=True)
    out = _lin(o.reshape(Bsz, L, Hq * D).to(hidden.dtype), sd, pfx + "o_proj")
    return (out, k, v) if return_kv else out


def mlp_mixer_ref(sd, pfx, hidden):
    """NemotronHMLP (:993-994) with relu2: down(relu(up x)^2)."""
    return _lin(F.relu(_lin(hidden, sd, pfx + "up_proj")).pow(2), sd, pfx + "down_proj")


def cross_attention_ref(sd, pfx, cfg: OracleConfig, text, dropped):
    """Qwen2VLSdpaCrossAttention.forward (merge_modules/cross_attention.py:226-324):
    non-causal GQA attention, Q = text tokens, K/V = dropped vision tokens."""
    Bsz, Lq, _ = text.shape
    Lk = dropped.shape[1]
    Hq, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    q = _lin(text, sd, pfx + "q_proj").view(Bsz, Lq, Hq, D)
    k = _lin(dropped, sd, pfx + "k_proj").view(Bsz, Lk, Hkv, D)
    v = _lin(dropped, sd, pfx + "v_proj").view(Bsz, Lk, Hkv, D)
    o, _ = ops.attention_ref(q, k, v, causal=False)
    return _lin(o.reshape(Bsz, Lq, Hq * D).to(text.dtype), sd, pfx + "o_proj")


# ------------------------------------------------------------------- T1/T2/T3
def parse_pdrop(pdrop_type: str):
    """'type_layer_ratio-...' -> types, layers, ratios (with the leading 1)  (:1469-1479)."""
    parts = [t.split("_") for t in pdrop_type.split("-")]
    assert all(len(p) == 3 for p in parts)
    return [p[0] for p in parts], [int(p[1]) for p in parts], [1] + [float(p[2]) for p in parts]


def pdrop_stage_ref(sd, bb: str, cfg: OracleConfig, features, stage: int, rank_layer: int,
                    vision_index: int, num_vision_tokens: int, text_prompt_len: int,
                    forced_kept=None):
    """pdrop_no_pack, eval / batch 1 / no padding (:1779-2095).  features (1,L,D).
    Returns (new_features (1,L',D), kept_indices (sorted, absolute), dropped_indices)."""
    types, layers, ratios = parse_pdrop(cfg.pdrop_type)
    ctype = types[stage]
    image_tokens = int(num_vision_tokens * ratios[stage])                 # :1795-1798
    keep = int(num_vision_tokens * ratios[stage + 1])                     # :1799-1802
    feats = features[0]
    if "attn" in ctype:
        pfx = f"{bb}layers.{rank_layer}.mixer."
        prompt_total_len = text_prompt_len + image_tokens                 # :1914
        scores = ops.attn_rank_scores_ref(
            feats, sd[pfx + "q_proj.weight"], sd[pfx + "k_proj.weight"], cfg.num_attention_heads,
            cfg.num_key_value_heads, cfg.head_dim, prompt_total_len - 1, vision_index,
            image_tokens)
        top = ops.topk_keep_ref(scores, keep)                             # :1942
    elif "uni" in ctype:
        top = ops.uniform_keep_indices_ref(image_tokens, keep)            # :1946-1953
    else:
        raise NotImplementedError(ctype)
    top = (top + vision_index).sort().values                              # :1957-1958
    if forced_kept is not None:   # tests: follow the selection made by a lower-precision run
        assert forced_kept.numel() == top.numel()
        top = forced_kept.sort().values
    start_index = vision_index + image_tokens                             # :1961
    all_idx = torch.arange(vision_index, start_index)
    dropped = all_idx[~torch.isin(all_idx, top)]                          # :1966-1970
    text = feats[start_index:]
    if cfg.merge_module == "CrossAttention" and "drop" not in ctype:      # :1482-1500, :1962-1979
        # alpha / merge_modules are indexed by stage (cur_num)  (:1761-1768)
        merged = cross_attention_ref(sd, f"{bb}merge_modules.{stage}.", cfg, text[None],
                                     feats[dropped][None])[0]
        text = text + torch.tanh(sd[f"{bb}alpha"][stage]) * merged
    new = torch.cat([feats[:vision_index], feats[top], text.to(feats.dtype)], dim=0)   # :1982-1989
    return new[None], top, dropped


# ------------------------------------------------------------------------ L1
def backbone_ref(sd, cfg: OracleConfig, inputs_embeds, pdrop_args: Optional[dict] = None,
                 bb: str = "backbone.", collect: Optional[dict] = None, forced_kept=None):
    """NemotronHModel.forward (:1550-1746): per layer [pdrop before the block
    (:1635-1665)] -> x + mixer(RMSNorm(x)) (:929-967) ; final norm_f (:1715)."""
    h = inputs_embeds
    types = layers = None
    if cfg.pdrop_type is not None and pdrop_args is not None:
        types, layers, _ = parse_pdrop(cfg.pdrop_type)
    for i, bt in enumerate(cfg.block_types):
        if layers is not None and i in layers and h.shape[1] != 1:
            stage = layers.index(i)
            h, kept, dropped = pdrop_stage_ref(
                sd, bb, cfg, h, stage, i, int(pdrop_args["first_vision_token_positions"][0]),
                int(pdrop_args["num_vision_tokens"][0]), int(pdrop_args["text_prompt_lens"][0]),
                None if forced_kept is None else forced_kept[stage])
            if collect is not None:
                collect.setdefault("kept", []).append(kept)
                collect.setdefault("dropped", []).append(dropped)
        pfx = f"{bb}layers.{i}."
        normed = ops.rmsnorm_ref(h, sd[pfx + "norm.weight"], cfg.layer_norm_epsilon).to(h.dtype)
        if bt == "mamba":
            o = mamba_mixer_ref(sd, pfx + "mixer.", cfg, normed)
        elif bt == "attention":
            o = attention_mixer_ref(sd, pfx + "mixer.", cfg, normed)
        else:
            o = mlp_mixer_ref(sd, pfx + "mixer.", normed)
        h = h + o.to(h.dtype)
        if collect is not None:
            collect.setdefault("hidden", []).append(h)
    return ops.rmsnorm_ref(h, sd[f"{bb}norm_f.weight"], cfg.layer_norm_epsilon).to(h.dtype)


def causal_lm_ref(sd, cfg: OracleConfig, inputs_embeds, pdrop_args=None, last_only=False,
                  collect=None, forced_kept=None):
    """NemotronHForCausalLM.forward (:2378-2457): lm_head(backbone(x)).float()."""
    h = backbone_ref(sd, cfg, inputs_embeds, pdrop_args, collect=collect, forced_kept=forced_kept)
    if last_only:
        h = h[:, -1:]
    return F.linear(h, sd["lm_head.weight"]).float()


# ------------------------------------------------------------------------ F1
def fuse_embeddings_ref(input_ids, visual_embeddings, embed_weight, image_token_id):
    """GenericTimeViperVLM.get_fused_data_nopacked (timeviper/model/generic_vlm.py:517-564),
    batch 1: text-before, then one (tokens_per_frame, D) block per <image> placeholder with
    any text between placeholders, then the trailing text."""
    ids = input_ids[0]
    pos = (ids == image_token_id).nonzero(as_tuple=False).flatten().tolist()
    out = [F.embedding(ids[: pos[0]], embed_weight)]
    for i, sidx in enumerate(pos):
        out.append(visual_embeddings[i].to(embed_weight.dtype))
        start = sidx + 1
        end = pos[i + 1] if i < len(pos) - 1 else ids.shape[0]
        if start < ids.shape[0] and ids[start] == image_token_id:
            continue
        out.append(F.embedding(ids[start:end], embed_weight))
    return torch.cat(out, dim=0)[None]


def pdrop_bookkeeping_ref(input_ids, n_frames, tokens_per_frame, image_token_id):
    """generic_vlm.py:291-309."""
    is_img = input_ids.eq(image_token_id)
    return {
        "first_vision_token_positions": torch.argmax(is_img.int(), dim=1),
        "text_prompt_lens": [int(input_ids.shape[1] - is_img.sum(dim=1)[0])],
        "num_vision_tokens": [n_frames * tokens_per_frame],
        "is_interleaved": False,
    }
