"""Qwen2 / Qwen2.5 backbone with pdrop + TransV (reference
timeviper/model/llm/llm_repo/qwen2/modeling_qwen2.py:59-1197, configuration_qwen2.py) —
SURVEY §8f "next" row, needed by BASELINE config 5.

Inference paths only (prefill and single-token decode, batch 1, no padding mask), same module
tree / parameter names as the reference (`model.embed_tokens`, `model.layers.{i}.{self_attn.
{q,k,v,o}_proj, mlp.{gate,up,down}_proj, input_layernorm, post_attention_layernorm}`,
`model.norm`, `model.merge_modules.*`, `model.alpha`, `lm_head`), so a reference / HF state
dict loads with strict=True.  Device work is on the HIP operators: `kernels.rms_norm` (with
the residual add fused), `kernels.apply_rotary_pos_emb_`, `kernels.flash_attn_func` (causal
GQA, bottom-right aligned for decode), `kernels.silu_mul`, and the shared pdrop / TransV
operators (`pdrop.PdropMixin`).  Linear layers stay on hipBLASLt.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from ... import kernels as K
from .nano import CausalLMOutputWithPast
from .pdrop import PdropMixin


class Qwen2Config:
    """configuration_qwen2.py:25-225 (fields the inference path reads) + the TimeViper extras
    (`use_pdrop`, `pdrop_type`, `merge_module`: llm_factory.py:95-100)."""
    model_type = "qwen2"

    def __init__(self, vocab_size=152064, hidden_size=3584, intermediate_size=18944,
                 num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                 hidden_act="silu", max_position_embeddings=32768, initializer_range=0.02,
                 rms_norm_eps=1e-6, use_cache=True, tie_word_embeddings=False, rope_theta=1000000.0,
                 rope_scaling=None, attention_dropout=0.0, head_dim=None, pad_token_id=None,
                 use_pdrop=False, pdrop_type=None, merge_module="no_merge", **unused):
        if hidden_act != "silu":
            raise NotImplementedError("Qwen2 checkpoints use SwiGLU (hidden_act='silu')")
        if rope_scaling is not None:
            raise NotImplementedError("only the default rotary embedding is on the path")
        self.vocab_size, self.hidden_size, self.intermediate_size = vocab_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.num_key_value_heads = num_key_value_heads
        self.hidden_act, self.max_position_embeddings = hidden_act, max_position_embeddings
        self.initializer_range, self.rms_norm_eps, self.use_cache = initializer_range, rms_norm_eps, use_cache
        self.tie_word_embeddings, self.rope_theta, self.rope_scaling = tie_word_embeddings, rope_theta, rope_scaling
        self.attention_dropout = attention_dropout
        self.head_dim = head_dim if head_dim is not None else hidden_size // num_attention_heads
        self.pad_token_id = pad_token_id
        self.layer_types = ["full_attention"] * num_hidden_layers
        self.use_pdrop, self.pdrop_type, self.merge_module = use_pdrop, pdrop_type, merge_module
        self.attention_bias = True          # q/k/v carry a bias, o_proj does not (:178-190)
        self._attn_implementation = "flash_attention_2"

    @staticmethod
    def qwen2_5_7b(**over) -> "Qwen2Config":
        """Qwen/Qwen2.5-7B-Instruct (public config.json; llm_registry.py:74)."""
        return Qwen2Config(**{**dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944,
                                     num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                                     max_position_embeddings=32768, rope_theta=1000000.0), **over})


class Qwen2KVCache:
    """The slice of transformers' DynamicCache the path uses: per-layer K/V kept (B, L, Hkv, D)."""

    def __init__(self, num_layers: int):
        self.key_cache: List[Optional[torch.Tensor]] = [None] * num_layers
        self.value_cache: List[Optional[torch.Tensor]] = [None] * num_layers

    def get_seq_length(self, layer_idx: int = 0) -> int:
        k = self.key_cache[layer_idx]
        return 0 if k is None else k.shape[1]

    def update(self, k, v, layer_idx):
        if self.key_cache[layer_idx] is None:
            self.key_cache[layer_idx], self.value_cache[layer_idx] = k, v
        else:
            self.key_cache[layer_idx] = torch.cat([self.key_cache[layer_idx], k], dim=1)
            self.value_cache[layer_idx] = torch.cat([self.value_cache[layer_idx], v], dim=1)
        return self.key_cache[layer_idx], self.value_cache[layer_idx]


class Qwen2RMSNorm(nn.Module):
    """:248-265."""

    def __init__(self, hidden_size, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x, residual=None, return_sum=False):
        return K.rms_norm(x, self.weight, self.variance_epsilon, residual=residual, return_sum=return_sum)


class Qwen2RotaryEmbedding(nn.Module):
    """:338-385, default rope: cos / sin of position * theta^(-2i/d), both halves equal."""

    def __init__(self, config: Qwen2Config, device=None):
        super().__init__()
        self.head_dim, self.rope_theta = config.head_dim, config.rope_theta
        self.attention_scaling = 1.0
        # not a buffer: the table must stay fp32 through `.to(bfloat16)` / `to_empty()` (the
        # reference's non-persistent buffer is re-derived from the config on load the same way)
        self._inv_freq = {}

    def inv_freq(self, device) -> torch.Tensor:
        key = str(device)
        if key not in self._inv_freq:
            d = self.head_dim
            self._inv_freq[key] = 1.0 / (self.rope_theta ** (
                torch.arange(0, d, 2, dtype=torch.int64, device=device).float() / d))
        return self._inv_freq[key]

    @torch.no_grad()
    def forward(self, x, position_ids):
        freqs = position_ids[:, :, None].float() * self.inv_freq(x.device)[None, None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        return emb.cos().to(x.dtype), emb.sin().to(x.dtype)


class Qwen2MLP(nn.Module):
    """:67-80."""

    def __init__(self, config):
        super().__init__()
        self.gate_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.up_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)

    def forward(self, x):
        return self.down_proj(K.silu_mul(self.gate_proj(x), self.up_proj(x)))


class Qwen2Attention(nn.Module):
    """:161-245; GQA without repeat_kv, causal, rotary on q and k."""

    def __init__(self, config: Qwen2Config, layer_idx: int):
        super().__init__()
        self.config, self.layer_idx = config, layer_idx
        self.head_dim = config.head_dim
        self.num_heads, self.num_key_value_heads = config.num_attention_heads, config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.scaling = self.head_dim ** -0.5
        self.is_causal = True
        self.q_proj = nn.Linear(config.hidden_size, self.num_heads * self.head_dim, bias=True)
        self.k_proj = nn.Linear(config.hidden_size, self.num_key_value_heads * self.head_dim, bias=True)
        self.v_proj = nn.Linear(config.hidden_size, self.num_key_value_heads * self.head_dim, bias=True)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, config.hidden_size, bias=False)

    def forward(self, hidden_states, position_embeddings, attention_mask=None, past_key_values=None,
                cache_position=None, **kwargs):
        B, L, _ = hidden_states.shape
        q = self.q_proj(hidden_states).view(B, L, self.num_heads, self.head_dim)
        k = self.k_proj(hidden_states).view(B, L, self.num_key_value_heads, self.head_dim)
        v = self.v_proj(hidden_states).view(B, L, self.num_key_value_heads, self.head_dim)
        cos, sin = position_embeddings
        K.apply_rotary_pos_emb_(q, k, cos, sin)
        if past_key_values is not None:
            k, v = past_key_values.update(k, v, self.layer_idx)
        o = K.flash_attn_func(q, k, v, softmax_scale=self.scaling, causal=True)
        return self.o_proj(o.reshape(B, L, self.num_heads * self.head_dim)), None


class Qwen2CrossAttention(nn.Module):
    """TransV merge module of the Qwen2 family (qwen2/merge_modules/cross_attention.py:65-324):
    text tokens (Q) attend non-causally to the dropped vision tokens (K, V); q/k/v with bias."""

    def __init__(self, config, layer_idx: Optional[int] = None):
        super().__init__()
        self.config, self.layer_idx = config, layer_idx
        self.hidden_size, self.num_heads = config.hidden_size, config.num_attention_heads
        self.head_dim = config.head_dim
        self.num_key_value_heads = config.num_key_value_heads
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=True)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=True)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=True)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=False)

    def forward(self, hidden_states, encoder_hidden_states, attention_mask=None,
                cross_attention_mask=None, **kwargs):
        if cross_attention_mask is not None:
            raise NotImplementedError("cross_attention_mask is never passed on the path")
        B, Lq, _ = hidden_states.shape
        Lk = encoder_hidden_states.shape[1]
        q = self.q_proj(hidden_states).view(B, Lq, self.num_heads, self.head_dim)
        k = self.k_proj(encoder_hidden_states).view(B, Lk, self.num_key_value_heads, self.head_dim)
        v = self.v_proj(encoder_hidden_states).view(B, Lk, self.num_key_value_heads, self.head_dim)
        o = K.flash_attn_func(q, k, v, causal=False)
        return self.o_proj(o.reshape(B, Lq, self.num_heads * self.head_dim)), None


Qwen2VLSdpaCrossAttention = Qwen2CrossAttention


class Qwen2DecoderLayer(nn.Module):
    """:268-318."""

    def __init__(self, config: Qwen2Config, layer_idx: int):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.self_attn = Qwen2Attention(config, layer_idx)
        self.mlp = Qwen2MLP(config)
        self.input_layernorm = Qwen2RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = Qwen2RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.attention_type = config.layer_types[layer_idx]

    def forward_fused(self, x, delta, position_embeddings, past_key_values=None, cache_position=None):
        """(stream, pending sub-layer output) -> the same pair after this layer: both residual
        adds ride inside the RMSNorm kernels."""
        if delta is None:
            h = self.input_layernorm(x)
        else:
            h, x = self.input_layernorm(x, residual=delta, return_sum=True)
        a, _ = self.self_attn(h, position_embeddings, past_key_values=past_key_values,
                              cache_position=cache_position)
        h, x = self.post_attention_layernorm(x, residual=a, return_sum=True)
        return x, self.mlp(h)

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None,
                use_cache=False, cache_position=None, position_embeddings=None, **kwargs):
        x, d = self.forward_fused(hidden_states, None, position_embeddings, past_key_values, cache_position)
        return x + d


class Qwen2Model(PdropMixin, nn.Module):
    """:388-1040 (inference paths)."""

    def __init__(self, config: Qwen2Config):
        super().__init__()
        self.config = config
        self.padding_idx, self.vocab_size = config.pad_token_id, config.vocab_size
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, self.padding_idx)
        self.layers = nn.ModuleList([Qwen2DecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = Qwen2RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.rotary_emb = Qwen2RotaryEmbedding(config)
        self.use_pdrop = getattr(config, "use_pdrop", False)
        self.pdrop_args: Dict[str, Any] = {"use_pdrop": self.use_pdrop}
        self.merge_modules, self.alpha, self.merge_ffn_modules, self.alpha_ffn = None, None, None, None
        if self.use_pdrop:
            assert config.pdrop_type is not None, "use_pdrop is True, but pdrop_type is not set"
            self.pdrop_types = [t.split("_") for t in config.pdrop_type.split("-")]
            assert all(len(t) == 3 for t in self.pdrop_types), \
                "pdrop_type should be like 'type_layernum_ratio-...' "
            self.pdrop_args.update({
                "pdrop_compress_types": [t[0] for t in self.pdrop_types],
                "pdrop_layers": [int(t[1]) for t in self.pdrop_types],
                "pdrop_ratios": [1] + [float(t[2]) for t in self.pdrop_types]})
            # the reference injects these later through set_pdrop_args (:1132-1136)
            self.pdrop_compress_types = self.pdrop_args["pdrop_compress_types"]
            self.pdrop_layers = self.pdrop_args["pdrop_layers"]
            self.pdrop_ratios = self.pdrop_args["pdrop_ratios"]
            if config.merge_module == "CrossAttention":
                self.merge_module_names, mods = [], []
                for i, _ in enumerate(self.pdrop_layers):
                    if "drop" in self.pdrop_compress_types[i]:
                        self.merge_module_names.append("none")
                        mods.append(nn.Identity())
                    else:
                        self.merge_module_names.append("attention")
                        mods.append(Qwen2CrossAttention(config, layer_idx=self.pdrop_layers[i]))
                self.merge_modules = nn.ModuleList(mods)
                self.alpha = nn.Parameter(torch.zeros(
                    sum(1 for m in self.merge_modules if not isinstance(m, nn.Identity))))
            elif config.merge_module != "no_merge":
                raise ValueError(f"Invalid merge module name for Qwen2: {config.merge_module}")
        self.last_pdrop_trace: List[Dict[str, torch.Tensor]] = []

    def _rank_attention(self, rank_layer):
        return self.layers[rank_layer].self_attn

    def get_input_embeddings(self):
        return self.embed_tokens

    def set_input_embeddings(self, value):
        self.embed_tokens = value

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels=None, use_cache=None, cache_position=None, **kwargs):
        if (input_ids is None) ^ (inputs_embeds is not None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if attention_mask is not None:
            raise NotImplementedError("only the mask-free path (batch 1, no padding) is implemented")
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(input_ids)
        use_cache = use_cache if use_cache is not None else self.config.use_cache
        if use_cache and past_key_values is None:
            past_key_values = Qwen2KVCache(self.config.num_hidden_layers)
        hidden = inputs_embeds
        if cache_position is None:
            seen = past_key_values.get_seq_length() if past_key_values is not None else 0
            cache_position = torch.arange(seen, seen + hidden.shape[1], device=hidden.device)
        if position_ids is None:
            position_ids = cache_position.unsqueeze(0).to(hidden.device)
        pos_emb = self.rotary_emb(hidden, position_ids)
        train_pdrop_args = kwargs.get("train_pdrop_args")
        self.last_pdrop_trace = []
        delta = None
        for layer_idx, layer in enumerate(self.layers):
            if self.use_pdrop and layer_idx in self.pdrop_layers:
                stage = self.pdrop_layers.index(layer_idx)
                if hidden.shape[1] != 1:                       # prefill (:918-966)
                    if train_pdrop_args is None:
                        raise ValueError("train_pdrop_args must be provided for pdrop during prefill/training.")
                    if delta is not None:
                        hidden, delta = hidden + delta, None
                    position_ids, attention_mask, hidden, labels, _ = self.flash_rank_drop(
                        cur_num=stage, rank_layer=layer_idx, features=hidden, position_ids=position_ids,
                        attention_mask=attention_mask, labels=labels, train_pdrop_args=train_pdrop_args)
                    pos_emb = self.rotary_emb(hidden, position_ids)
                else:                                          # decode: shift the position (:968-987)
                    nv = train_pdrop_args["num_vision_tokens"][0]
                    position_ids = position_ids - (int(nv * self.pdrop_ratios[stage])
                                                   - int(nv * self.pdrop_ratios[stage + 1]))
                    pos_emb = self.rotary_emb(hidden, position_ids)
            hidden, delta = layer.forward_fused(hidden, delta, pos_emb, past_key_values, cache_position)
        hidden = self.norm(hidden, residual=delta)
        return hidden, past_key_values if use_cache else None, labels


class Qwen2ForCausalLM(nn.Module):
    """:1042-1197."""

    def __init__(self, config: Qwen2Config):
        super().__init__()
        self.config = config
        self.model = Qwen2Model(config)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)

    @property
    def backbone(self):           # the name the hybrid family uses for the stack of layers
        return self.model

    @property
    def device(self):
        return self.lm_head.weight.device

    @property
    def dtype(self):
        return self.lm_head.weight.dtype

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def set_input_embeddings(self, value):
        self.model.embed_tokens = value

    def get_output_embeddings(self):
        return self.lm_head

    def new_cache(self, batch_size=1, dtype=None, device=None):
        return Qwen2KVCache(self.config.num_hidden_layers)

    def set_pdrop_args(self, **kwargs):
        for key, value in kwargs.items():
            setattr(self.model, key, value)

    def init_cross_attn_from_self_attn(self):
        if self.model.merge_modules is not None:
            for idx, module in enumerate(self.model.merge_modules):
                if not isinstance(module, nn.Identity):
                    module.load_state_dict(self.model.layers[self.model.pdrop_layers[idx]].self_attn.state_dict())

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels=None, use_cache=None, cache_position=None,
                logits_to_keep=0, **kwargs):
        hidden, cache, labels = self.model(input_ids=input_ids, attention_mask=attention_mask,
                                           position_ids=position_ids, past_key_values=past_key_values,
                                           inputs_embeds=inputs_embeds, labels=labels, use_cache=use_cache,
                                           cache_position=cache_position, **kwargs)
        if labels is not None:
            raise NotImplementedError("the loss is a training feature")
        # the reference's default logits_to_keep = 0 computes every position (:1105-1110); callers on
        # the evaluation path read the last one — None asks for exactly that
        if logits_to_keep is None:
            logits_to_keep = 1
        sl = slice(-logits_to_keep, None) if isinstance(logits_to_keep, int) else logits_to_keep
        logits = self.lm_head(hidden[:, sl, :])
        return CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=cache)
