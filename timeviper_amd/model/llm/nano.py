"""Host-side mirror of the reference's NemotronH hybrid LM
(`timeviper/model/llm/llm_repo/nano/{configuration_nano,modeling_nano}.py`): same
class names, module tree, parameter names / shapes (state-dict compatible,
SURVEY.md §8b) and forward() signatures — with every hot operator bound to the
gfx950 kernels in `timeviper_amd.kernels` instead of mamba_ssm / causal_conv1d /
flash_attn.  GEMMs stay on torch (hipBLASLt).  Inference (prefill + decode) only.

What is deliberately different from the reference (SURVEY.md §3.2 notes):
  * no per-layer `torch.isnan(...).any()` host sync (:1690) — `check_nan=True`
    restores it;
  * `lm_head` is applied to the positions asked for by `logits_to_keep`
    (default: last token) instead of all L positions (:2433);
  * pdrop "attn" ranking never builds the (L, L) mask (:1848-1857);
  * the residual add of layer i is fused into the RMSNorm of layer i+1.
"""
from __future__ import annotations

import math
import os
import re
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import kernels as K
from .pdrop import PdropMixin


# ------------------------------------------------------------------ config
class NemotronHConfig:
    """Field-for-field mirror of the reference's NemotronHConfig
    (configuration_nano.py:133-258)."""

    model_type = "nano"

    def __init__(self, vocab_size=131072, tie_word_embeddings=False, hidden_size=4096,
                 intermediate_size=21504, num_hidden_layers=52,
                 hybrid_override_pattern="M-M-M-M*-M-M-M-M-M*-M-M-M-M-M*-M-M-M-M-M*-M-M-M-M-M-",
                 num_attention_heads=32, head_dim=128, num_key_value_heads=8,
                 mlp_hidden_act="relu2", attention_bias=False, mlp_bias=False, use_bias=False,
                 initializer_range=0.02, layer_norm_epsilon=1e-5, residual_in_fp32=False,
                 use_cache=True, num_logits_to_keep=1, pad_token_id=0, bos_token_id=1,
                 eos_token_id=2, sliding_window=None, max_position_embeddings=4096,
                 attention_dropout=0.0, hidden_dropout=0.0, use_mamba_kernels=True,
                 ssm_state_size=128, mamba_num_heads=128, mamba_n_groups=8, mamba_head_dim=64,
                 mamba_d_conv=4, mamba_expand=2, mamba_hidden_act="silu", mamba_dt_min=0.001,
                 mamba_dt_max=0.1, mamba_dt_limit=(0.0, float("inf")), mamba_dt_init_floor=1e-4,
                 mamba_conv_bias=True, mamba_proj_bias=False, mamba_chunk_size=256,
                 rescale_prenorm_residual=True, merge_module="no_merge", use_pdrop=False,
                 pdrop_type=None, **kwargs):
        assert len(hybrid_override_pattern) == num_hidden_layers, \
            "hybrid_override_pattern must have the same length as num_hidden_layers"
        assert re.match(r"^[*\-M]+$", hybrid_override_pattern), \
            "hybrid_override_pattern must only contain characters 'M', '*', or '-'"
        self.vocab_size = vocab_size
        self.tie_word_embeddings = tie_word_embeddings
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.hybrid_override_pattern = hybrid_override_pattern
        self.num_attention_heads = num_attention_heads
        self.head_dim = head_dim
        self.sliding_window = sliding_window
        self.max_position_embeddings = max_position_embeddings
        self.attention_dropout = attention_dropout
        self.hidden_dropout = hidden_dropout
        self.num_key_value_heads = num_attention_heads if num_key_value_heads is None \
            else num_key_value_heads
        self.mlp_hidden_act = mlp_hidden_act
        self.attention_bias = attention_bias
        self.mlp_bias = mlp_bias
        self.use_bias = use_bias
        self.initializer_range = initializer_range
        self.layer_norm_epsilon = layer_norm_epsilon
        self.residual_in_fp32 = residual_in_fp32
        self.use_cache = use_cache
        self.num_logits_to_keep = num_logits_to_keep
        self.use_mamba_kernels = use_mamba_kernels
        self.n_groups = mamba_n_groups
        self.mamba_head_dim = mamba_head_dim
        self.ssm_state_size = ssm_state_size
        self.mamba_num_heads = mamba_num_heads
        self.conv_kernel = mamba_d_conv
        self.expand = mamba_expand
        self.mamba_hidden_act = mamba_hidden_act
        self.time_step_min = mamba_dt_min
        self.time_step_max = mamba_dt_max
        self.time_step_limit = tuple(mamba_dt_limit)
        self.time_step_floor = mamba_dt_init_floor
        self.use_conv_bias = mamba_conv_bias
        self.mamba_proj_bias = mamba_proj_bias
        self.chunk_size = mamba_chunk_size
        self.rescale_prenorm_residual = rescale_prenorm_residual
        self.merge_module = merge_module
        self.use_pdrop = use_pdrop
        self.pdrop_type = pdrop_type
        self.pad_token_id, self.bos_token_id, self.eos_token_id = \
            pad_token_id, bos_token_id, eos_token_id
        self._attn_implementation = kwargs.pop("attn_implementation", "flash_attention_2")
        self.output_attentions = False
        self.output_hidden_states = False
        self.use_return_dict = True
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def layers_block_type(self):
        return ["mamba" if c == "M" else "attention" if c == "*" else "mlp"
                for c in self.hybrid_override_pattern]

    @classmethod
    def nemotron_nano_9b_v2(cls, **over):
        """NVIDIA-Nemotron-Nano-9B-v2 dims (public config.json)."""
        kw = dict(vocab_size=131072, hidden_size=4480, intermediate_size=15680,
                  num_hidden_layers=56,
                  hybrid_override_pattern="M-M-M-MM-M-M-M*-M-M-M*-M-M-M-M*-M-M-M-M*-M-MM-M-M-M-M-M-",
                  num_attention_heads=40, head_dim=128, num_key_value_heads=8,
                  ssm_state_size=128, mamba_num_heads=128, mamba_n_groups=8, mamba_head_dim=80,
                  mamba_d_conv=4, mamba_chunk_size=128, layer_norm_epsilon=1e-5,
                  max_position_embeddings=131072)
        kw.update(over)
        return cls(**kw)


@dataclass
class BaseModelOutputWithPastAndLabels:
    last_hidden_state: Optional[torch.Tensor] = None
    past_key_values: Any = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Optional[Tuple[torch.Tensor, ...]] = None
    labels: Optional[torch.Tensor] = None

    def __getitem__(self, i):
        return [v for v in (self.last_hidden_state, self.past_key_values, self.hidden_states,
                            self.labels) if v is not None][i]


@dataclass
class CausalLMOutputWithPast:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Any = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Optional[Tuple[torch.Tensor, ...]] = None


# ------------------------------------------------------------------- cache
class HybridMambaAttentionDynamicCache:
    """Per-layer conv / ssm / key / value lists (reference :205-360).  SSM states
    are kept in fp32 (what the scan kernel returns and the decode kernel updates).

    Keys / values: `key_cache[i]` is what the reference holds — (B, L_i, Hkv, D) with every token seen so far — but it is a
    VIEW of a buffer with spare capacity: the reference's `torch.cat` per generated token (:246-251) re-copies the whole
    cache (2 x 135 MB per attention layer and token behind a 2 048-frame prefill); here a token is written in place and the
    buffer grows geometrically when it runs out.

    `begin_static_decode(n)` fixes the buffers for the next n tokens and moves the write position and the key count to
    DEVICE memory: every launch of a decode step then has the same parameters, which is what
    lets `GraphedDecodeStep` replay it as one hipGraph."""

    def __init__(self, config, batch_size, dtype=torch.bfloat16, device=None, kv_reserve=256):
        self.dtype = dtype
        self.hybrid_override_pattern = config.hybrid_override_pattern
        self.has_previous_state = False
        self.conv_kernel_size = config.conv_kernel
        n = config.num_hidden_layers
        empty = lambda: torch.empty((batch_size, 0), device=device)
        self.conv_states: List[torch.Tensor] = [empty() for _ in range(n)]
        self.ssm_states: List[torch.Tensor] = [empty() for _ in range(n)]
        self.key_cache: List[torch.Tensor] = [empty() for _ in range(n)]
        self.value_cache: List[torch.Tensor] = [empty() for _ in range(n)]
        self.transformer_layers = [i for i, c in enumerate(config.hybrid_override_pattern)
                                   if c != "M"]
        self.attention_layers = [i for i, c in enumerate(config.hybrid_override_pattern)
                                 if c == "*"]
        self.kv_reserve = int(kv_reserve)
        self._kv_buf = [None] * n              # (k, v) capacity buffers, (B, cap, Hkv, D)
        self._kv_len = [0] * n
        # static decode: per distinct cache length an int64 (1,) write slot and an int32 (B,) key count, on the device
        self._static, self._static_of, self.static_decode = {}, {}, False

    def _reserve(self, layer_idx, need):
        have = self._kv_len[layer_idx]
        buf = self._kv_buf[layer_idx]
        if buf is not None and buf[0].shape[1] >= need:
            return
        if self.static_decode:
            raise RuntimeError("static decode: the key / value buffers are fixed (begin_static_decode reserved too little)")
        k_old, v_old = self.key_cache[layer_idx], self.value_cache[layer_idx]
        shape = (k_old.shape[0], need + max(self.kv_reserve, have // 8)) + tuple(k_old.shape[2:])
        kb, vb = k_old.new_empty(shape), v_old.new_empty(shape)
        kb[:, :have], vb[:, :have] = k_old, v_old
        self._kv_buf[layer_idx] = (kb, vb)
        self.key_cache[layer_idx], self.value_cache[layer_idx] = kb[:, :have], vb[:, :have]

    def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
        """key/value (B, L, Hkv, D) — sequence-major, the layout the attention kernel reads."""
        new = key_states.shape[1]
        if self.key_cache[layer_idx].shape[-1] == 0:        # prefill: adopt the projections' output, no copy
            self.key_cache[layer_idx], self.value_cache[layer_idx] = key_states, value_states
            self._kv_buf[layer_idx], self._kv_len[layer_idx] = None, new
            return key_states, value_states
        have = self._kv_len[layer_idx]
        self._reserve(layer_idx, have + new)
        kb, vb = self._kv_buf[layer_idx]
        if self.static_decode:
            if new != 1:
                raise RuntimeError("static decode takes one token per step")
            pos = self._static_of[layer_idx][0]
            kb.index_copy_(1, pos, key_states)
            vb.index_copy_(1, pos, value_states)
        else:
            kb[:, have:have + new], vb[:, have:have + new] = key_states, value_states
        self._kv_len[layer_idx] = have + new
        self.key_cache[layer_idx], self.value_cache[layer_idx] = kb[:, :have + new], vb[:, :have + new]
        return self.key_cache[layer_idx], self.value_cache[layer_idx]

    def kv_buffers(self, layer_idx):
        """The capacity buffers of a layer (static decode: what the attention kernel is handed, with `decode_lens`)."""
        return self._kv_buf[layer_idx]

    # ---- static decode (fixed launch parameters; see GraphedDecodeStep)
    def begin_static_decode(self, max_new_tokens: int):
        """Token drop (pdrop) leaves the attention layers behind a stage with fewer tokens than the ones in front of it:
        one (write position, key count) pair per distinct length, shared by the layers that hold it."""
        if not self.attention_layers or any(self._kv_len[i] == 0 for i in self.attention_layers):
            raise RuntimeError("static decode starts behind a prefill (every attention layer holding its tokens)")
        self.end_static_decode()
        for i in self.attention_layers:
            self._reserve(i, self._kv_len[i] + int(max_new_tokens))
        k0 = self.key_cache[self.attention_layers[0]]
        self._static = {}
        for n in sorted({self._kv_len[i] for i in self.attention_layers}):
            self._static[n] = (torch.full((1,), n, dtype=torch.int64, device=k0.device),
                               torch.full((k0.shape[0],), n + 1, dtype=torch.int32, device=k0.device))
        self._static_of = {i: self._static[self._kv_len[i]] for i in self.attention_layers}
        self.static_decode = True

    def end_static_decode(self):
        self._static, self._static_of, self.static_decode = {}, {}, False

    def static_room(self) -> int:
        """Tokens the fixed buffers still take (a replayed graph writes where its device-side position says: the caller
        must stop before the buffers end — GraphedDecodeStep checks this before every launch)."""
        return min(self._kv_buf[i][0].shape[1] - self._kv_len[i] for i in self.attention_layers)

    def advance_static_device(self):
        """Device side of one finished step (part of the captured graph)."""
        for pos, lens in self._static.values():
            pos += 1
            lens += 1

    def decode_lens(self, layer_idx):
        return self._static_of[layer_idx][1]

    def advance_static_host(self):
        """Host side of one REPLAYED step: the bookkeeping `update` does when it runs."""
        for i in self.attention_layers:
            n = self._kv_len[i] + 1
            kb, vb = self._kv_buf[i]
            self._kv_len[i] = n
            self.key_cache[i], self.value_cache[i] = kb[:, :n], vb[:, :n]

    def get_seq_length(self, layer_idx: Optional[int] = 0) -> int:
        if not self.attention_layers:
            return 0
        k = self.key_cache[self.attention_layers[0]]
        return 0 if k.shape[-1] == 0 else k.shape[1]

    def update_conv_state(self, layer_idx, new_conv_state, cache_init=False):
        assert cache_init
        self.conv_states[layer_idx] = new_conv_state
        return new_conv_state

    def update_ssm_state(self, layer_idx, new_ssm_state):
        self.ssm_states[layer_idx] = new_ssm_state
        return new_ssm_state


# ------------------------------------------------------------------ decode-step helpers
def _fused_decode(x: torch.Tensor) -> bool:
    """The linear layers of a decode step (1..4 rows) run on tv_gemv_bf16_fwd with the operator in front of them in
    the prologue; TV_DECODE_FUSED=0 keeps torch.nn.Linear + the stand-alone kernels (same arithmetic, library GEMV)."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 3 and x.shape[1] == 1 and x.shape[0] <= 4
            and not torch.is_grad_enabled() and os.environ.get("TV_DECODE_FUSED", "1") != "0")


def _linear(mod: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    if _fused_decode(x) and K.gemv_takes(x, mod.weight):
        return K.gemv_fused(x, mod.weight, mod.bias)
    return mod(x)


# ------------------------------------------------------------------ layers
class MambaRMSNormGated(nn.Module):
    def __init__(self, hidden_size, group_size, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps
        self.group_size = group_size

    def forward(self, hidden_states, gate=None):
        return K.rmsnorm_fn(x=hidden_states, weight=self.weight, bias=None, z=gate,
                            eps=self.variance_epsilon, group_size=self.group_size,
                            norm_before_gate=False)


class NemotronHRMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states, residual=None, return_sum=False):
        return K.rms_norm(hidden_states, self.weight, self.variance_epsilon, residual=residual,
                          return_sum=return_sum)


class NemotronHMamba2Mixer(nn.Module):
    """Reference :383-885.  forward() == cuda_kernels_forward's inference branches."""

    def __init__(self, config: NemotronHConfig, layer_idx: int):
        super().__init__()
        self.num_heads = config.mamba_num_heads
        self.hidden_size = config.hidden_size
        self.ssm_state_size = config.ssm_state_size
        self.conv_kernel_size = config.conv_kernel
        self.intermediate_size = config.mamba_num_heads * config.mamba_head_dim
        self.layer_idx = layer_idx
        self.use_conv_bias = config.use_conv_bias
        self.activation = config.mamba_hidden_act
        self.layer_norm_epsilon = config.layer_norm_epsilon
        self.n_groups = config.n_groups
        self.head_dim = config.mamba_head_dim
        self.chunk_size = config.chunk_size
        self.time_step_limit = tuple(config.time_step_limit)
        self.conv_dim = self.intermediate_size + 2 * self.n_groups * self.ssm_state_size
        self.conv1d = nn.Conv1d(self.conv_dim, self.conv_dim, bias=config.use_conv_bias,
                                kernel_size=config.conv_kernel, groups=self.conv_dim,
                                padding=config.conv_kernel - 1)
        projection_size = self.intermediate_size + self.conv_dim + self.num_heads
        self.in_proj = nn.Linear(self.hidden_size, projection_size, bias=config.use_bias)
        self.dt_bias = nn.Parameter(torch.ones(self.num_heads))
        self.A_log = nn.Parameter(torch.log(torch.arange(1, self.num_heads + 1).float()))
        self.norm = MambaRMSNormGated(self.intermediate_size, eps=self.layer_norm_epsilon,
                                      group_size=self.intermediate_size // self.n_groups)
        self.D = nn.Parameter(torch.ones(self.num_heads))
        self.out_proj = nn.Linear(self.intermediate_size, self.hidden_size, bias=config.use_bias)
        self.use_bias = config.use_bias
        self.group_map = "block"  # "tile" reproduces the reference CPU quirk (tests only)

    def _consts(self):
        """(-exp(A_log), D, dt_bias) in fp32 — what the scan kernels take (modeling_nano.py:640, :514-522).
        Derived once per parameter version, not once per call: in a bf16 model the per-call form costs five
        small launches a layer (float, exp, neg, two casts), 27 layers a forward, every decode step."""
        ps = (self.A_log, self.D, self.dt_bias)
        key = K.param_key(ps)
        if key != getattr(self, "_ckey", None):
            with torch.no_grad():
                self._cval = (-torch.exp(self.A_log.detach().float()), self.D.detach().float().contiguous(),
                              self.dt_bias.detach().float().contiguous())
            self._ckey = key
        return self._cval

    def _neg_A(self):
        return self._consts()[0]

    def forward(self, hidden_states, cache_params: Optional[HybridMambaAttentionDynamicCache] = None,
                cache_position=None, attention_mask=None, seq_idx=None,
                initial_states=None, conv_halo=None, return_shard_state=False):
        if seq_idx is not None:
            raise NotImplementedError("sequence packing (seq_idx) is a training feature")
        batch_size, seq_len, _ = hidden_states.shape
        if attention_mask is not None and attention_mask.shape[1] > 1 and attention_mask.shape[0] > 1:
            hidden_states = (hidden_states * attention_mask[:, :, None]).to(hidden_states.dtype)
        projected_states = self.in_proj(hidden_states)
        gts = self.n_groups * self.ssm_state_size
        d_in = self.intermediate_size
        decode = (cache_params is not None and cache_position is not None
                  and int(cache_position[0]) > 0)
        gate, xBC, dt = projected_states.split([d_in, self.conv_dim, self.num_heads], dim=-1)
        w = self.conv1d.weight.squeeze(1)
        negA, D32, dtb32 = self._consts()
        if decode:
            assert seq_len == 1
            xBC = K.causal_conv1d_update(xBC[:, 0], cache_params.conv_states[self.layer_idx], w,
                                         self.conv1d.bias, self.activation)
            x, Bm, Cm = torch.split(xBC, [d_in, gts, gts], dim=-1)
            y = K.selective_state_update(
                cache_params.ssm_states[self.layer_idx],
                x.reshape(batch_size, self.num_heads, self.head_dim), dt[:, 0], negA,
                Bm.reshape(batch_size, self.n_groups, -1), Cm.reshape(batch_size, self.n_groups, -1),
                D32, z=None, dt_bias=dtb32, dt_softplus=True)
            y = self.norm(y.reshape(batch_size, 1, d_in), gate)
            return self.out_proj(y)

        if cache_params is not None:  # conv state = last K pre-conv inputs, (B, C, K)  (:596-610)
            Kw = self.conv_kernel_size
            xt = xBC.transpose(1, 2)
            cs = F.pad(xt, (Kw - seq_len, 0)) if seq_len < Kw else xt[..., -Kw:]
            cache_params.update_conv_state(self.layer_idx, cs.contiguous(), cache_init=True)
        # causal_conv1d_fn (:619-624) + the x|B|C split (:628-636) in one pass; B and C come
        # back as (B, L, G, N) views of group-major storage
        # (+ the scan's causal C.B^T fragments, multiplied while the B / C tiles are on the chip)
        x, Bm, Cm, cb = K.causal_conv1d_xbc(xBC, w, self.conv1d.bias, d_in, self.n_groups,
                                            self.ssm_state_size, activation=self.activation,
                                            halo=conv_halo, return_cb=True)
        dt_limit = {} if self.time_step_limit == (0.0, float("inf")) \
            else {"dt_limit": self.time_step_limit}
        res = K.mamba_chunk_scan_combined(
            x.view(batch_size, seq_len, -1, self.head_dim), dt, negA, Bm, Cm,
            chunk_size=self.chunk_size, D=D32,
            z=None, seq_idx=None, return_final_states=True, dt_bias=dtb32, dt_softplus=True,
            initial_states=initial_states, group_map=self.group_map,
            return_total_decay=return_shard_state, cb=cb, **dt_limit)
        scan_output, ssm_state = res[0], res[1]
        if cache_params is not None:
            cache_params.update_ssm_state(self.layer_idx, ssm_state)
        scan_output = self.norm(scan_output.view(batch_size, seq_len, -1), gate)
        out = self.out_proj(scan_output)
        if return_shard_state:
            return out, ssm_state, res[2]
        return out


    def decode_fused(self, hidden, delta, norm, cache_params):
        """One decode token with the block's RMSNorm (+ residual add) inside in_proj and the gated norm inside out_proj:
        in_proj (+ conv update in its epilogue) -> state update -> out_proj, 3 launches.  Returns (residual stream, mixer output)."""
        B = hidden.shape[0]
        new_hidden = torch.empty_like(hidden) if delta is not None else hidden
        gts = self.n_groups * self.ssm_state_size
        d_in = self.intermediate_size
        conv_state = cache_params.conv_states[self.layer_idx]
        w = self.conv1d.weight.squeeze(1)
        # the conv update rides in the product's epilogue where the kernel offers it (width 4, SiLU, K < 8192)
        conv_in = (self.conv_kernel_size == 4 and self.activation in ("silu", "swish") and hidden.shape[-1] < 8192
                   and conv_state.dtype == hidden.dtype and conv_state.is_contiguous() and w.is_contiguous())
        proj = K.gemv_fused(hidden, self.in_proj.weight, self.in_proj.bias, K.GEMV_RMSNORM, delta=delta,
                            sum_out=new_hidden if delta is not None else None, norm_weight=norm.weight,
                            eps=norm.variance_epsilon,
                            conv=(conv_state, w, self.conv1d.bias, d_in) if conv_in else None)
        gate, xBC, dt = proj.split([d_in, self.conv_dim, self.num_heads], dim=-1)
        negA, D32, dtb32 = self._consts()
        xBC = xBC[:, 0] if conv_in else K.causal_conv1d_update(xBC[:, 0], conv_state, w, self.conv1d.bias, self.activation)
        x, Bm, Cm = torch.split(xBC, [d_in, gts, gts], dim=-1)
        y = K.selective_state_update(
            cache_params.ssm_states[self.layer_idx], x.reshape(B, self.num_heads, self.head_dim), dt[:, 0], negA,
            Bm.reshape(B, self.n_groups, -1), Cm.reshape(B, self.n_groups, -1), D32, z=None, dt_bias=dtb32,
            dt_softplus=True)
        out = K.gemv_fused(y.reshape(B, 1, d_in), self.out_proj.weight, self.out_proj.bias, K.GEMV_GATED,
                           norm_weight=self.norm.weight, eps=self.norm.variance_epsilon, gate=gate,
                           group_size=self.norm.group_size)
        return new_hidden, out


class ReLUSquared(nn.Module):
    def forward(self, x):
        return torch.square(F.relu(x))


class NemotronHMLP(nn.Module):
    def __init__(self, config, layer_idx: Optional[int] = None):
        super().__init__()
        self.config, self.layer_idx = config, layer_idx
        self.hidden_size, self.intermediate_size = config.hidden_size, config.intermediate_size
        self.up_proj = nn.Linear(self.hidden_size, self.intermediate_size, bias=config.mlp_bias)
        self.down_proj = nn.Linear(self.intermediate_size, self.hidden_size, bias=config.mlp_bias)
        assert config.mlp_hidden_act == "relu2"
        self.act_fn = ReLUSquared()

    def forward(self, x):
        # `act_fn` stays for the module tree; the activation runs as one in-place HIP pass
        return self.down_proj(K.relu2(self.up_proj(x), inplace=True))

    def decode_fused(self, hidden, delta, norm):
        """One decode token: RMSNorm (+ residual add) inside up_proj, relu^2 inside down_proj — 2 launches."""
        new_hidden = torch.empty_like(hidden) if delta is not None else hidden
        u = K.gemv_fused(hidden, self.up_proj.weight, self.up_proj.bias, K.GEMV_RMSNORM, delta=delta,
                         sum_out=new_hidden if delta is not None else None, norm_weight=norm.weight,
                         eps=norm.variance_epsilon)
        return new_hidden, K.gemv_fused(u, self.down_proj.weight, self.down_proj.bias, K.GEMV_RELU2)


class NemotronHAttention(nn.Module):
    """Causal GQA attention without positional encoding (reference :1012-1220)."""

    def __init__(self, config: NemotronHConfig, layer_idx: Optional[int] = None):
        super().__init__()
        self.config, self.layer_idx = config, layer_idx
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = config.head_dim if config.head_dim is not None \
            else config.hidden_size // config.num_attention_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.is_causal = True
        b = config.attention_bias
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=b)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=b)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=b)
        self.o_proj = nn.Linear(self.head_dim * self.num_heads, self.hidden_size, bias=b)

    def forward(self, hidden_states, attention_mask=None, position_ids=None,
                past_key_value: Optional[HybridMambaAttentionDynamicCache] = None,
                output_attentions=False, use_cache=False, cache_position=None, **kwargs):
        if attention_mask is not None:
            raise NotImplementedError("padding masks are not on the inference path (mask is None "
                                      "under flash_attention_2, reference :2208-2213)")
        bsz, q_len, _ = hidden_states.size()
        q = _linear(self.q_proj, hidden_states).view(bsz, q_len, self.num_heads, self.head_dim)
        k = _linear(self.k_proj, hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim)
        v = _linear(self.v_proj, hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim)
        if past_key_value is not None:
            k, v = past_key_value.update(k, v, self.layer_idx)
        if past_key_value is not None and q_len == 1 and past_key_value.static_decode:
            # static decode: the whole buffers + the key count on the device (same launch parameters every token)
            kb, vb = past_key_value.kv_buffers(self.layer_idx)
            o = K.flash_attn_decode(q, kb, vb, seqlens_k=past_key_value.decode_lens(self.layer_idx))
        else:
            o = K._flash_attention_forward(q, k, v, attention_mask=None, query_length=q_len,
                                           is_causal=self.is_causal)
        o = _linear(self.o_proj, o.reshape(bsz, q_len, self.num_heads * self.head_dim))
        return o, None, past_key_value


    def _qkv_weight(self):
        """[q; k; v] as one (N, K) matrix for the decode token's single product (a copy: 64 MB per layer at 9B dims), rebuilt
        when a projection's storage or version changes."""
        ws = (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight)
        key = K.param_key(ws)
        if getattr(self, "_qkv_key", None) != key:
            self._qkv_cat = torch.cat([w.detach() for w in ws], dim=0).contiguous()
            self._qkv_key = key
        return self._qkv_cat

    def decode_fused(self, hidden, delta, norm, cache):
        """One decode token: the block's RMSNorm (+ residual add) in the prologue of ONE q / k / v product, cache append,
        split-KV attention, o_proj — 4 launches + the cache writes, against 7."""
        B = hidden.shape[0]
        new_hidden = torch.empty_like(hidden) if delta is not None else hidden
        nq, nkv = self.num_heads * self.head_dim, self.num_key_value_heads * self.head_dim
        qkv = K.gemv_fused(hidden, self._qkv_weight(), None, K.GEMV_RMSNORM, delta=delta,
                           sum_out=new_hidden if delta is not None else None, norm_weight=norm.weight,
                           eps=norm.variance_epsilon)
        q, k, v = qkv.split([nq, nkv, nkv], dim=-1)
        q = q.view(B, 1, self.num_heads, self.head_dim)
        k = k.view(B, 1, self.num_key_value_heads, self.head_dim)
        v = v.view(B, 1, self.num_key_value_heads, self.head_dim)
        kc, vc = cache.update(k, v, self.layer_idx)
        if cache.static_decode:
            kb, vb = cache.kv_buffers(self.layer_idx)
            o = K.flash_attn_decode(q, kb, vb, seqlens_k=cache.decode_lens(self.layer_idx))
        else:
            o = K._flash_attention_forward(q, kc, vc, attention_mask=None, query_length=1, is_causal=self.is_causal)
        return new_hidden, _linear(self.o_proj, o.reshape(B, 1, nq))


NemotronHFlashAttention2 = NemotronHAttention
NemotronHSdpaAttention = NemotronHAttention
NEMOTRONH_ATTENTION_CLASSES = {"eager": NemotronHAttention,
                               "flash_attention_2": NemotronHFlashAttention2,
                               "sdpa": NemotronHSdpaAttention}


class Qwen2VLCrossAttention(nn.Module):
    """TransV merge module (reference merge_modules/cross_attention.py:65-324):
    text tokens (Q) attend non-causally to the dropped vision tokens (K, V)."""

    def __init__(self, config, layer_idx: Optional[int] = None):
        super().__init__()
        self.config, self.layer_idx = config, layer_idx
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = config.head_dim if config.head_dim is not None \
            else config.hidden_size // config.num_attention_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.is_causal = False
        b = config.attention_bias
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=b)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=b)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=b)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=b)

    def forward(self, hidden_states, encoder_hidden_states, attention_mask=None,
                cross_attention_mask=None, **kwargs):
        if cross_attention_mask is not None:
            raise NotImplementedError("cross_attention_mask is never passed on the path (:1761-1765)")
        bsz, q_len, _ = hidden_states.size()
        kv_len = encoder_hidden_states.size(1)
        q = self.q_proj(hidden_states).view(bsz, q_len, self.num_heads, self.head_dim)
        k = self.k_proj(encoder_hidden_states).view(bsz, kv_len, self.num_key_value_heads, self.head_dim)
        v = self.v_proj(encoder_hidden_states).view(bsz, kv_len, self.num_key_value_heads, self.head_dim)
        o = K.flash_attn_func(q, k, v, causal=False)
        return self.o_proj(o.reshape(bsz, q_len, self.num_heads * self.head_dim)), None


Qwen2VLSdpaCrossAttention = Qwen2VLCrossAttention


class NemotronHBlock(nn.Module):
    def __init__(self, config, layer_idx):
        super().__init__()
        self.config, self.layer_idx = config, layer_idx
        self.residual_in_fp32 = config.residual_in_fp32
        if self.residual_in_fp32:
            # the reference upcasts the residual stream to fp32 (modeling_nano.py:942-943); the fused
            # residual-add + RMSNorm path here carries it in the activation dtype, so a config that
            # asks for fp32 residuals must not run silently with different arithmetic
            raise NotImplementedError("residual_in_fp32=True is not implemented (the checkpoints on the "
                                      "evaluation path, Nemotron-Nano-9B-v2, use False)")
        self.norm = NemotronHRMSNorm(config.hidden_size, eps=config.layer_norm_epsilon)
        self.block_type = config.layers_block_type[layer_idx]
        if self.block_type == "mamba":
            self.mixer = NemotronHMamba2Mixer(config, layer_idx=layer_idx)
        elif self.block_type == "attention":
            self.mixer = NEMOTRONH_ATTENTION_CLASSES[config._attn_implementation](config, layer_idx=layer_idx)
        elif self.block_type == "mlp":
            self.mixer = NemotronHMLP(config, layer_idx=layer_idx)
        else:
            raise ValueError(f"Invalid layer pattern {config.hybrid_override_pattern[layer_idx]}")

    def mix(self, normed, cache_params=None, cache_position=None, attention_mask=None,
            position_ids=None):
        if self.block_type == "mamba":
            return self.mixer(normed, cache_params=cache_params, cache_position=cache_position)
        if self.block_type == "attention":
            return self.mixer(normed, past_key_value=cache_params, cache_position=cache_position,
                              attention_mask=attention_mask, position_ids=position_ids)[0]
        return self.mixer(normed)

    def forward(self, hidden_states, cache_params=None, cache_position=None, attention_mask=None,
                seq_idx=None, position_ids=None):
        residual = hidden_states
        normed = self.norm(hidden_states.to(dtype=self.norm.weight.dtype))
        out = self.mix(normed, cache_params, cache_position, attention_mask, position_ids)
        return residual + out


# --------------------------------------------------------------------- model
class NemotronHModel(PdropMixin, nn.Module):
    """Reference :1449-2273 (inference paths)."""

    def __init__(self, config: NemotronHConfig):
        super().__init__()
        self.config = config
        self.embeddings = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([NemotronHBlock(config, layer_idx=i)
                                     for i in range(config.num_hidden_layers)])
        self.use_pdrop = config.use_pdrop
        self.pdrop_args: Dict[str, Any] = {"use_pdrop": config.use_pdrop}
        self.pdrop_types = None
        if self.use_pdrop:
            assert config.pdrop_type is not None, "use_pdrop is True, but pdrop_type is not set"
            self.pdrop_types = [t.split("_") for t in config.pdrop_type.split("-")]
            assert all(len(t) == 3 for t in self.pdrop_types), \
                "pdrop_type should be like 'type_layernum_ratio-...' "
            self.pdrop_args.update({
                "pdrop_compress_types": [t[0] for t in self.pdrop_types],
                "pdrop_layers": [int(t[1]) for t in self.pdrop_types],
                "pdrop_ratios": [1] + [float(t[2]) for t in self.pdrop_types]})
            # the reference injects these later through set_pdrop_args (:2459-2462)
            self.pdrop_compress_types = self.pdrop_args["pdrop_compress_types"]
            self.pdrop_layers = self.pdrop_args["pdrop_layers"]
            self.pdrop_ratios = self.pdrop_args["pdrop_ratios"]
        if config.merge_module == "CrossAttention":
            self.merge_module_names, mods = [], []
            for i, _ in enumerate(self.pdrop_args.get("pdrop_layers", [])):
                if "drop" in self.pdrop_args["pdrop_compress_types"][i]:
                    self.merge_module_names.append("none")
                    mods.append(nn.Identity())
                else:
                    self.merge_module_names.append("attention")
                    mods.append(Qwen2VLSdpaCrossAttention(
                        config, layer_idx=self.pdrop_args["pdrop_layers"][i]))
            self.merge_modules = nn.ModuleList(mods)
            self.alpha = nn.Parameter(torch.zeros(
                sum(1 for m in self.merge_modules if not isinstance(m, nn.Identity))))
        elif config.merge_module == "no_merge":
            self.merge_modules, self.alpha = None, None
        else:
            raise ValueError(f"Invalid merge module name: {config.merge_module}")
        self.merge_ffn_modules, self.alpha_ffn = None, None
        self.norm_f = NemotronHRMSNorm(config.hidden_size, eps=config.layer_norm_epsilon)
        self.check_nan = False
        self.last_pdrop_trace: List[Dict[str, torch.Tensor]] = []
        self._register_load_state_dict_pre_hook(self.load_hook)

    @staticmethod
    def load_hook(state_dict, prefix, *args):
        for k in list(state_dict):
            if "embedding." in k:
                state_dict[k.replace("embedding.", "embeddings.")] = state_dict.pop(k)
                break

    def get_input_embeddings(self):
        return self.embeddings

    def set_input_embeddings(self, new_embeddings):
        self.embeddings = new_embeddings

    def _all_matrices_bf16(self) -> bool:
        """Every 2-D parameter of every block is dense bf16 — what the fused decode step's matrix-vector kernel takes (its
        C ABI carries no weight dtype).  Decided once per parameter set (`_apply` — .to() / .bfloat16() — and
        load_state_dict replace or rewrite the parameters: both reset it)."""
        ok = getattr(self, "_bf16_ok", None)
        if ok is None:
            ok = all(p.dtype == torch.bfloat16 and p.stride(-1) == 1
                     for blk in self.layers for p in blk.mixer.parameters() if p.dim() == 2)
            self._bf16_ok = ok
        return ok

    def _apply(self, fn, *a, **kw):
        self._bf16_ok = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._bf16_ok = None
        return super().load_state_dict(*a, **kw)

    def _rank_attention(self, rank_layer):
        """the self-attention module whose q/k projections rank the vision tokens (:1822-1830)"""
        assert self.layers[rank_layer].block_type == "attention"
        return self.layers[rank_layer].mixer

    def forward(self, input_ids=None, inputs_embeds=None, position_ids=None, past_key_values=None,
                use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, cache_position=None, attention_mask=None, **kwargs):
        if (input_ids is None) ^ (inputs_embeds is not None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if inputs_embeds is None:
            inputs_embeds = self.embeddings(input_ids)
        if attention_mask is not None:
            if self.config._attn_implementation == "flash_attention_2":
                raise ValueError("attention_mask must be None if using flash_attention_2")
            raise NotImplementedError("only the mask-free flash_attention_2 path is implemented")
        use_cache = use_cache if use_cache is not None else self.config.use_cache
        hidden = inputs_embeds
        if cache_position is None:
            cache_position = torch.arange(hidden.shape[1], device=hidden.device)
        if position_ids is None:
            position_ids = cache_position.unsqueeze(0)
        labels = kwargs.get("labels", None)
        all_hidden = () if output_hidden_states else None
        train_pdrop_args = kwargs.get("train_pdrop_args")
        self.last_pdrop_trace = []
        delta = None  # pending mixer output, added inside the next fused norm
        fused_step = (_fused_decode(hidden) and past_key_values is not None and cache_position is not None
                      and int(cache_position[0]) > 0 and not output_hidden_states and not self.check_nan
                      and hidden.shape[-1] % 8 == 0 and hidden.shape[-1] <= 8192
                      and self._all_matrices_bf16())
        # attention blocks join when their projections carry no bias (the stacked q / k / v product takes none)
        attn_fused = fused_step and not self.config.attention_bias
        for layer_idx, block in enumerate(self.layers):
            if self.use_pdrop and train_pdrop_args is not None \
                    and layer_idx in self.pdrop_layers \
                    and train_pdrop_args.get("is_interleaved", False) is False:
                stage = self.pdrop_layers.index(layer_idx)
                if hidden.shape[1] != 1:
                    if delta is not None:
                        hidden, delta = hidden + delta, None
                    position_ids, attention_mask, hidden, labels, _ = self.flash_rank_drop(
                        cur_num=stage, rank_layer=layer_idx, features=hidden,
                        position_ids=position_ids, attention_mask=attention_mask, labels=labels,
                        train_pdrop_args=train_pdrop_args)
                else:  # decode: shift positions by the tokens dropped at this stage (:1666-1689)
                    nv = train_pdrop_args["num_vision_tokens"][0]
                    position_ids = position_ids - (int(nv * self.pdrop_ratios[stage])
                                                   - int(nv * self.pdrop_ratios[stage + 1]))
            if self.check_nan and torch.isnan(hidden).any():
                raise ValueError("NaN detected in hidden_states before mixer block")
            if output_hidden_states:
                all_hidden += ((hidden if delta is None else hidden + delta),)
            # x_{i+1} = x_i + mixer(norm(x_i)); the add of layer i-1 is fused into this norm
            if fused_step and (block.block_type != "attention" or attn_fused):
                # decode token: the norm (+ add) runs in the prologue of the block's first matrix-vector product
                if block.block_type == "mlp":
                    hidden, delta = block.mixer.decode_fused(hidden, delta, block.norm)
                else:
                    hidden, delta = block.mixer.decode_fused(hidden, delta, block.norm, past_key_values)
                continue
            if delta is None:
                normed = block.norm(hidden)
            else:
                normed, hidden = block.norm(hidden, residual=delta, return_sum=True)
            delta = block.mix(normed, past_key_values, cache_position, None, position_ids)
        hidden = self.norm_f(hidden, residual=delta) if delta is not None else self.norm_f(hidden)
        if output_hidden_states:
            all_hidden += (hidden,)
        return BaseModelOutputWithPastAndLabels(
            last_hidden_state=hidden, past_key_values=past_key_values if use_cache else None,
            hidden_states=all_hidden, attentions=None, labels=labels)


class NemotronHForCausalLM(nn.Module):
    """Reference :2283-2504."""

    def __init__(self, config: NemotronHConfig):
        super().__init__()
        self.config = config
        self.backbone = NemotronHModel(config)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.apply(self._init_weights)

    def _init_weights(self, module):
        """Reference :1339-1383."""
        cfg = self.config
        if isinstance(module, NemotronHMamba2Mixer):
            dt = torch.exp(torch.rand(cfg.mamba_num_heads)
                           * (math.log(cfg.time_step_max) - math.log(cfg.time_step_min))
                           + math.log(cfg.time_step_min)).clamp(min=cfg.time_step_floor)
            with torch.no_grad():
                module.dt_bias.copy_(dt + torch.log(-torch.expm1(-dt)))
        if isinstance(module, nn.Linear) and module.bias is not None:
            nn.init.zeros_(module.bias)
        elif isinstance(module, nn.Embedding):
            nn.init.normal_(module.weight, std=cfg.initializer_range)

    @property
    def device(self):
        return self.lm_head.weight.device

    @property
    def dtype(self):
        return self.lm_head.weight.dtype

    def get_input_embeddings(self):
        return self.backbone.get_input_embeddings()

    def set_input_embeddings(self, new_embeddings):
        return self.backbone.set_input_embeddings(new_embeddings)

    def get_output_embeddings(self):
        return self.lm_head

    def new_cache(self, batch_size=1, dtype=None, device=None):
        return HybridMambaAttentionDynamicCache(self.config, batch_size, dtype=dtype or self.dtype,
                                                device=device or self.device)

    def set_pdrop_args(self, **kwargs):
        for key, value in kwargs.items():
            setattr(self.backbone, key, value)

    def init_cross_attn_from_self_attn(self):
        if self.backbone.merge_modules is not None:
            layers = self.backbone.pdrop_args.get("pdrop_layers", [])
            for idx, module in enumerate(self.backbone.merge_modules):
                if not isinstance(module, nn.Identity):
                    module.load_state_dict(self.backbone.layers[layers[idx]].mixer.state_dict())

    def forward(self, input_ids=None, inputs_embeds=None, position_ids=None, past_key_values=None,
                labels=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                use_cache=None, cache_position=None, attention_mask=None, logits_to_keep=None,
                **kwargs):
        out = self.backbone(input_ids, past_key_values=past_key_values, inputs_embeds=inputs_embeds,
                            position_ids=position_ids, output_hidden_states=output_hidden_states,
                            use_cache=use_cache, cache_position=cache_position,
                            attention_mask=attention_mask, labels=labels, **kwargs)
        hidden = out.last_hidden_state
        if labels is not None:
            raise NotImplementedError("loss computation is a training feature")
        # reference computes lm_head over all L positions (:2433); generate() needs [:, -1]
        keep = self.config.num_logits_to_keep if logits_to_keep is None else logits_to_keep
        if isinstance(keep, int) and keep > 0:
            hidden = hidden[:, -keep:]
        logits = _linear(self.lm_head, hidden.to(self.lm_head.weight.dtype)).float()
        return CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=out.past_key_values,
                                      hidden_states=out.hidden_states)
