"""`GenericLLMBackbone` (reference timeviper/model/llm/llm_factory.py:41-198): thin
wrapper owning `.llm` (the causal LM), the tokenizer and the `<image>` token.  There is
no network here, so instead of `from_pretrained` the backbone is built from a config
(random init, the reference's `_from_config` branch :101) and the tokenizer is either
handed in or a minimal stand-in that only knows the special-token ids."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .nano import NemotronHConfig, NemotronHForCausalLM
from .qwen2 import Qwen2Config, Qwen2ForCausalLM

MODEL_REGISTRY = {
    # llm_registry.py:64-97 — ids the reference accepts for the nano family
    "nanov2-9b": "nvidia/NVIDIA-Nemotron-Nano-9B-v2",
    "nanov2-9b-base": "nvidia/NVIDIA-Nemotron-Nano-9B-v2-Base",
}
QWEN2_REGISTRY = {
    # llm_registry.py:65-77.  Only the 7B geometries have a config preset here (public
    # config.json values); the other ids need an explicit `config=`.
    "qwen2-7b": "Qwen/Qwen2-7B", "qwen2-7b-instruct": "Qwen/Qwen2-7B-Instruct",
    "qwen2-1.5b": "Qwen/Qwen2-1.5B", "qwen2-1.5b-instruct": "Qwen/Qwen2-1.5B-Instruct",
    "qwen2.5-7b-instruct": "Qwen/Qwen2.5-7B-Instruct", "qwen2.5-7b-base": "Qwen/Qwen2.5-7B-Base",
    "qwen2.5-3b-instruct": "Qwen/Qwen2.5-3B-Instruct", "qwen2.5-3b-base": "Qwen/Qwen2.5-3B-Base",
}
_QWEN2_7B = ("qwen2-7b", "qwen2-7b-instruct", "qwen2.5-7b-instruct", "qwen2.5-7b-base")
DEFAULT_TOKEN = "<image>"


def llm_family_of(llm_backbone_id: str) -> str:
    if llm_backbone_id in MODEL_REGISTRY:
        return "nano"
    if llm_backbone_id in QWEN2_REGISTRY:
        return "qwen2"
    raise ValueError(f"LLM backbone `{llm_backbone_id}` is not supported!")


def get_llm_config(llm_backbone_id: str, **over):
    family = llm_family_of(llm_backbone_id)
    if family == "nano":
        return NemotronHConfig.nemotron_nano_9b_v2(**over)
    if llm_backbone_id in _QWEN2_7B:
        return Qwen2Config.qwen2_5_7b(**over)
    raise ValueError(f"no built-in geometry for `{llm_backbone_id}`; pass config=Qwen2Config(...)")


class SyntheticTokenizer:
    """Stand-in used for synthetic benchmarks/tests (no tokenizer files offline)."""

    def __init__(self, vocab_size: int):
        self.vocab_size = vocab_size
        self.image_token_id = vocab_size - 1
        self.eos_token_id = 2
        self.pad_token_id = 0

    def convert_tokens_to_ids(self, tok):
        return self.image_token_id if tok == DEFAULT_TOKEN else 3

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        """ids -> text: one `<id>` word per token (there is no vocabulary offline)."""
        return " ".join(f"<{int(i)}>" for i in ids)

    def __call__(self, text: str, add_special_tokens: bool = False, return_tensors: str = "pt"):
        """text -> ids, one id per whitespace-separated word (a `<id>` word maps back to its id,
        anything else to a stable id in [3, vocab-1)) — the call shape of
        `tokenizer(answer_prompt, add_special_tokens=False, return_tensors="pt").input_ids`
        (generic_vlm.py:774-776)."""
        import types
        import zlib
        out = []
        for w in text.split():
            if w.startswith("<") and w.endswith(">") and w[1:-1].isdigit():
                out.append(int(w[1:-1]))
            else:
                out.append(3 + zlib.crc32(w.encode()) % max(1, self.vocab_size - 4))
        return types.SimpleNamespace(input_ids=torch.tensor([out], dtype=torch.long))

    def __len__(self):
        return self.vocab_size


class GenericLLMBackbone(nn.Module):
    def __init__(self, llm_backbone_id: str, config=None,
                 tokenizer=None, llm_max_length: Optional[int] = None, inference_mode: bool = True,
                 attn_implementation: str = "flash_attention_2", merge_module: str = "no_merge",
                 use_pdrop: bool = False, pdrop_type: Optional[str] = None) -> None:
        super().__init__()
        self.identifier = llm_backbone_id
        self.llm_family = llm_family_of(llm_backbone_id)
        self.llm_max_length = llm_max_length
        self.inference_mode = inference_mode
        if config is None:
            config = get_llm_config(llm_backbone_id, merge_module=merge_module, use_pdrop=use_pdrop,
                                    pdrop_type=pdrop_type)
        else:
            config.merge_module, config.use_pdrop, config.pdrop_type = merge_module, use_pdrop, pdrop_type
        config._attn_implementation = attn_implementation
        want = NemotronHConfig if self.llm_family == "nano" else Qwen2Config
        if not isinstance(config, want):
            raise TypeError(f"`{llm_backbone_id}` needs a {want.__name__}, got {type(config).__name__}")
        self.llm = NemotronHForCausalLM(config) if self.llm_family == "nano" else Qwen2ForCausalLM(config)
        self.tokenizer = tokenizer or SyntheticTokenizer(config.vocab_size)
        self.terminators = [self.tokenizer.eos_token_id]

    @property
    def embed_dim(self) -> int:
        return self.llm.config.hidden_size

    @property
    def half_precision_dtype(self) -> torch.dtype:
        return torch.bfloat16

    def embed_input_ids(self, input_ids: torch.LongTensor) -> torch.Tensor:
        return self.llm.get_input_embeddings()(input_ids)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, inference_params=None,
                num_last_tokens: int = 0, cache_position=None, logits_to_keep=None, seq_idx=None,
                train_pdrop_args=None):
        """llm_factory.py:177-198."""
        return self.llm(input_ids=input_ids, attention_mask=attention_mask,
                        position_ids=position_ids, past_key_values=past_key_values,
                        inputs_embeds=inputs_embeds, labels=labels, use_cache=use_cache,
                        output_hidden_states=output_hidden_states, cache_position=cache_position,
                        logits_to_keep=logits_to_keep, train_pdrop_args=train_pdrop_args)
