"""`GenericTimeViperVLM` / `HybridTimeViperVLM` — the outer drop-in boundary
(reference timeviper/model/generic_vlm.py:60-972, hybrid_vlm.py:28-50): same constructor,
attributes (`vision_backbone`, `projector`, `llm_backbone`, `arch_specifier`,
`llm_tokenizer`, `default_token_id`, `config`, `device`) and `forward` / `generate`
signatures, inference paths only.

forward (prefill) = ViT over 256-frame clips -> ToMe+MLP projector -> splice one
(tokens_per_frame, D) block per `<image>` placeholder -> hybrid LM (:221-399).
"""
from __future__ import annotations

import os

from typing import List, Optional, Union

import torch
import torch.nn as nn

from .llm import GenericLLMBackbone
from .llm.nano import CausalLMOutputWithPast
from .projector import MLPProjector, MultiMLPProjector, MultiToMe16_mlp_hd64, ToMe16_mlp_hd64
from .vit import VisionBackbone

IGNORE_INDEX = -100
DEFAULT_TOKEN = "<image>"


def _parse_compressed_tokens(arch_specifier: str) -> int:
    parts = arch_specifier.split("-")
    assert parts[-1].isdigit(), f"Cannot parse compressed tokens from {arch_specifier}"
    return int(parts[-1])


class GenericTimeViperVLM(nn.Module):
    supports_gradient_checkpointing = True
    _is_stateful = False

    def __init__(self, model_id: str, vision_backbone: VisionBackbone,
                 llm_backbone: GenericLLMBackbone, enable_mixed_precision_training: bool = True,
                 arch_specifier: str = "gelu_mlp", visual_token_order: str = "raw",
                 disable_data_packing: bool = False) -> None:
        super().__init__()
        self.model_family = f"{llm_backbone.llm_family}"
        self.model_id = model_id
        self.vision_backbone = vision_backbone
        self.llm_backbone = llm_backbone
        self.enable_mixed_precision_training = enable_mixed_precision_training
        self.main_input_name = "input_ids"
        self.arch_specifier = arch_specifier
        assert visual_token_order in ["raw", "ascending", "descending"]
        self._initialize_projector(visual_token_order)
        self.vision_backbone_requires_grad = False
        self.all_module_keys = ["vision_backbone", "llm_backbone", "projector"]
        self.eos_token_ids_to_use = getattr(llm_backbone, "terminators",
                                            [llm_backbone.tokenizer.eos_token_id])
        bb = self.llm_backbone.llm.backbone
        self.use_pdrop = bool(getattr(bb, "use_pdrop", False))
        self.pdrop_args = {"use_pdrop": self.use_pdrop}
        if self.use_pdrop:  # :105-126
            types = bb.pdrop_types
            self.pdrop_args.update({"pdrop_compress_types": [t[0] for t in types],
                                    "pdrop_layers": [int(t[1]) for t in types],
                                    "pdrop_ratios": [1] + [float(t[2]) for t in types]})
            self.llm_backbone.llm.set_pdrop_args(**self.pdrop_args)
        self.disable_data_packing = disable_data_packing
        self.vit_clip_frames = 256  # :274
        # The reference's 256-frame clips bound activation memory on 80 GB parts.  Towers that treat
        # every frame on its own (timm ViTs + frame-wise ToMe) give the same result for any clip
        # size, and larger GEMM / elementwise launches run ~6 % faster on MI355X (288 GB), so such
        # towers are fed `vit_clip_fuse` clips at a time.  InternVideo2 regroups frames into tubes
        # with a reshape that depends on the clip length (model.py:178-182): it regroups every 256 frames
        # as a separate call would, then runs the tubes of all fused clips as one batch.
        self.vit_clip_fuse = 8

    # ---- attributes read by callers (evaluate.py:223,392-393,619) ----
    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def dtype(self) -> torch.dtype:
        return torch.bfloat16

    @staticmethod
    def can_generate() -> bool:
        return True

    @property
    def llm_tokenizer(self):
        return self.llm_backbone.tokenizer

    @property
    def default_token_id(self):
        return self.llm_tokenizer.convert_tokens_to_ids(DEFAULT_TOKEN)

    @property
    def config(self):
        return self.llm_backbone.llm.config

    def _initialize_projector(self, visual_token_order):
        vdim, ldim = self.vision_backbone.embed_dim, self.llm_backbone.embed_dim
        multi = hasattr(self.vision_backbone, "backbone_ids") and \
            isinstance(self.vision_backbone.backbone_ids, list)
        if multi:
            dims = {bid: self.vision_backbone.backbones[bid.replace("-", "_")].embed_dim
                    for bid in self.vision_backbone.backbone_ids}
            if "gelu_mlp" in self.arch_specifier:
                self.projector = MultiMLPProjector(dims, ldim)
            elif "tome_mlp" in self.arch_specifier:
                self.num_compressed_tokens = _parse_compressed_tokens(self.arch_specifier)
                self.projector = MultiToMe16_mlp_hd64(
                    dims, ldim, mlp_type=self.arch_specifier.split("-")[0],
                    num_compressed_tokens=self.num_compressed_tokens, token_order=visual_token_order)
            else:
                raise ValueError(f"MultiViTBackbone only supports gelu_mlp or tome_mlp projector "
                                 f"for now, got {self.arch_specifier}")
        elif "gelu_mlp" in self.arch_specifier:
            self.projector = MLPProjector(vdim, ldim)
        elif "tome_mlp" in self.arch_specifier:
            self.num_compressed_tokens = _parse_compressed_tokens(self.arch_specifier)
            self.projector = ToMe16_mlp_hd64(
                vdim, ldim, mlp_type=self.arch_specifier.split("-")[0],
                num_compressed_tokens=self.num_compressed_tokens, token_order=visual_token_order)
        else:
            raise ValueError(f"GenericTimeViperVLM with projector architecture "
                             f"`{self.arch_specifier}` is not supported!")

    # ---- vision ----
    def projector_forward(self, patch_features, is_video=False):
        """:401-438."""
        ident = self.vision_backbone.get_identifier
        if "tome_mlp" in self.arch_specifier:
            if ident in ["siglip", "dinov2siglip", "dinov2"]:
                return self.projector(patch_features, compress=True, local_num_frames=1)
            if ident in ["internvideo2"]:
                lnf = 4 if is_video else 1
                ve = self.projector(patch_features, compress=True, local_num_frames=lnf)
                if is_video:
                    b, n, d = ve.shape
                    ve = ve.view(b * 4, n // 4, d)
                return ve
            if ident in ["multivit"]:
                lnf = {bid: (4 if ("internvideo2" in bid and is_video) else 1)
                       for bid in self.vision_backbone.backbone_ids}
                return self.projector(patch_features, compress=True, local_num_frames=lnf)
            raise ValueError(ident)
        ve = self.projector(patch_features)
        if ident in ["internvideo2"] and is_video:
            b, n, d = ve.shape
            ve = ve.view(b * 4, n // 4, d)
        self.num_compressed_tokens = ve.shape[1]
        return ve

    @torch.no_grad()
    def encode_vision(self, vision_inputs, is_video: bool):
        """eval branch of :266-281: clips of 256 frames through ViT + projector."""
        vb, n, kw = self.vision_backbone, self.vit_clip_frames, {}
        if getattr(vb, "frame_independent", False):
            n *= max(1, int(self.vit_clip_fuse))
        elif getattr(vb, "batched_clips", False) and is_video:
            # the backbone regroups every `vit_clip_frames` frames as a separate call would and runs
            # the tubes of `vit_clip_fuse` clips as one batch
            kw = {"clip_frames": n}
            n *= max(1, int(self.vit_clip_fuse))
        feats = [self.projector_forward(vb(clip, is_video=is_video, **kw), is_video=is_video)
                 for clip in vision_inputs.split(split_size=n)]
        return torch.cat(feats, dim=0)

    # ---- fusion ----
    def get_fused_data_nopacked(self, visual_embeddings, input_ids, labels=None):
        """:517-564, batch 1.  The reference walks the placeholders in a Python loop (one
        iteration per frame); here the common layout — one contiguous run of `<image>`
        tokens — is a single concatenation, and anything else takes the general walk."""
        if labels is not None:
            raise NotImplementedError("labels are a training feature")
        ids = input_ids[0]
        is_img = ids == self.default_token_id
        pos = is_img.nonzero(as_tuple=False).flatten()
        n = pos.numel()
        embed = self.llm_backbone.embed_input_ids
        first, last = int(pos[0]), int(pos[-1])
        vis = visual_embeddings
        if last - first + 1 == n and n == vis.shape[0]:
            parts = [embed(ids[None, :first]), vis.reshape(1, -1, vis.shape[-1]).to(self.dtype_of(embed))]
            if last + 1 < ids.shape[0]:
                parts.append(embed(ids[None, last + 1:]))
            return torch.cat(parts, dim=1), None
        plist = pos.tolist()
        out = [embed(ids[None, :plist[0]])]
        for i, s in enumerate(plist):
            out.append(vis[i:i + 1].to(out[0].dtype))
            start = s + 1
            end = plist[i + 1] if i < n - 1 else ids.shape[0]
            if start < ids.shape[0] and bool(is_img[start]):
                continue
            out.append(embed(ids[None, start:end]))
        return torch.cat(out, dim=1), None

    def dtype_of(self, embed):
        return self.llm_backbone.llm.get_input_embeddings().weight.dtype

    def pdrop_bookkeeping(self, input_ids, visual_embeddings):
        """:291-309."""
        is_img = input_ids.eq(self.default_token_id)
        return {"first_vision_token_positions": torch.argmax(is_img.int(), dim=1),
                "text_prompt_lens": [int(input_ids.shape[1] - int(is_img[0].sum()))],
                "num_vision_tokens": [visual_embeddings.size(0) * visual_embeddings.size(1)],
                "is_interleaved": False}

    # ---- forward / generate ----
    def forward(self, input_ids: Optional[torch.LongTensor] = None, attention_mask=None,
                pixel_values=None, pixel_values_videos=None, labels=None, inputs_embeds=None,
                past_key_values=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, position_ids=None,
                cache_position=None, inference_params=None, num_last_tokens: int = 0,
                answer_prompt: Optional[str] = None, logits_to_keep: Union[int, torch.Tensor, None] = None,
                image_grid_thw=None, video_grid_thw=None, logits_cache=None, txt_seq_lens=None,
                img_seq_lens=None, vid_seq_lens=None, multimodal_indices=None,
                visual_embeddings: Optional[torch.Tensor] = None) -> CausalLMOutputWithPast:
        if self.training:
            raise NotImplementedError("timeviper_amd implements the inference forward only")
        if pixel_values is None and pixel_values_videos is None and visual_embeddings is None:
            return self.llm_backbone(input_ids=input_ids, attention_mask=None,
                                     position_ids=position_ids, past_key_values=past_key_values,
                                     inputs_embeds=inputs_embeds, use_cache=use_cache,
                                     output_hidden_states=output_hidden_states,
                                     cache_position=cache_position, logits_to_keep=logits_to_keep,
                                     train_pdrop_args=self.pdrop_args if self.use_pdrop else None)
        if visual_embeddings is None:
            vin = pixel_values_videos if pixel_values_videos is not None else pixel_values
            visual_embeddings = self.encode_vision(vin, is_video=pixel_values_videos is not None)
        assert input_ids is not None and input_ids.shape[0] == 1
        train_pdrop_args = None
        if self.use_pdrop:
            train_pdrop_args = self.pdrop_bookkeeping(input_ids, visual_embeddings)
            self.pdrop_args.update(train_pdrop_args)
            train_pdrop_args = self.pdrop_args
        fused, _ = self.get_fused_data_nopacked(visual_embeddings, input_ids, None)
        L = fused.shape[1]
        # flash_attention_2 => mask None, positions = arange(L)   (:496-497, :512-514)
        position_ids = torch.arange(L, device=fused.device).unsqueeze(0)
        fused_cache_position = torch.arange(L, device=fused.device) if cache_position is not None else None
        return self.llm_backbone(input_ids=None, attention_mask=None, position_ids=position_ids,
                                 past_key_values=past_key_values, inputs_embeds=fused,
                                 use_cache=use_cache, output_hidden_states=output_hidden_states,
                                 cache_position=fused_cache_position, logits_to_keep=logits_to_keep,
                                 train_pdrop_args=train_pdrop_args)

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None,
                                      inputs_embeds=None, cache_position=None, **kwargs):
        """generic_vlm.py:762-848 for the native greedy loop below.  Prefill (no cache yet, or more
        than one position): the tokenised `answer_prompt` is appended to the prompt (:772-785) and the
        pixels ride along (:793-795).  Decode step: no pixels; position = tokens the attention cache
        holds (:797-826; the reference reads layer 7 / 14's KV length, `get_seq_length()` here finds
        the first attention layer) and, under flash_attention_2, no mask (:828-832)."""
        is_prefill = past_key_values is None or cache_position is None or cache_position.shape[0] != 1
        if is_prefill:
            ap = kwargs.get("answer_prompt")
            if ap:
                ap_ids = self.llm_backbone.tokenizer(ap, add_special_tokens=False, return_tensors="pt").input_ids
                ap_ids = ap_ids.to(input_ids.device).expand(input_ids.shape[0], -1)
                input_ids = torch.cat([input_ids, ap_ids], dim=1)
                if attention_mask is not None:
                    attention_mask = torch.cat([attention_mask, torch.ones_like(ap_ids)], dim=1)
            if attention_mask is not None and not bool(attention_mask.all()):
                raise NotImplementedError("padded prompts (attention_mask with zeros) are not on the batch-1 "
                                          "evaluation path")
            return {"input_ids": input_ids, "attention_mask": None, "past_key_values": past_key_values,
                    "pixel_values": kwargs.get("pixel_values"),
                    "pixel_values_videos": kwargs.get("pixel_values_videos"), "use_cache": True,
                    "cache_position": torch.zeros(1, dtype=torch.long)}
        past_len = past_key_values.get_seq_length()
        cp = torch.tensor([max(past_len, 1)])          # host side: the mixers branch on it
        return {"input_ids": input_ids[:, -1:], "attention_mask": None, "past_key_values": past_key_values,
                "pixel_values": None, "pixel_values_videos": None, "use_cache": True, "cache_position": cp,
                "position_ids": cp.view(1, 1).to(input_ids.device)}

    @torch.inference_mode()
    def generate(self, *args, **kwargs):
        """Greedy decoding with the hybrid cache: what evaluate.py:507-525 asks of HF's
        GenerationMixin (`do_sample=False, use_cache=True, temperature=0, answer_prompt=...`), with
        the reference's return contract (generic_vlm.py:743-760): the DECODED text of the new tokens
        (prompt stripped, `.strip()`-ed), which evaluate.py:529 hands to `extract_answer`.
        `return_ids=True` (or HF's `return_dict_in_generate=True`, or a tokenizer without `decode`)
        returns the (1, n_new) id tensor instead."""
        input_ids = args[0] if args else kwargs.pop("input_ids", None)
        pixel_values = kwargs.pop("pixel_values", None)
        pixel_values_videos = kwargs.pop("pixel_values_videos", None)
        attention_mask = kwargs.pop("attention_mask", None)
        max_new_tokens = int(kwargs.pop("max_new_tokens", 128))
        return_ids = bool(kwargs.pop("return_ids", False)) or bool(kwargs.pop("return_dict_in_generate", False))
        if kwargs.pop("do_sample", False):
            raise NotImplementedError("only greedy decoding is on the evaluation path")
        eos = kwargs.pop("eos_token_id", None)
        eos = self.eos_token_ids_to_use if eos is None else eos
        eos = set(int(t) for t in (eos if isinstance(eos, (list, tuple)) else [eos]))
        llm = self.llm_backbone.llm
        cache = llm.new_cache(1, dtype=self.dtype, device=self.device)
        mi = self.prepare_inputs_for_generation(input_ids, past_key_values=None, attention_mask=attention_mask,
                                                pixel_values=pixel_values, pixel_values_videos=pixel_values_videos,
                                                answer_prompt=kwargs.pop("answer_prompt", None))
        out = self.forward(input_ids=mi["input_ids"], pixel_values=mi["pixel_values"],
                           pixel_values_videos=mi["pixel_values_videos"], past_key_values=cache,
                           use_cache=True, cache_position=mi["cache_position"])
        new_tokens: List[int] = []
        tok = out.logits[:, -1].argmax(-1)
        pdargs = self.pdrop_args if self.use_pdrop else None
        stepper = self._graphed_decode_step(cache, max_new_tokens, pdargs)
        for _ in range(max_new_tokens):
            t = int(tok)
            new_tokens.append(t)
            if t in eos:
                break
            if stepper is not None:            # the same step as below, replayed as one hipGraph launch
                tok = stepper.step(tok.view(1, 1))
                continue
            mi = self.prepare_inputs_for_generation(tok.view(1, 1), past_key_values=cache,
                                                    cache_position=torch.zeros(1, dtype=torch.long))
            out = self.llm_backbone(input_ids=mi["input_ids"], past_key_values=cache, use_cache=True,
                                    cache_position=mi["cache_position"], position_ids=mi["position_ids"],
                                    train_pdrop_args=pdargs)
            tok = out.logits[:, -1].argmax(-1)
        ids = torch.tensor([new_tokens], device=self.device)
        tokenizer = self.llm_backbone.tokenizer
        if return_ids or not hasattr(tokenizer, "decode"):
            return ids
        return tokenizer.decode(ids[0], skip_special_tokens=False).strip()

    def _graphed_decode_step(self, cache, max_new_tokens, pdargs):
        """The decode step behind `generate` as a captured graph (llm/decode_graph.py) where the stack allows it: the
        Nemotron-H hybrid (no positional encoding), bf16, attention head_dim 128 (tv_attn_decode_fwd), on the GPU, and
        enough tokens to pay for two eager warm-up steps and the capture.  TV_DECODE_GRAPH=0 keeps the eager loop."""
        from .llm.decode_graph import GraphedDecodeStep
        from .llm.nano import HybridMambaAttentionDynamicCache, NemotronHForCausalLM
        llm = self.llm_backbone.llm
        if (os.environ.get("TV_DECODE_GRAPH", "1") == "0" or self.device.type != "cuda" or max_new_tokens < 8
                or not isinstance(llm, NemotronHForCausalLM) or not isinstance(cache, HybridMambaAttentionDynamicCache)
                or self.dtype != torch.bfloat16 or not cache.attention_layers
                or cache.key_cache[cache.attention_layers[0]].shape[-1] != 128):
            return None
        cache.begin_static_decode(max_new_tokens)
        host_pos = torch.ones(1, dtype=torch.long)              # "not the prefill" for the mixers' host-side branch
        dev_pos = torch.ones((1, 1), dtype=torch.long, device=self.device)

        def step_fn(ids):
            return self.llm_backbone(input_ids=ids, past_key_values=cache, use_cache=True, cache_position=host_pos,
                                     position_ids=dev_pos, train_pdrop_args=pdargs).logits
        return GraphedDecodeStep(step_fn, cache, 1, self.device)

    @classmethod
    def from_pretrained(cls, pretrained_checkpoint, model_id: str, vision_backbone: VisionBackbone,
                        llm_backbone: GenericLLMBackbone, enable_mixed_precision_training: bool = True,
                        arch_specifier: str = "gelu_mlp", visual_token_order: str = "raw"):
        """generic_vlm.py:874-910 (called at evaluate.py:198-214): build, `torch.load` the
        state dict, `load_state_dict(strict=True)`, freeze, eval, move to the GPU in bf16."""
        vlm = cls(model_id, vision_backbone, llm_backbone,
                  enable_mixed_precision_training=enable_mixed_precision_training,
                  arch_specifier=arch_specifier, visual_token_order=visual_token_order)
        pretrained_weights = torch.load(pretrained_checkpoint, map_location="cpu")
        vlm.load_state_dict(pretrained_weights, strict=True)
        vlm.requires_grad_(False)
        vlm.eval()
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        vlm.to(device, dtype=torch.bfloat16)
        return vlm


class HybridTimeViperVLM(GenericTimeViperVLM):
    """hybrid_vlm.py:28-50 only patches HF's cache preparation; with the native greedy
    `generate` above there is nothing left to patch."""
