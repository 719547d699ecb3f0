"""pdrop / TransV machinery shared by the LLM backbones (reference: the same code appears in
llm_repo/nano/modeling_nano.py:1761-2095 and llm_repo/qwen2/modeling_qwen2.py:482-876): eval,
batch 1, no padding.  The host class provides `layers`, `pdrop_ratios`, `pdrop_compress_types`,
`merge_modules`, `merge_module_names`, `alpha`, `last_pdrop_trace` and `_rank_attention(layer)`
(the attention module whose q_proj / k_proj rank the vision tokens, with `num_heads`,
`num_key_value_heads`, `head_dim`)."""
from __future__ import annotations

import torch

from ... import kernels as K


class PdropMixin:
    # ---- TransV / pdrop (reference pdrop_no_pack :1779-2095, eval, batch 1) ----
    def merge_dropped_information(self, features, cur_num, vision_index, start_index,
                                  top_rank_index, dropped_index):
        text = features[start_index:, :]
        if self.merge_module_names[cur_num] == "attention":
            dropped = K.gather_rows(features, dropped_index)
            merged = self.merge_modules[cur_num](text.unsqueeze(0), dropped.unsqueeze(0))[0].squeeze(0)
            return text + self.alpha[cur_num].tanh() * merged
        return text

    def pdrop_no_pack(self, features, cur_num, rank_layer, pdrop_compress_type, labels,
                      position_ids, attention_mask, first_vision_token_positions,
                      num_vision_tokens, text_prompt_lens=None):
        if features.shape[0] != 1 or attention_mask is not None:
            raise NotImplementedError("pdrop: batch size 1 without padding (reference eval path)")
        image_tokens = int(num_vision_tokens[0] * self.pdrop_ratios[cur_num])
        keep_length = int(num_vision_tokens[0] * self.pdrop_ratios[cur_num + 1])
        vision_index = int(first_vision_token_positions[0])
        feats = features[0]
        L = feats.shape[0]
        if "attn" in pdrop_compress_type:
            sa = self._rank_attention(rank_layer)
            prompt_total_len = text_prompt_lens[0] + image_tokens
            row = prompt_total_len - 1
            q_row = sa.q_proj(feats[row:row + 1]).view(sa.num_heads, sa.head_dim)
            k_all = sa.k_proj(feats[:row + 1]).view(row + 1, sa.num_key_value_heads, sa.head_dim)
            scores = K.attn_rank_scores(q_row, k_all, row + 1, vision_index, image_tokens)
            # topk with a defined tie-break: stable descending sort keeps the lower index
            order = torch.sort(scores, descending=True, stable=True).indices
            top_rank_index = order[:keep_length] + vision_index
            top_rank_index = top_rank_index.sort().values
        elif "uni" in pdrop_compress_type:
            # strictly increasing already (keep <= image_tokens), so the reference's sort is a no-op
            top_rank_index = K.uniform_keep_indices(image_tokens, keep_length, offset=vision_index,
                                                    device=feats.device)
        else:
            raise NotImplementedError(pdrop_compress_type)
        start_index = vision_index + image_tokens
        dropped_index = None
        if self.merge_modules is not None and self.merge_module_names[cur_num] != "none":
            dropped_index = K.dropped_indices(top_rank_index, vision_index, image_tokens)
            text_features = self.merge_dropped_information(
                feats, cur_num, vision_index, start_index, top_rank_index, dropped_index)
        else:
            text_features = feats[start_index:, :]
        # one gather builds [pre | kept vision | text]; the (few) text rows are then overwritten
        dev = feats.device
        full_index = torch.cat([torch.arange(vision_index, device=dev), top_rank_index,
                                torch.arange(start_index, L, device=dev)])
        new = K.gather_rows(feats, full_index)
        n_text = L - start_index
        if dropped_index is not None and n_text > 0:
            new[new.shape[0] - n_text:] = text_features.to(new.dtype)
        self.last_pdrop_trace.append({"kept": top_rank_index, "dropped": dropped_index})
        new_pos = torch.arange(new.shape[0], device=dev).unsqueeze(0) if position_ids is not None else None
        return new_pos, None, new.unsqueeze(0), None, None

    def flash_rank_drop(self, cur_num, rank_layer, features, position_ids, attention_mask, labels,
                        is_packed=False, seq_idx=None, train_pdrop_args=None):
        if is_packed:
            raise NotImplementedError("packed pdrop is a training feature")
        return self.pdrop_no_pack(features, cur_num, rank_layer, self.pdrop_compress_types[cur_num],
                                  labels, position_ids, attention_mask,
                                  train_pdrop_args["first_vision_token_positions"],
                                  train_pdrop_args["num_vision_tokens"],
                                  train_pdrop_args["text_prompt_lens"])
