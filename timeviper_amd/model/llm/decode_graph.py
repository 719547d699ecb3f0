"""One decode step as ONE hipGraph launch.

A generated token of the hybrid stack is ~330 small launches (27 Mamba-2 mixers x {in_proj, conv update, state update,
gated norm, out_proj, norm}, 25 MLPs, 4 attention layers) that together move the 16.6 GB of weights once: ~2 ms of HBM
time at batch 1, against ~10 ms of host work when every launch is issued from Python (bench.py --config decode).  The
reference's loop (HF `generate`, evaluate.py:507-525 -> modeling_nano.py:484-546, 1666-1689) has the same shape and the
same problem.  On MI355X the answer is a captured graph rather than a tracing compiler: the step is made
launch-parameter-static — the K / V buffers have fixed capacity, the write slot and the key count live in device
memory (`HybridMambaAttentionDynamicCache.begin_static_decode`), the attention kernel reads the count there
(`tv_attn_decode_fwd(seqlens_k)`), the conv / SSM states are updated in place — captured once after two eager warm-up
tokens, and replayed per token with one `hipGraphLaunch`.

Valid for stacks whose decode step does not depend on the host-side position: the Nemotron-H hybrid has no positional
encoding (modeling_nano.py:1012-1220); a rotary backbone (Qwen2) would bake its angle in — `GraphedDecodeStep` is not
offered for it.
"""
from __future__ import annotations

from typing import Callable

import warnings

import torch

__all__ = ["GraphedDecodeStep"]


class GraphedDecodeStep:
    """`step(token_ids)` -> next token ids (greedy), the same arithmetic as the eager loop.

    step_fn(ids) runs the language model on ONE token per sequence — ids is a (B, 1) int64 tensor that lives at a fixed
    address — against `cache` (already in static-decode mode) and returns the logits of that token, (B, V) or
    (B, 1, V).  The first `warmup` calls run eagerly (lazy initialisations: kernel attributes, cached constants,
    library workspaces), the next one is captured, every later one replays; if the capture raises, the stepper restores the
    cache's host bookkeeping and keeps running the eager step (`capture_error` holds the reason)."""

    def __init__(self, step_fn: Callable[[torch.Tensor], torch.Tensor], cache, batch_size: int, device,
                 warmup: int = 2):
        if not getattr(cache, "static_decode", False):
            raise RuntimeError("GraphedDecodeStep: call cache.begin_static_decode(max_new_tokens) first")
        self.step_fn, self.cache, self.warmup = step_fn, cache, int(warmup)
        self.ids = torch.zeros((batch_size, 1), dtype=torch.int64, device=device)
        self.next_ids = None
        self.logits = None
        self.graph = None
        self.capture_error = None     # set when the capture failed: the stepper then runs every step eagerly
        self.calls = 0

    def _run(self):
        logits = self.step_fn(self.ids)
        logits = logits[:, -1] if logits.dim() == 3 else logits
        nxt = logits.argmax(-1)
        self.cache.advance_static_device()
        return logits, nxt

    @torch.inference_mode()
    def step(self, token_ids: torch.Tensor) -> torch.Tensor:
        if self.cache.static_room() < 1:
            raise RuntimeError("GraphedDecodeStep: the key / value buffers reserved by begin_static_decode are full")
        self.ids.copy_(token_ids.view(self.ids.shape))
        self.calls += 1
        if self.graph is not None:
            self.graph.replay()
            self.cache.advance_static_host()
        elif self.calls <= self.warmup or self.capture_error is not None:
            self.logits, self.next_ids = self._run()
        else:
            # host bookkeeping the capture is about to advance (no kernel runs during it): restored if it fails
            lens = {i: self.cache._kv_len[i] for i in self.cache.attention_layers}
            try:
                side = torch.cuda.Stream(device=self.ids.device)
                side.wait_stream(torch.cuda.current_stream())
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    self.logits, self.next_ids = self._run()
                torch.cuda.current_stream().wait_stream(side)
            except Exception as e:  # a non-capturable op, a lazy library initialisation the warm-up did not reach, ...
                # the eager step works where the capture does not: fall back to it for the rest of the generation
                # (the device-side position / key counts were not touched: nothing of the capture executed)
                self.capture_error = e
                if isinstance(e, torch.cuda.OutOfMemoryError):
                    raise               # not a capture problem: the eager loop would only run out of memory later
                warnings.warn(f"decode: graph capture failed ({type(e).__name__}: {e}); every further token runs as an eager "
                              "sequence of launches (several times slower per token)", RuntimeWarning, stacklevel=2)
                for i, n in lens.items():
                    kb, vb = self.cache._kv_buf[i]
                    self.cache._kv_len[i] = n
                    self.cache.key_cache[i], self.cache.value_cache[i] = kb[:, :n], vb[:, :n]
                torch.cuda.synchronize()
                self.logits, self.next_ids = self._run()
                return self.next_ids
            # the capture ran the host bookkeeping of one step (cache lengths) and no kernel: this replay is that step
            self.graph = graph
            graph.replay()
        return self.next_ids
