"""Host logic of the hybrid cache (no GPU): in-place K / V growth hands out what torch.cat would (modeling_nano.py:246-251),
and the static-decode bookkeeping (fixed buffers, one device-side (slot, count) pair per distinct cache length)."""
import pytest
import torch

from timeviper_amd.model.llm.nano import HybridMambaAttentionDynamicCache, NemotronHConfig


def _cfg(pattern="M*-*"):
    return NemotronHConfig(vocab_size=64, hidden_size=32, intermediate_size=48, num_hidden_layers=len(pattern),
                           hybrid_override_pattern=pattern, num_attention_heads=2, head_dim=16, num_key_value_heads=1,
                           ssm_state_size=8, mamba_num_heads=4, mamba_n_groups=1, mamba_head_dim=8, mamba_chunk_size=8)


def test_update_is_concatenation_with_spare_capacity():
    cache = HybridMambaAttentionDynamicCache(_cfg(), 2, dtype=torch.float32, kv_reserve=3)
    g = torch.Generator().manual_seed(0)
    ks = [torch.randn(2, n, 1, 16, generator=g) for n in (7, 1, 1, 1, 1, 5, 1)]
    vs = [torch.randn(2, n, 1, 16, generator=g) for n in (7, 1, 1, 1, 1, 5, 1)]
    for i, (k, v) in enumerate(zip(ks, vs)):
        kc, vc = cache.update(k, v, 1)
        assert torch.equal(kc, torch.cat(ks[:i + 1], 1)) and torch.equal(vc, torch.cat(vs[:i + 1], 1))
        assert cache.get_seq_length() == kc.shape[1]
    assert cache.key_cache[3].shape[-1] == 0                      # the other attention layer is untouched
    buf = cache.kv_buffers(1)[0]
    assert buf.shape[1] >= 17 and cache.key_cache[1].data_ptr() == buf.data_ptr()


def test_static_decode_bookkeeping_with_unequal_layer_lengths():
    cfg = _cfg()
    cache = HybridMambaAttentionDynamicCache(cfg, 1, dtype=torch.float32)
    with pytest.raises(RuntimeError):
        cache.begin_static_decode(4)                              # nothing prefilled yet
    g = torch.Generator().manual_seed(1)
    lens = {1: 20, 3: 12}                                         # a token-drop stage between the two attention layers
    for i, n in lens.items():
        cache.update(torch.randn(1, n, 1, 16, generator=g), torch.randn(1, n, 1, 16, generator=g), i)
    before = {i: cache.key_cache[i].clone() for i in lens}
    cache.begin_static_decode(3)
    assert cache.static_decode and cache.static_room() >= 3
    assert cache.decode_lens(1) is not cache.decode_lens(3)
    assert int(cache.decode_lens(1)) == 21 and int(cache.decode_lens(3)) == 13
    ptrs = {i: cache.kv_buffers(i)[0].data_ptr() for i in lens}
    for step in range(3):
        for i, n in lens.items():
            k = torch.full((1, 1, 1, 16), float(step + 1))
            kc, _ = cache.update(k, -k, i)
            assert kc.shape[1] == n + step + 1 and torch.equal(kc[:, -1:], k) and torch.equal(kc[:, :n], before[i])
            assert cache.kv_buffers(i)[0].data_ptr() == ptrs[i]   # the buffers a captured graph points at never move
        cache.advance_static_device()
        assert int(cache.decode_lens(1)) == 22 + step and int(cache.decode_lens(3)) == 14 + step
    with pytest.raises(RuntimeError):                             # one token per step
        cache.update(torch.zeros(1, 2, 1, 16), torch.zeros(1, 2, 1, 16), 1)
    # what a REPLAYED step leaves to the host: lengths and views only
    room = cache.static_room()
    cache.advance_static_host()
    assert cache.get_seq_length() == 24 and cache.static_room() == room - 1
    cache.end_static_decode()
    assert not cache.static_decode
    cache.update(torch.zeros(1, 400, 1, 16), torch.zeros(1, 400, 1, 16), 1)      # an ordinary cache again: it may grow
    assert cache.key_cache[1].shape[1] == 424
