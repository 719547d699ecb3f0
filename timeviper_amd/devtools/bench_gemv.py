"""Dev tool: tv_gemv_bf16_fwd against torch.nn.functional.linear (+ the stand-alone single-row kernels) on the linear
layers of one Nemotron-Nano-9B-v2 decode token.  usage (GPU box): python timeviper_amd/devtools/bench_gemv.py"""
import os
import sys
import pathlib

import torch
import torch.nn.functional as F

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from timeviper_amd import kernels as K      # noqa: E402

DEV = torch.device("cuda", 0)
HID, DIN, CONV, NH, INTER, VOCAB = 4480, 10240, 12288, 128, 15680, 131072


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    from timeviper_amd.model.llm.nano import NemotronHConfig
    cfg = NemotronHConfig.nemotron_nano_9b_v2()
    hid, inter = cfg.hidden_size, cfg.intermediate_size
    d_in = cfg.mamba_num_heads * cfg.mamba_head_dim
    conv = d_in + 2 * cfg.n_groups * cfg.ssm_state_size
    shapes = [("in_proj  (rmsnorm)", d_in + conv + cfg.mamba_num_heads, hid, "rmsnorm"),
              ("out_proj (gated)", hid, d_in, "gated"),
              ("up_proj  (rmsnorm)", inter, hid, "rmsnorm"),
              ("down_proj (relu2)", hid, inter, "relu2"),
              ("q_proj", cfg.num_attention_heads * cfg.head_dim, hid, "none"),
              ("lm_head", cfg.vocab_size, hid, "none")]
    g = torch.Generator(device=DEV).manual_seed(0)
    # many distinct weight buffers so that no call finds its matrix in the 256 MB Infinity Cache
    for name, N, Kd, pro in shapes:
        nbuf = max(2, int(600e6 // (N * Kd * 2)) + 1)
        Ws = [(torch.randn(N, Kd, device=DEV, generator=g) / Kd ** 0.5).bfloat16() for _ in range(nbuf)]
        x = torch.randn(1, 1, Kd, device=DEV, generator=g).bfloat16()
        d = torch.randn(1, 1, Kd, device=DEV, generator=g).bfloat16()
        nw = torch.ones(Kd, device=DEV)
        it = [0]

        def nextw():
            it[0] = (it[0] + 1) % nbuf
            return Ws[it[0]]
        if pro == "rmsnorm":
            ours = lambda: K.gemv_fused(x, nextw(), None, K.GEMV_RMSNORM, delta=d, sum_out=torch.empty_like(x), norm_weight=nw, eps=1e-5)
            lib = lambda: F.linear(K.rms_norm(x, nw, 1e-5, residual=d, return_sum=True)[0], nextw())
        elif pro == "gated":
            ours = lambda: K.gemv_fused(x, nextw(), None, K.GEMV_GATED, norm_weight=nw, eps=1e-5, gate=d, group_size=Kd // 8)
            lib = lambda: F.linear(K.rmsnorm_fn(x, nw, None, z=d, eps=1e-5, group_size=Kd // 8, norm_before_gate=False), nextw())
        elif pro == "relu2":
            ours = lambda: K.gemv_fused(x, nextw(), None, K.GEMV_RELU2)
            lib = lambda: F.linear(K.relu2(x), nextw())
        else:
            ours = lambda: K.gemv_fused(x, nextw(), None)
            lib = lambda: F.linear(x, nextw())
        gemm_only = lambda: F.linear(x, nextw())
        if os.environ.get("TV_BENCH_GEMV_FAST"):         # ours only, with and without the prologue
            plain = lambda: K.gemv_fused(x, nextw(), None)
            t_o, t_p = timeit(ours), timeit(plain)
            gb = N * Kd * 2 / 1e9
            print(f"{name:20s} N {N:6d} K {Kd:5d}: ours {t_o:7.1f} us = {gb / t_o * 1e6:6.0f} GB/s | without the prologue {t_p:7.1f} us", flush=True)
            continue
        t_o, t_l, t_g = timeit(ours), timeit(lib), timeit(gemm_only)
        gb = N * Kd * 2 / 1e9
        print(f"{name:20s} N {N:6d} K {Kd:5d}: ours {t_o:7.1f} us = {gb / t_o * 1e6:6.0f} GB/s | torch {t_l:7.1f} us "
              f"(linear alone {t_g:7.1f} us = {gb / t_g * 1e6:6.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
