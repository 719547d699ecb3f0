"""Dev tool: tv_selective_state_update as the decode step runs it — the 27 Mamba layers' states of Nemotron-Nano-9B-v2
(27 x 5.2 MB fp32, no launch finds its state in a cache) in one hipGraph — against the fp32 formula.
usage (GPU box): [TV_SSU_MODE=..] python timeviper_amd/devtools/bench_ssu.py"""
import os
import sys
import pathlib

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from timeviper_amd import kernels as K      # noqa: E402

DEV = torch.device("cuda", 0)
H, P, G, N, LAYERS = 128, 80, 8, 128, 27


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    states = [torch.randn(1, H, P, N, device=DEV, generator=g) for _ in range(LAYERS)]
    x = torch.randn(1, H, P, device=DEV, generator=g).bfloat16()
    dt = torch.randn(1, H, device=DEV, generator=g).bfloat16()
    A = -(torch.rand(H, device=DEV, generator=g) * 15 + 1)
    Bm = torch.randn(1, G, N, device=DEV, generator=g).bfloat16()
    Cm = torch.randn(1, G, N, device=DEV, generator=g).bfloat16()
    D, bias = torch.ones(H, device=DEV), torch.full((H,), -2.0, device=DEV)
    # check one step against the formula
    s0 = states[0].clone()
    y = K.selective_state_update(states[0], x, dt, A, Bm, Cm, D=D, dt_bias=bias, dt_softplus=True)
    d = torch.nn.functional.softplus(dt.float() + bias)
    dec = torch.exp(d * A)[0, :, None, None]
    Bh, Ch = Bm.float().repeat_interleave(H // G, 1)[0], Cm.float().repeat_interleave(H // G, 1)[0]
    ref = dec * s0[0] + (d[0, :, None] * x.float()[0])[:, :, None] * Bh[:, None, :]
    yref = (ref * Ch[:, None, :]).sum(-1) + D[:, None] * x.float()[0]
    es, ey = (states[0][0] - ref).abs().max().item(), (y.float()[0] - yref).abs().max().item() / yref.abs().max().item()
    assert os.environ.get('TV_SSU_MODE') == '99' or (es < 1e-4 and ey < 1e-2), (es, ey)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for s in states:
            K.selective_state_update(s, x, dt, A, Bm, Cm, D=D, dt_bias=bias, dt_softplus=True)
        st.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            for s in states:
                K.selective_state_update(s, x, dt, A, Bm, Cm, D=D, dt_bias=bias, dt_softplus=True)
        for _ in range(3):
            graph.replay()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(st)
            for _ in range(20):
                graph.replay()
            e1.record(st)
            st.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 / LAYERS * 1e3)
    by = 2 * states[0].numel() * 4
    print(f"mode {os.environ.get('TV_SSU_MODE', 'default')}: {best:6.2f} us a launch in the graph, {by / best / 1e3:7.1f} GB/s "
          f"({by / best / 1e3 / 8000:.3f} of 8 TB/s); state err {es:.2e}, y rel err {ey:.2e}")


if __name__ == "__main__":
    main()
