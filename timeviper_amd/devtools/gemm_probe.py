import os, sys, time
os.environ.setdefault("PYTORCH_TUNABLEOP_ROCBLAS_ENABLED", "0")
import torch
dev = torch.device("cuda", 0)
def bench(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = frames * 729
shapes = [(1152, 3456), (1152, 1152), (1152, 4304), (4304, 1152)]
res = {}
for K_, N in shapes:
    x = torch.randn(M, K_, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K_, device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: torch.addmm(b, x, w.t(), out=y)
    ms = bench(f)
    res[(K_, N)] = ms
    print(f"default  M={M} K={K_} N={N}: {ms:.3f} ms  {2*M*N*K_/ms/1e9:.0f} TFLOP/s", flush=True)
if len(sys.argv) > 2:
    tn = torch.cuda.tunable
    tn.enable(True); tn.tuning_enable(True)
    tn.set_max_tuning_duration(40); tn.set_max_tuning_iterations(3); tn.set_rotating_buffer_size(0)
    tn.set_filename("gpurun_out/tune_probe.csv")
    for K_, N in shapes:
        x = torch.randn(M, K_, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K_, device=dev, dtype=torch.bfloat16) * 0.02
        b = torch.randn(N, device=dev, dtype=torch.bfloat16)
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        f = lambda: torch.addmm(b, x, w.t(), out=y)
        t0 = time.perf_counter(); f(); torch.cuda.synchronize()
        print(f"  tuned in {time.perf_counter()-t0:.1f} s", flush=True)
        ms = bench(f)
        print(f"tuned    M={M} K={K_} N={N}: {ms:.3f} ms  {2*M*N*K_/ms/1e9:.0f} TFLOP/s  ({res[(K_,N)]/ms:.3f}x)", flush=True)
    for r in tn.get_results(): print(r)
