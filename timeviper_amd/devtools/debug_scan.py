import sys, math
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from timeviper_amd import kernels as K
from oracle import ops as R
torch.manual_seed(0)
def run(L, H, P, G, N=128, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, L, H, P, generator=g).bfloat16()
    dt = (torch.randn(1, L, H, generator=g) * 0.5).bfloat16()
    A = -(torch.rand(H, generator=g) * 15 + 1)
    Bm = (torch.randn(1, L, G, N, generator=g) * 0.5).bfloat16()
    Cm = (torch.randn(1, L, G, N, generator=g) * 0.5).bfloat16()
    D = torch.rand(H, generator=g) + 0.5
    dtb = torch.full((H,), -1.0)
    y_ref, fin_ref, dec_ref = R.ssd_recurrence_ref(x.float(), dt.float(), A, Bm.float(), Cm.float(), D=D, dt_bias=dtb)
    d = lambda t: t.cuda()
    y, fin, dec = K.mamba_chunk_scan_combined(d(x), d(dt), d(A), d(Bm), d(Cm), chunk_size=64, D=d(D), dt_bias=d(dtb), dt_softplus=True, return_final_states=True, return_total_decay=True)
    y = y.float().cpu(); fin = fin.cpu(); dec = dec.cpu()
    ey = (y - y_ref).abs()
    tol = 2e-2 + 2e-2 * y_ref.abs()
    bad = ey > tol
    print(f"L={L} H={H} P={P} G={G}: y bad {bad.sum().item()}/{bad.numel()} max {ey.max():.3e} | fin relerr {((fin-fin_ref).norm()/fin_ref.norm()).item():.3e} | decay err {(dec-dec_ref).abs().max():.3e}")
    if bad.any():
        bt = bad[0].any(-1).any(-1)  # per t
        print("  bad t:", bt.nonzero().flatten().tolist()[:40])
        bh = bad[0].any(0).any(-1); print("  bad h:", bh.nonzero().flatten().tolist())
        bp = bad[0].any(0).any(0); print("  bad p:", bp.nonzero().flatten().tolist())
        t0 = bt.nonzero().flatten()[0].item(); h0 = bh.nonzero().flatten()[0].item()
        print("  y[t0,h0,:8]   ", y[0, t0, h0, :8].tolist())
        print("  ref[t0,h0,:8] ", y_ref[0, t0, h0, :8].float().tolist())
    ef = (fin - fin_ref).abs()
    if ef.max() > 1e-2:
        print("  fin bad per h:", (ef.amax((2, 3))[0] > 1e-2).nonzero().flatten().tolist())
for cfg in [(16, 8, 80, 1), (64, 8, 80, 1), (65, 8, 80, 1), (128, 8, 80, 1), (192, 8, 80, 2), (300, 16, 80, 2), (129, 8, 64, 8)]:
    run(*cfg)
