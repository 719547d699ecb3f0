import sys, torch
sys.path.insert(0, ".")
from timeviper_amd import kernels as K
dev = torch.device("cuda", 0)
def run(B, L, H=16, D=72, n=20):
    qkv = torch.randn(B, L, 3, H, D, device=dev, dtype=torch.bfloat16)
    q, k, v = qkv.unbind(2)
    for _ in range(3): K.flash_attn_func(q, k, v)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): K.flash_attn_func(q, k, v)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for k in (1, 2, 4, 8):
    L = 729 * k
    B = 256 // k
    wg = B * 16 * ((L + 255) // 256)
    tiles = (L + 95) // 96
    ms = run(B, L)
    print(f"L={L} B={B} WGs={wg} tiles/WG={tiles} ms={ms:.3f} us/WG-slot={ms*1e3/(wg/256):.2f}")
