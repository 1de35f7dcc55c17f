"""K2 / K3 / K5 timing on one 4096x4096 tensor (HIP events on the launch stream)."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from quantization_analysis_amd import hip_backend as hb
hb.require_gpu()
g = torch.Generator(device='cuda'); g.manual_seed(0)
x = (torch.randn((8192, 8192), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
amap = torch.randint(0, 4, (256, 256), device='cuda', dtype=torch.int8)
out = torch.empty((8192, 8192), dtype=torch.float32, device='cuda')
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
el = 8192 * 8192
for name, fn, bytes_ in (("K2 quantize bfp8 (bf16 in)", lambda: hb.quantize(x, "bfp8", out=out), el * 6),
                         ("K3 apply_assignment (bf16 in)", lambda: hb.apply_assignment(x, amap, out=out), el * 6)):
    ms = t(fn); print(f"{name}: {ms:.3f} ms  {bytes_/ms/1e6:.0f} GB/s  frac {bytes_/ms/1e6/8000:.3f}")
xf = x.float()
ms = t(lambda: hb.quantize(xf, "bfp4", out=out)); print(f"K2 quantize bfp4 (fp32 in): {ms:.3f} ms  {el*8/ms/1e6:.0f} GB/s")
w = torch.randint(0, 256, (8192, 8192), device='cuda', dtype=torch.uint8); sc = torch.rand((64, 64), device='cuda') + 0.5
ms = t(lambda: hb.dequant_fp8_block(w, sc)); print(f"K5 dequant_fp8_block: {ms:.3f} ms  {el*5/ms/1e6:.0f} GB/s (incl. output alloc)")
ms = t(lambda: hb.tile_stats(xf, 0xF)); print(f"K1 generic fp32 8192^2: {ms:.3f} ms  {65536/ms/1e3:.1f} M tiles/s  {el*4/ms/1e6:.0f} GB/s")
