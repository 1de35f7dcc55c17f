"""K1-only timing: batched tile_stats over N 4096x4096 bf16 tensors (HIP events on the launch stream)."""
import sys, os, time
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mask = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0xE   # what bench.py launches for bf16 storage (the bf16 candidate is the identity)
hb.require_gpu()
g = torch.Generator(device='cuda'); g.manual_seed(0)
x = (torch.randn((n, 4096, 4096), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
out = hb.tile_stats_batched(x, mask)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); hb.tile_stats_batched(x, mask, out=out); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
ms = ts[len(ts)//2]
tiles = n * 128 * 128
print(f"K1 n={n} mask={mask:#x}: median {ms:.3f} ms  min {ts[0]:.3f} ms  {tiles/ms/1e6*1e3:.1f} M tiles/s  {2048*tiles/ms/1e6:.1f} GB/s read  frac {2048*tiles/ms/1e6/8000:.4f}")
