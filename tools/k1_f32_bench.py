"""K1 timing on float32 storage (the route gpt2 / dequantised DeepSeek tensors take): batched tile_stats over N 4096x4096
float32 tensors, HIP events on the launch stream.  usage: k1_f32_bench.py [n] [reps] [kind: normal|fp8block]"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kind = sys.argv[3] if len(sys.argv) > 3 else "normal"
hb.require_gpu()
g = torch.Generator(device='cuda'); g.manual_seed(0)
x = torch.randn((n, 4096, 4096), generator=g, device='cuda') * 0.02
if kind == "fp8block":
    m, e = torch.frexp(torch.randn((n, 4096, 4096), generator=g, device='cuda'))
    base = torch.ldexp(torch.round(m * 16) / 16, e)
    sc = torch.exp(torch.randn((n, 32, 32), generator=g, device='cuda')) * 0.01
    x = base * sc.repeat_interleave(128, 1).repeat_interleave(128, 2)
for mask in (0xF, 0xE, 0x6, 0x2):
    out = hb.tile_stats_batched(x, mask)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); hb.tile_stats_batched(x, mask, out=out); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    ms = ts[len(ts) // 2]
    tiles = n * 128 * 128
    print(f"K1 f32 {kind} mask={mask:#x} n={n}: median {ms:.3f} ms  {tiles/ms/1e6*1e3:.1f} M tiles/s  {4096*tiles/ms/1e6:.1f} GB/s read  frac {4096*tiles/ms/1e6/8000:.4f}", flush=True)
