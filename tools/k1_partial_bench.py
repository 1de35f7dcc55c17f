"""K1 with partial records (mtq_tile_stats_partial): the promised statistics against a full launch's, bit for bit, and the timing of
both on N 4096x4096 bf16 tensors (HIP events).  python tools/k1_partial_bench.py [n] [reps] [layout] [full] [sums]"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
layout = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0xE
full = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0x2
sums = int(sys.argv[5], 0) if len(sys.argv) > 5 else 0x4
hb.require_gpu()
g = torch.Generator(device='cuda'); g.manual_seed(0)
x = (torch.randn((n, 4096, 4096), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
x[0, :64, :256] = 0                       # an all-zero corner
x[0, 100, 300] = 3.0e4                    # a wide exponent spread inside a group (tail class)
x[1, 7, 9] = float('inf')                 # a tile the exact route hands to the literal fix-up
ref = hb.tile_stats_batched(x, layout)
got = hb.tile_stats_partial(x, layout, full, sums)
torch.cuda.synchronize()
r = ref.view(torch.int64); q = got.view(torch.int64)
cols = [0, 1]
slot = 0
for f in range(4):
    if not layout & (1 << f):
        continue
    o = 2 + 5 * slot
    if full & (1 << f):
        cols += list(range(o, o + 5))
    elif sums & (1 << f):
        cols += list(range(o, o + 3))
    slot += 1
bad = (r[..., cols] != q[..., cols]).sum().item()
rest = [c for c in range(ref.shape[-1]) if c not in cols]
print(f"layout {layout:#x} full {full:#x} sums {sums:#x}: promised columns {cols}: {bad} differing values; "
      f"unpromised columns {rest}: {int(torch.isnan(got[..., rest]).sum().item())} NaN of {got[..., rest].numel()}")


def timeit(fn):
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


tiles = n * 128 * 128
for name, fn in (("full   ", lambda: hb.tile_stats_batched(x, layout, out=ref)), ("partial", lambda: hb.tile_stats_partial(x, layout, full, sums, out=got))):
    ms, mn = timeit(fn)
    print(f"K1 {name} n={n}: median {ms:.3f} ms  min {mn:.3f} ms  {tiles / ms / 1e6 * 1e3:.1f} M tiles/s  frac {2048 * tiles / ms / 1e6 / 8000:.4f}")
sys.exit(1 if bad else 0)
