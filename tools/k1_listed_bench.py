"""The listed K1 launch alone (mtq_tile_stats_listed on bf16 storage → tile_stats_bf16_rolled<4,6,·,LISTED>): N 4096x4096 bf16 tensors, a random
`frac` of the tiles listed in ascending order (what phase 1 of the split search hands over: 15 % at pcc >= 0.999), bfp2 whole + bfp4's
Σ|x−y| / max|x−y| evaluated for the listed tiles; the written statistics against a whole-record launch, bit for bit, and HIP-event timing.
python tools/k1_listed_bench.py [n] [reps] [frac]"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 11
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.154
hb.require_gpu()
g = torch.Generator(device='cuda'); g.manual_seed(0)
x = (torch.randn((n, 4096, 4096), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
tiles = 128 * 128
layout, full, sums = 0xE, 0x2, 0x4
ref = hb.tile_stats_batched(x, layout)
got = hb.tile_stats_partial(x, layout, full, sums)
pick = torch.rand((n * tiles,), generator=g, device='cuda') < frac
listed = torch.nonzero(pick).flatten().to(torch.int32)
k = int(listed.numel())
n_listed = torch.tensor([k], dtype=torch.int32, device='cuda')
scratch = torch.zeros((k + 1,), dtype=torch.int32, device='cuda')
hb.tile_stats_listed(x, layout, 0x8, 0x4, listed, n_listed, got, scratch)
torch.cuda.synchronize()
r = ref.view(torch.int64).reshape(n * tiles, -1)[listed.long()]
q = got.view(torch.int64).reshape(n * tiles, -1)[listed.long()]
bad = int((r != q).sum().item())
print(f"{k} listed tiles of {n * tiles} ({k / (n * tiles):.3f}): {bad} differing values in their records after the listed launch")
ts = []
for _ in range(reps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); hb.tile_stats_listed(x, layout, 0x8, 0x4, listed, n_listed, got, scratch); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"listed K1 n={n}: median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms  {k / ts[len(ts) // 2] / 1e3:.1f} M listed tiles/s  "
      f"{2048 * k / ts[len(ts) // 2] / 1e6 / 8000:.4f} of the read roofline on the listed tiles' bytes")
sys.exit(1 if bad else 0)
