"""Device-resident greedy scan (csrc/mtq_scan.hip) timing: N 4096x4096 bf16 tensors, K1 once, then the scan kernel alone
(HIP events), and the host scan on the same records for comparison."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from quantization_analysis_amd import hip_backend as hb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
cols = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
hb.require_gpu()
ALL = ["bf16", "bfp8", "bfp4", "bfp2"]
g = torch.Generator(device='cuda'); g.manual_seed(0)
x = (torch.randn((n, rows, cols), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
recs = hb.tile_stats_batched(x, 0xE)
T = recs.shape[1]
seed_list = [123 + 7919 * i for i in range(n)]   # tensor 0: the headline case's seed; the others distinct
sd = torch.tensor(seed_list, dtype=torch.int64, device='cuda')
dec = 0xE | hb.MASK_BF16_IDENTITY
maps = torch.empty((n, T), dtype=torch.int8, device='cuda'); status = torch.empty((n,), dtype=torch.int32, device='cuda')
scratch = torch.empty((int(hb.lib().mtq_greedy_scan_scratch_bytes(n, T)),), dtype=torch.uint8, device='cuda')
hb.greedy_scan_device(recs, dec, ALL, "pcc", 0.999, float(rows * cols), sd, maps, status, scratch); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); hb.greedy_scan_device(recs, dec, ALL, "pcc", 0.999, float(rows * cols), sd, maps, status, scratch); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"device scan: {n} tensors x {T} tiles, one wave each: median {ts[len(ts)//2]:.3f} ms per launch (min {ts[0]:.3f}); status {status.cpu().unique().tolist()}")
h = recs.cpu().numpy()
t0 = time.perf_counter(); want, _c, _o = hb.greedy_run_batch(h, dec, ALL, "pcc", 0.999, float(rows * cols), seed_list, 16); dt = time.perf_counter() - t0
print(f"host scan, 16 threads: {dt*1e3:.2f} ms for {n} tensors; maps equal: {np.array_equal(want, maps.cpu().numpy())}")
import ctypes
tk = (ctypes.c_uint64 * 16)()
hb.check(hb.lib().mtq_debug_scan_ticks(tk))
names = ["init sums", "base pass + draw-only shuffle"] + [f"pass {p}: {w}" for p in (1, 2, 3) for w in ("candidates + shuffle", "deltas", "visits")]
tot = sum(tk[:11])
print("phase ticks of tensor 0 (shader clock):", ", ".join(f"{nm} {tk[i]/1e3:.0f}k" for i, nm in enumerate(names)), f"| total {tot/1e3:.0f}k")
if any(tk[12:16]):   # a library built with -DMTQ_SCAN_PROFILE: inside the shuffle batches of tensor 0
    nb = max(int(tk[15]), 1)
    print(f"shuffle batches of tensor 0: {nb}; per batch: draws + acceptance {tk[12]/nb:.0f}, swaps {tk[13]/nb:.0f}, hand-back {tk[14]/nb:.0f} ticks")
