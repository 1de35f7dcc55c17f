"""run_steps over K batches for several K, each from a drained pipeline (what bench.py's timed region sees for a given --steps):
ms per step, and the fixed cost per run that a linear fit of the totals gives."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch, gc
from quantization_analysis_amd import hip_backend as hb, pipeline as pl
import bench
hb.require_gpu(); hb.bind_to_gpu_numa_node(0)
batch = bench.make_batch(128, 0, torch.device('cuda', 0))
pipe = pl.GreedyPipeline(bench.FORMATS, bench.METRIC, bench.THRESHOLD, bench.SEED, chunk=32, workers=16)
pipe.reserve(batch)
pipe.run_steps(batch for _ in range(5))
gc.collect(); gc.freeze()
tot = {}
for rep in range(3):
    for K in (1, 2, 5, 10, 20, 50):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); pipe.run_steps(batch for _ in range(K)); torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
        tot.setdefault(K, []).append(dt)
for K, v in tot.items():
    print(f"K={K:3d}: total {min(v):7.2f} ms (best of 3)  {min(v) / K:6.2f} ms/step")
k1, k2 = 10, 50
slope = (min(tot[k2]) - min(tot[k1])) / (k2 - k1)
print(f"steady step {slope:.2f} ms; fixed cost per run {min(tot[k1]) - slope * k1:.2f} ms")
pipe.close()
