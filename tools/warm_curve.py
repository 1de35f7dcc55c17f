"""Per-step wall time of the first 40 steps in a fresh process (how long the pipeline takes to reach its steady state)."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb, pipeline as pl
import bench
hb.require_gpu()
batch = bench.make_batch(128, 0, torch.device('cuda', 0))
pipe = pl.GreedyPipeline(bench.FORMATS, bench.METRIC, bench.THRESHOLD, bench.SEED, chunk=16, workers=int(sys.argv[1]) if len(sys.argv) > 1 else 16)
pipe.reserve(batch)
torch.cuda.synchronize()
ts = []
for i in range(40):
    t0 = time.perf_counter(); pipe.run(batch); ts.append(1e3 * (time.perf_counter() - t0))
print("run() per step:", " ".join(f"{t:.1f}" for t in ts))
t0 = time.perf_counter(); pipe.run_steps(batch for _ in range(20)); torch.cuda.synchronize(); print(f"then run_steps x20: {1e3*(time.perf_counter()-t0)/20:.2f} ms/step")
pipe.close()
