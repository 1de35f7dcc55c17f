"""A/B in one process, same pipeline object and buffers: per-step run() vs overlapped run_steps()."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb, pipeline as pl
import bench
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hb.require_gpu()
batch = bench.make_batch(128, 0, torch.device('cuda', 0))
pipe = pl.GreedyPipeline(bench.FORMATS, bench.METRIC, bench.THRESHOLD, bench.SEED, chunk=16, workers=workers)
pipe.run_steps(batch for _ in range(3)); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); c0 = time.process_time()
    for _ in range(steps): pipe.run(batch)
    torch.cuda.synchronize(); t1 = time.perf_counter(); c1 = time.process_time()
    pipe.run_steps(batch for _ in range(steps)); torch.cuda.synchronize(); t2 = time.perf_counter(); c2 = time.process_time()
    print(f"rep {rep}: run() {1e3*(t1-t0)/steps:.2f} ms/step (cpu {1e3*(c1-c0)/steps:.0f} ms)   run_steps {1e3*(t2-t1)/steps:.2f} ms/step (cpu {1e3*(c2-c1)/steps:.0f} ms)", flush=True)
pipe.close()
