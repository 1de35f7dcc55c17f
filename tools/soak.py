import sys, time, resource
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb, pipeline as pl
import bench
hb.require_gpu(); hb.bind_to_gpu_numa_node(0)
batch = bench.make_batch(128, 0, torch.device('cuda', 0))
pipe = pl.GreedyPipeline(bench.FORMATS, bench.METRIC, bench.THRESHOLD, bench.SEED, chunk=128, workers=16)   # the bench configuration: one K1 launch per step, the lazy route
pipe.reserve(batch)
ref = None
for rnd in range(6):
    t0 = time.perf_counter()
    res = pipe.run_steps(batch for _ in range(400))
    dt = time.perf_counter() - t0
    sig = (sum(r.counts['bfp8'] for r in res), sum(r.counts['bfp4'] for r in res), round(sum(r.pcc for r in res), 12))
    ref = ref or sig
    assert sig == ref, (sig, ref)
    print(f"round {rnd}: {dt / 400 * 1e3:.3f} ms/step  maxrss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:.0f} MiB  gpu alloc {torch.cuda.memory_allocated() / 2**20:.0f} MiB  {sig}", flush=True)
pipe.close()
