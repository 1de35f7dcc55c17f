"""CPU time (all threads) a pipeline step costs, against its wall time — under a cgroup CPU quota the former is the budget."""
import sys, time, os
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb, pipeline as pl
import bench
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 32
hb.require_gpu()
hb.bind_to_gpu_numa_node(0)
batch = bench.make_batch(128, 0, torch.device('cuda', 0))
pipe = pl.GreedyPipeline(bench.FORMATS, bench.METRIC, bench.THRESHOLD, bench.SEED, chunk=chunk, workers=workers)
pipe.reserve(batch)
pipe.run_steps(batch for _ in range(5)); torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
def thr():
    s = open('/sys/fs/cgroup/cpu.stat').read().split()
    d = dict(zip(s[0::2], s[1::2])); return int(d.get('nr_throttled', 0)), int(d.get('throttled_usec', 0)), int(d.get('usage_usec', 0))
t0, c0, q0 = time.perf_counter(), time.process_time(), thr()
pipe.run_steps(batch for _ in range(steps)); torch.cuda.synchronize()
t1, c1, q1 = time.perf_counter(), time.process_time(), thr()
print(f"workers {workers}: wall {1e3*(t1-t0)/steps:.2f} ms/step, process CPU {1e3*(c1-c0)/steps:.1f} ms/step ({(c1-c0)/(t1-t0):.1f} CPUs busy), "
      f"cgroup usage {(q1[2]-q0[2])/1e3/steps:.1f} ms/step, throttled {q1[0]-q0[0]} periods / {(q1[1]-q0[1])/1e3:.1f} ms")
pipe.close()
