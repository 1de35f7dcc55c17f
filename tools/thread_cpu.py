"""CPU time per thread over N pipeline steps (who burns the CPU budget: scan pool, Python driver threads, HIP runtime threads)."""
import sys, time, os, glob, collections
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb, pipeline as pl
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
hb.require_gpu()
print(hb.bind_to_gpu_numa_node(0))
batch = bench.make_batch(128, 0, torch.device('cuda', 0))
pipe = pl.GreedyPipeline(bench.FORMATS, bench.METRIC, bench.THRESHOLD, bench.SEED, chunk=128, workers=bench.default_workers())
pipe.reserve(batch); pipe.run_steps(batch for _ in range(3)); torch.cuda.synchronize()
def snap():
    out = {}
    for d in glob.glob('/proc/self/task/*'):
        try:
            f = open(d + '/stat').read(); name = f[f.index('(') + 1:f.rindex(')')]; rest = f[f.rindex(')') + 2:].split()
            out[d.rsplit('/', 1)[1]] = (name, (int(rest[11]) + int(rest[12])) / os.sysconf('SC_CLK_TCK'))
        except OSError:
            pass
    return out
import gc; gc.collect(); gc.freeze()
a = snap(); t0 = time.perf_counter()
pipe.run_steps(batch for _ in range(steps)); torch.cuda.synchronize()
dt = time.perf_counter() - t0; b = snap()
agg = collections.Counter()
for tid, (name, t) in b.items():
    agg[name] += t - a.get(tid, (name, 0.0))[1]
main_tid = str(os.getpid())
print(f"{steps} steps, {1e3*dt/steps:.2f} ms/step wall; CPU ms/step by thread name:")
for name, t in agg.most_common(12): print(f"   {name:24s} {1e3*t/steps:7.2f}")
print(f"   main thread            {1e3*(b[main_tid][1]-a[main_tid][1])/steps:7.2f}")
pipe.close()
