"""Main-thread time budget of GreedyPipeline.run_steps: where the driver thread spends a step (enqueue, waiting for record
events, waiting for scans, launching / resolving the device-side columns), bench.py's configuration."""
import sys, time, collections
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb, pipeline as pl
import bench
hb.require_gpu()
hb.bind_to_gpu_numa_node(0)
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
batch = bench.make_batch(128, 0, torch.device('cuda', 0))
pipe = pl.GreedyPipeline(bench.FORMATS, bench.METRIC, bench.THRESHOLD, bench.SEED, chunk=chunk, workers=16)
pipe.reserve(batch)
pipe.run_steps(batch for _ in range(5))
acc = collections.Counter()
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[key] += time.perf_counter() - t0; return r
    setattr(obj, name, g)
wrap(pipe, "enqueue", "enqueue"); wrap(pipe, "finish", "finish (total)"); wrap(pipe, "_launch_columns", "  launch columns"); wrap(pipe, "resolve", "resolve")
orig_sync = torch.cuda.Event.synchronize
def traced_sync(self):
    t0 = time.perf_counter(); orig_sync(self); acc["  event waits"] += time.perf_counter() - t0
torch.cuda.Event.synchronize = traced_sync
import concurrent.futures as cf
orig_res = cf.Future.result
def traced_res(self, *a, **k):
    t0 = time.perf_counter(); r = orig_res(self, *a, **k); acc["  future waits"] += time.perf_counter() - t0; return r
cf.Future.result = traced_res
import gc; gc.collect(); gc.freeze()
torch.cuda.synchronize()
T0 = time.perf_counter(); pipe.run_steps(batch for _ in range(steps)); torch.cuda.synchronize(); T1 = time.perf_counter()
print(f"chunk {chunk}: {(T1 - T0) * 1e3 / steps:.3f} ms/step")
ev = list(pipe.timing.events)[-(steps * (128 // chunk)):]
gaps = [ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(len(ev) - 1)]
durs = [a.elapsed_time(b) for a, b, _ in ev]
per = 128 // chunk
inner = [g for i, g in enumerate(gaps) if (i + 1) % per != 0]
outer = [g for i, g in enumerate(gaps) if (i + 1) % per == 0]
print(f"  K1 launches: mean {sum(durs) / len(durs):.3f} ms; gap to the next launch inside a step: mean {sum(inner) / max(len(inner), 1):.3f} ms, "
      f"across steps: mean {sum(outer) / max(len(outer), 1):.3f} ms (max {max(outer):.3f})")
for k, v in acc.items():
    print(f"  {k:18s} {v * 1e3 / steps:7.3f} ms/step")
pipe.close()
