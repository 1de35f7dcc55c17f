"""Timeline of overlapped steps (run_steps): driver-thread calls, K1 launches (GPU time mapped onto the host clock), record
arrival and scan completion per chunk — where the step period comes from."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch, gc
from quantization_analysis_amd import hip_backend as hb, pipeline as pl, pipeline_greedy as plg   # plg: where the host-scan task lives (patched below)
import bench
hb.require_gpu(); hb.bind_to_gpu_numa_node(0)
batch = bench.make_batch(128, 0, torch.device('cuda', 0))
pipe = pl.GreedyPipeline(bench.FORMATS, bench.METRIC, bench.THRESHOLD, bench.SEED, chunk=32, workers=16)
pipe.reserve(batch)
pipe.run_steps(batch for _ in range(6))
gc.collect(); gc.freeze()
marks = []
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); marks.append((label, t0, time.perf_counter())); return r
    setattr(obj, name, g)
wrap(pipe, "enqueue", "main enqueue"); wrap(pipe, "finish", "main finish"); wrap(pipe, "_launch_columns", "main   columns"); wrap(pipe, "resolve", "main resolve")
orig_scan = plg._scan_chunk
def traced_scan(first, *a, **k):
    t0 = time.perf_counter(); r = orig_scan(first, *a, **k); marks.append((f"scan chunk@{first}", t0, time.perf_counter())); return r
plg._scan_chunk = traced_scan
orig_sync = torch.cuda.Event.synchronize
def traced_sync(self):
    t0 = time.perf_counter(); orig_sync(self); marks.append(("main   wait records", t0, time.perf_counter()))
torch.cuda.Event.synchronize = traced_sync
torch.cuda.synchronize()
ref = torch.cuda.Event(enable_timing=True); ref.record(pipe.stream); ref.synchronize(); T0 = time.perf_counter()
n0 = len(pipe.timing.events)
pipe.run_steps(batch for _ in range(8)); torch.cuda.synchronize(); T1 = time.perf_counter()
torch.cuda.Event.synchronize = orig_sync
for e0, e1, _ in list(pipe.timing.events)[n0:]:
    marks.append(("gpu K1", T0 + ref.elapsed_time(e0) * 1e-3, T0 + ref.elapsed_time(e1) * 1e-3))
print(f"8 steps: {(T1 - T0) * 1e3 / 8:.3f} ms/step")
lo, hi = T0 + 3 * (T1 - T0) / 8, T0 + 5.2 * (T1 - T0) / 8
for label, a, b in sorted(marks, key=lambda m: m[1]):
    if lo <= a <= hi:
        print(f"{(a - T0) * 1e3:8.3f} .. {(b - T0) * 1e3:8.3f}  ({(b - a) * 1e3:6.3f})  {label}")
pipe.close()
