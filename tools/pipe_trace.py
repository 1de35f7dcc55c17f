"""Where one GreedyPipeline step spends its time: enqueue, per-chunk records-on-host times, per-chunk scan completion."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb, pipeline as pl, pipeline_greedy as plg   # plg: where the host-scan task lives (patched below)
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
hb.require_gpu()
batch = bench.make_batch(n, 0, torch.device('cuda', 0))
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 16
pipe = pl.GreedyPipeline(bench.FORMATS, bench.METRIC, bench.THRESHOLD, bench.SEED, chunk=chunk, workers=workers)
for _ in range(2): pipe.run(batch)
torch.cuda.synchronize()

orig_scan = plg._scan_chunk
marks = []
def traced_scan(first, *a, **k):
    t0 = time.perf_counter(); r = orig_scan(first, *a, **k); marks.append(("scan", first, t0, time.perf_counter())); return r
plg._scan_chunk = traced_scan
orig_sync = torch.cuda.Event.synchronize
def traced_sync(self):
    t0 = time.perf_counter(); orig_sync(self); marks.append(("evt", None, t0, time.perf_counter()))
torch.cuda.Event.synchronize = traced_sync
for rep in range(3):
    marks.clear()
    T0 = time.perf_counter(); res = pipe.run(batch); T1 = time.perf_counter()
    print(f"step {rep}: {(T1-T0)*1e3:.2f} ms")
    for kind, first, a, b in sorted(marks, key=lambda m: m[2]):
        print(f"   {kind:5s} {'' if first is None else first:>4} start {(a-T0)*1e3:7.2f}  end {(b-T0)*1e3:7.2f}  dur {(b-a)*1e3:6.2f}")
pipe.close()

# overlapped steps (run_steps): scans of batch s against the GPU work of batch s+1
pipe = pl.GreedyPipeline(bench.FORMATS, bench.METRIC, bench.THRESHOLD, bench.SEED, chunk=chunk, workers=workers)
pipe.run_steps(batch for _ in range(2))
torch.cuda.synchronize()
marks.clear()
T0 = time.perf_counter(); pipe.run_steps(batch for _ in range(4)); torch.cuda.synchronize(); T1 = time.perf_counter()
print(f"run_steps x4: {(T1-T0)*1e3:.2f} ms  ({(T1-T0)*1e3/4:.2f} ms/step)")
for kind, first, a, b in sorted(marks, key=lambda m: m[2]):
    print(f"   {kind:5s} {'' if first is None else first:>4} start {(a-T0)*1e3:7.2f}  end {(b-T0)*1e3:7.2f}  dur {(b-a)*1e3:6.2f}")
pipe.close()
