"""Find where the occasional 50-60 ms step of the streamed driver goes: per-step marks, printed for slow steps only."""
import sys, time, gc
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb, pipeline as pl, pipeline_greedy as plg   # plg: where the host-scan task lives (patched below)
import bench
hb.require_gpu()
batch = bench.make_batch(128, 0, torch.device('cuda', 0))
pipe = pl.GreedyPipeline(bench.FORMATS, bench.METRIC, bench.THRESHOLD, bench.SEED, chunk=16, workers=13)
pipe.reserve(batch)
marks = []
orig_scan = plg._scan_chunk
def traced_scan(first, *a, **k):
    t0 = time.perf_counter(); r = orig_scan(first, *a, **k); marks.append(("scan", first, t0, time.perf_counter())); return r
plg._scan_chunk = traced_scan
orig_sync = torch.cuda.Event.synchronize
def traced_sync(self):
    t0 = time.perf_counter(); orig_sync(self); marks.append(("evt", None, t0, time.perf_counter()))
torch.cuda.Event.synchronize = traced_sync
orig_enq = pipe.enqueue
def traced_enq(*a, **k):
    t0 = time.perf_counter(); r = orig_enq(*a, **k); marks.append(("enq", None, t0, time.perf_counter())); return r
pipe.enqueue = traced_enq
gc_t = []
def gc_cb(phase, info):
    if phase == "start": gc_t.append([time.perf_counter(), None, info.get("generation")])
    else: gc_t[-1][1] = time.perf_counter()
gc.callbacks.append(gc_cb)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 80):
    marks.clear(); n_gc = len(gc_t)
    T0 = time.perf_counter(); pipe.run(batch); T1 = time.perf_counter()
    if (T1 - T0) > 0.02 and i > 2:
        print(f"step {i}: {(T1-T0)*1e3:.1f} ms; gc during step: {[(round((b-a)*1e3,2), g) for a,b,g in gc_t[n_gc:]]}")
        for kind, first, a, b in sorted(marks, key=lambda m: m[2]):
            print(f"   {kind:5s} {'' if first is None else first:>4} start {(a-T0)*1e3:7.2f}  end {(b-T0)*1e3:7.2f}  dur {(b-a)*1e3:6.2f}")
pipe.close()
