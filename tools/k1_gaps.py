"""Idle time of the K1 stream between consecutive steps of the device-scan pipeline (HIP events of the launches themselves:
no profiler in the way).  python tools/k1_gaps.py [tensors] [steps]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from quantization_analysis_amd import hip_backend as hb
from quantization_analysis_amd.pipeline import GreedyPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
hb.require_gpu()
g = torch.Generator(device='cuda'); g.manual_seed(0)
x = (torch.randn((n, 4096, 4096), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
pipe = GreedyPipeline(["bf16", "bfp8", "bfp4", "bfp2"], "pcc", 0.999, 123, chunk=n)
pipe.reserve(x)
import gc
gc.collect(); gc.freeze()
pipe.run_steps(x for _ in range(warm))
pipe.timing.drain(); pipe.timing.__init__()
import os
if os.environ.get("K1_KEEP_BUSY"):
    k1_mask = pipe._layout(x)[0]
    tmp = hb.tile_stats_batched(x, k1_mask)
    for _ in range(int(os.environ["K1_KEEP_BUSY"])):
        hb.tile_stats_batched(x, k1_mask, out=tmp)
keep = []
class Keep(list):   # drain() consumes the events: keep a second reference
    def append(self, item):
        keep.append(item); list.append(self, item)
pipe.timing.events = Keep()
torch.cuda.synchronize(); t0 = time.perf_counter()
pipe.run_steps(x for _ in range(steps))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
dur = [a.elapsed_time(b) for a, b, _t in keep]
gap = [keep[i][1].elapsed_time(keep[i + 1][0]) for i in range(len(keep) - 1)]
per = [keep[i][0].elapsed_time(keep[i + 1][0]) for i in range(len(keep) - 1)]
print(f"{steps} steps of {n} tensors: {dt / steps * 1e3:.3f} ms per step; K1 launch {np.median(dur):.3f} ms (median), "
      f"gap to the next launch {np.median(gap) * 1e3:.0f} us (median; min {min(gap) * 1e3:.0f}, max {max(gap) * 1e3:.0f}), period {np.median(per):.3f} ms")
print("launch durations (ms):", " ".join(f"{d_:.2f}" for d_ in dur))
print("gaps (us):", " ".join(f"{g_ * 1e3:.0f}" for g_ in gap))
pipe.close()
