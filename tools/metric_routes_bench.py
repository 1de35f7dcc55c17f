"""mixed-tile-greedy through GreedyPipeline for each metric, with the search on the device and with the host scan:
tiles/s over 128 x 4096x4096 bf16 per step.  python tools/metric_routes_bench.py [tensors] [steps]"""
import gc, sys, time
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb
from quantization_analysis_amd.pipeline import GreedyPipeline, default_workers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
hb.require_gpu()
g = torch.Generator(device='cuda'); g.manual_seed(0)
x = (torch.randn((n, 4096, 4096), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
ALL = ["bf16", "bfp8", "bfp4", "bfp2"]
for metric, thr in (("pcc", 0.999), ("mae", 3e-4), ("atol", 4e-3)):
    for scan in ("device", "host"):
        pipe = GreedyPipeline(ALL, metric, thr, 123, chunk=n if scan == "device" else 32, workers=default_workers(), scan=scan)
        pipe.reserve(x)
        gc.collect(); gc.freeze()       # as bench.py: no full collection inside the timed steps
        pipe.run_steps(x for _ in range(8))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = pipe.run_steps(x for _ in range(steps))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        c = {k: sum(r.counts[k] for r in res) for k in res[0].counts}
        print(f"{metric} {thr} scan={scan}: {n * 16384 * steps / dt / 1e6:.1f} M tiles/s, {dt / steps * 1e3:.3f} ms/step, counts of the last step {c}", flush=True)
        pipe.close()
