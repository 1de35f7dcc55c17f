"""Streamed mixed-tile-threshold (records never leave the GPU): tiles/s on batches of 4096x4096 tensors, bf16 and float32."""
import gc, sys, time
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb
from quantization_analysis_amd.pipeline import ThresholdPipeline
hb.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device='cuda'); g.manual_seed(0)
for name, x in (("bf16", (torch.randn((n, 4096, 4096), generator=g, device='cuda') * 0.02).to(torch.bfloat16)),
                ("f32", torch.randn((n // 2, 4096, 4096), generator=g, device='cuda') * 0.02)):
    for thr in (0.94, 0.9925):
        pipe = ThresholdPipeline(["bf16", "bfp8", "bfp4", "bfp2"], "pcc", thr, chunk=16)
        pipe.run(x); torch.cuda.synchronize()
        gc.collect(); gc.disable()          # a generation-2 collection of the results' objects is 30-40 ms when it lands in a run
        dts = []
        for _ in range(7):
            t0 = time.perf_counter(); res = pipe.run(x); torch.cuda.synchronize(); dts.append(time.perf_counter() - t0)
        dt = sorted(dts)[len(dts) // 2]
        gc.enable()
        tiles = x.shape[0] * 16384
        c = {k: sum(r.counts[k] for r in res) for k in res[0].counts}
        print(f"{name} thr {thr}: {dt*1e3/x.shape[0]:.3f} ms/tensor  {tiles/dt/1e6:.1f} M tiles/s  knife tiles {pipe.knife_tiles // 8}  (median of 7 runs, slowest {max(dts)*1e3/x.shape[0]:.3f})  counts {c}", flush=True)
