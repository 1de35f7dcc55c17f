"""mixed-tile-threshold and the 50-step threshold sweep on one 4096x4096 tensor through the hip backend (records stay on
the device: K1 -> K4 / scores / column sums on the device)."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb
from quantization_analysis_amd.compression_algorithms import create_algorithm
from quantization_analysis_amd.compression_algorithms.quantizer import Quantizer
from quantization_analysis_amd.sweep import sweep_tensor
hb.require_gpu()
q = Quantizer("hip")
ALL = ["bf16", "bfp8", "bfp4", "bfp2"]
g = torch.Generator(device='cuda'); g.manual_seed(0)
for name, x in (("bf16", (torch.randn((4096, 4096), generator=g, device='cuda') * 0.02).to(torch.bfloat16)),
                ("f32", torch.randn((4096, 4096), generator=g, device='cuda') * 0.02)):
    algo = create_algorithm("mixed-tile-threshold", {"metric": "pcc", "threshold": 0.9925, "materialize_y": False})
    algo.run(x, ALL, q, None); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): r = algo.run(x, ALL, q, None)[0]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"threshold {name}: {dt*1e3:.2f} ms/tensor  counts {r.tile_counts} knife {r.meta['knife_edge_tiles']}", flush=True)
    sweep_tensor(x, ALL, "pcc", 0.9, 50, q); torch.cuda.synchronize()
    t0 = time.perf_counter(); rows, base, thr = sweep_tensor(x, ALL, "pcc", 0.9, 50, q); torch.cuda.synchronize()
    print(f"sweep 50 steps {name}: {(time.perf_counter()-t0)*1e3:.1f} ms  ({len(rows)} rows)", flush=True)
