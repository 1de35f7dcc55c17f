"""cProfile of one ThresholdPipeline.run over n bf16 4096x4096 tensors (where the host time of the streamed threshold path goes)."""
import cProfile, pstats, sys
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb
from quantization_analysis_amd.pipeline import ThresholdPipeline
hb.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.9925
g = torch.Generator(device='cuda'); g.manual_seed(0)
x = (torch.randn((n, 4096, 4096), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
pipe = ThresholdPipeline(["bf16", "bfp8", "bfp4", "bfp2"], "pcc", thr, chunk=16)
pipe.run(x); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); pipe.run(x); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
