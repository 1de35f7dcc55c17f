"""cProfile of the bench's threshold leg (ThresholdPipeline.run_batches over the seven DeepSeek-R1 layer-0 tensors) and of its sweep leg
(sweep.sweep_tensor, 50 steps): where the HOST time of configs[2] and configs[4] goes.  python tools/legs_profile.py [threshold|sweep]"""
import cProfile, pstats, sys, time
sys.path.insert(0, '/root/repo')
import torch
import bench
from quantization_analysis_amd import hip_backend as hb
hb.require_gpu()
what = sys.argv[1] if len(sys.argv) > 1 else "threshold"
dev = torch.device("cuda:0")
names, xs = bench.deepseek_tensors(dev)
if what == "threshold":
    from quantization_analysis_amd.pipeline import ThresholdPipeline
    def as_batch(x):
        if x.dim() == 2:
            return (x[None], None)
        n = x.numel(); rows = -(-n // 32)
        m = torch.zeros((rows * 32,), dtype=x.dtype, device=x.device); m[:n] = x
        return (m.view(1, rows, 32), n)
    batches = [as_batch(x) for x in xs]
    pipe = ThresholdPipeline(bench.FORMATS, "pcc", bench.THRESHOLD, chunk=1)
    fn = lambda: pipe.run_batches(batches)
else:
    from quantization_analysis_amd.compression_algorithms.quantizer import Quantizer
    from quantization_analysis_amd.sweep import sweep_tensor
    q = Quantizer("hip")
    fn = lambda: [sweep_tensor(x, bench.FORMATS, "pcc", 0.9, 50, q)[0] for x in xs]
for _ in range(3):
    fn(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(what, "wall ms:", [round(t, 3) for t in ts])
pr = cProfile.Profile(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
