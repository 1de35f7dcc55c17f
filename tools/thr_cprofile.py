"""Host-side profile of the threshold leg's run_batches (cProfile over `reps` calls): python tools/thr_cprofile.py [reps]"""
import cProfile, pstats, sys, time
sys.path.insert(0, '/root/repo')
import torch
import bench
from quantization_analysis_amd.pipeline import ThresholdPipeline

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
names, xs = bench.deepseek_tensors(dev)
def as_batch(x):
    if x.dim() == 2:
        return (x[None], None)
    n = x.numel(); rows = -(-n // 32)
    m = torch.zeros((rows * 32,), dtype=x.dtype, device=x.device); m[:n] = x
    return (m.view(1, rows, 32), n)
batches = [as_batch(x) for x in xs]
with ThresholdPipeline(bench.FORMATS, "pcc", bench.THRESHOLD, chunk=1) as pipe:
    for _ in range(5):
        pipe.run_batches(batches)
    t0 = time.perf_counter()
    for _ in range(reps):
        pipe.run_batches(batches)
    print(f"plain: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        pipe.run_batches(batches)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(28)
