"""The threshold leg's run_batches alone, for a kernel trace: python tools/thr_trace.py [reps]  — DeepSeek-R1 layer 0's seven tensors
(bench.deepseek_tensors), `reps` calls, host wall time per call printed.  Under rocprofv3 --kernel-trace, tools/thr_timeline.py prints
the last call's dispatches."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
import bench
from quantization_analysis_amd.pipeline import ThresholdPipeline

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
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
    for i in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.run_batches(batches)
        print(f"call {i}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
