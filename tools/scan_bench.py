"""Host greedy scan (mtq_greedy_run) on the records of one 4096x4096 bf16 tensor: single-thread time, and the time per
tensor when T threads scan T tensors at once (what a chunk of the streamed pipeline does)."""
import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from quantization_analysis_amd import hip_backend as hb
import bench
hb.require_gpu()
x = bench.make_batch(1, 0, torch.device('cuda', 0))
full = hb.tile_stats(x[0], 0xF).cpu().numpy()
slim = hb.tile_stats(x[0], 0xE).cpu().numpy()
print("cpu.max:", open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else "n/a", " affinity:", len(os.sched_getaffinity(0)))
for name, st, mask in (("22-double records", full, 0xF), ("17-double + identity", slim, 0xE | hb.MASK_BF16_IDENTITY)):
    copies = [st.copy() for _ in range(16)]
    t0 = time.perf_counter()
    for s in copies: hb.greedy_run(s, mask, bench.FORMATS, "pcc", 0.999, float(4096 * 4096), 123)
    print(f"{name}: single thread {1e3 * (time.perf_counter() - t0) / 16:.3f} ms/tensor")
    for T in (8, 16, 32, 64):
        batch = np.stack([st] * T)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            hb.greedy_run_batch(batch, mask, bench.FORMATS, "pcc", 0.999, float(4096 * 4096), [123] * T, T)
            best = min(best, time.perf_counter() - t0)
        print(f"   {T} tensors on {T} threads: {1e3 * best:.3f} ms  ({1e3 * best / T:.3f} ms/tensor of wall)")
