"""Chain-record stage on its own: pack kernel and the sequential initial-sum kernel over a chunk of 16 x 4096^2 records (HIP events)."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb
import bench
hb.require_gpu()
x = bench.make_batch(16, 0, torch.device('cuda', 0))
recs = hb.tile_stats_batched(x, 0xE)
mask = 0xE | hb.MASK_BF16_IDENTITY
chain, base = hb.pack_chain_records(recs, mask, bench.FORMATS)
torch.cuda.synchronize()
import ctypes
fm = (ctypes.c_int * 4)(0, 1, 2, 3)
def timed(fn, reps=10):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
T = recs.shape[1]
print("pack_chain_records (16 tensors): %.3f ms" % timed(lambda: hb.check(hb.lib().mtq_pack_chain_records(recs.data_ptr(), 16 * T, mask, fm, 4, chain.data_ptr(), base.data_ptr(), 2, hb._stream_ptr()))))
print("pack_slim_records (16 tensors): %.3f ms" % timed(lambda: hb.pack_slim_records(recs, 0xE)))
