"""The bench's latency and Llama legs alone (python tools/legs_ab.py [reps]): for A/B runs under environment switches (MTQ_LIB=...), one
process per setting; the first repetition of a process is the cold one."""
import sys
sys.path.insert(0, '/root/repo')
import torch
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
sample = bench.make_batch(1, 0, dev)
for _ in range(reps):
    q = bench.leg_latency(sample, dev)
    print(f"latency: algo.run {q['algo_run_ms']:.3f} ms, pipeline batch of one {q['pipeline_batch_of_one_ms']:.3f} ms", flush=True)
    r = bench.leg_llama(dev)
    print(f"llama3_8b: {r['value'] / 1e6:.1f} M tiles/s, {r['pipeline_ms']:.3f} ms per model (all: {[round(v, 2) for v in r['pipeline_ms_all']]})", flush=True)
