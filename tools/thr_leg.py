import sys, json
sys.path.insert(0, '/root/repo')
import torch, bench
dev = torch.device('cuda:0')
for _ in range(3):
    r = bench.leg_threshold(dev)
    print(json.dumps({k: r[k] for k in ('value', 'ms', 'knife_edge_tiles', 'maps_equal_oracle')}), r['roofline']['launch_ms'], flush=True)
