"""Shader-clock split of one tensor's search (a -DMTQ_SCAN_PROFILE build: MTQ_LIB=build/libmtq_prof.so): phase stamps (slots 0-7: initial
sums, base pass, then per pass shuffle / deltas / visits) and, inside the visit rounds, window fetch / staging + prefix chains / pcc /
decision (slots 8-11) with the number of rounds (15).  One 4096x4096 bf16 tensor with the launch's shared orders, phase 0.
usage: MTQ_LIB=build/libmtq_prof.so python tools/scan_ticks.py [rows] [cols] [tensors]   (the stamps are block 0's: with more tensors, tensor 0 among them)"""
import ctypes, sys
sys.path.insert(0, '/root/repo')
import torch
from quantization_analysis_amd import hip_backend as hb
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hb.require_gpu(); L = hb.lib()
ALL = ["bf16", "bfp8", "bfp4", "bfp2"]
g = torch.Generator(device='cuda'); g.manual_seed(0)
x = (torch.randn((cnt, rows, cols), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
recs = hb.tile_stats_batched(x, 0xE); T = recs.shape[1]
sd = torch.full((cnt,), 123, dtype=torch.int64, device='cuda')
scratch = torch.empty((int(L.mtq_greedy_scan_scratch_bytes(cnt, T)),), dtype=torch.uint8, device='cuda')
maps = torch.empty((cnt, T), dtype=torch.int8, device='cuda'); status = torch.empty((cnt,), dtype=torch.int32, device='cuda')
orders = hb.scan_orders_device(123, T, 2)
for _ in range(2):
    hb.greedy_scan_device_ex(recs, 0xE | hb.MASK_BF16_IDENTITY, ALL, "pcc", 0.999, float(rows * cols), sd, maps, status, scratch, orders=orders)
torch.cuda.synchronize()
tk = (ctypes.c_uint64 * 16)(); hb.check(L.mtq_debug_scan_ticks(tk))
names = ["init sums", "base pass"] + [f"pass {p} {w}" for p in (1, 2) for w in ("order", "deltas", "visits")]
print(f"{T} tiles:", ", ".join(f"{n} {tk[i]/1e3:.0f}k" for i, n in enumerate(names)))
r = max(int(tk[15]), 1)
print(f"visit rounds {r}: per round window fetch {tk[8]/r:.0f}, staging + chains {tk[9]/r:.0f}, pcc {tk[10]/r:.0f}, decision + map {tk[11]/r:.0f} ticks")
print(f"own shuffles (the last pass's): draws + acceptance {tk[12]/1e3:.0f}k, swaps {tk[13]/1e3:.0f}k, hand-back {tk[14]/1e3:.0f}k ticks; pass 3: order {tk[8]/1e3:.0f}k deltas {tk[9]/1e3:.0f}k visits {tk[10]/1e3:.0f}k"
      " (slots 8-10 hold pass 3's stamps when the search has four formats: the per-round figures above are then meaningless)")
