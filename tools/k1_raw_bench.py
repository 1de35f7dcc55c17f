"""K1 through a given build of the library by bare ctypes (no hip_backend: works with older builds of the ABI): the whole-record launch
on N 4096x4096 bf16 tensors, a few times.  For PMC comparisons between builds.  usage: k1_raw_bench.py lib.so [n] [reps] [mask]"""
import ctypes, sys
import torch
lib = ctypes.CDLL(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mask = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0xE
vp, i64, u32, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32, ctypes.c_int
lib.mtq_tile_stats_batched.argtypes = [vp, ci, i64, i64, i64, i64, i64, u32, vp, vp]
g = torch.Generator(device='cuda'); g.manual_seed(0)
x = (torch.randn((n, 4096, 4096), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
rec = 2 + 5 * bin(mask).count("1")
out = torch.empty((n, 16384, rec), dtype=torch.float64, device='cuda')
for _ in range(reps + 1):
    rc = lib.mtq_tile_stats_batched(x.data_ptr(), 0, n, 4096 * 4096, 4096, 4096, 4096, mask, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
print("ok", float(out[0, 0, 0]))
