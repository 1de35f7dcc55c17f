"""Round-3 device search: shared visiting orders, the helper wave, and the split (lazy) search, against the host scan on the same records;
timings by HIP events.  python tools/scan_v3_check.py [n] [reps]"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from quantization_analysis_amd import hip_backend as hb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
hb.require_gpu()
L = hb.lib()
ALL = ["bf16", "bfp8", "bfp4", "bfp2"]
DEC = 0xE | hb.MASK_BF16_IDENTITY
bad = 0


def run_case(x, formats, thr, seed, metric="pcc", label=""):
    """x: (count, rows, cols) bf16 device tensor.  Host scan on full records is the reference for all device variants."""
    global bad
    count, rows, cols = x.shape
    ident = "bf16" in formats and any(f != "bf16" for f in formats)
    k1 = hb.fmt_mask(formats) & 0xE if ident else hb.fmt_mask(formats)
    dec = k1 | hb.MASK_BF16_IDENTITY if ident else k1
    recs = hb.tile_stats_batched(x, k1)
    T = recs.shape[1]
    numel = float(rows * cols)
    want, wcounts, wouts = hb.greedy_run_batch(recs.cpu().numpy(), dec, formats, metric, thr, numel, [seed] * count, 8)
    sd = torch.full((count,), seed, dtype=torch.int64, device='cuda')
    scratch = torch.empty((int(L.mtq_greedy_scan_scratch_bytes(count, T)),), dtype=torch.uint8, device='cuda')
    maps = torch.empty((count, T), dtype=torch.int8, device='cuda'); status = torch.empty((count,), dtype=torch.int32, device='cuda')
    cnt = torch.zeros((count, 4), dtype=torch.int32, device='cuda')

    def check(tag):
        global bad
        torch.cuda.synchronize()
        ok = status.cpu().tolist() == [0] * count and np.array_equal(maps.cpu().numpy(), want) and np.array_equal(cnt.cpu().numpy(), wcounts)
        if not ok:
            bad += 1
            print(f"  MISMATCH {label} {tag}: status {status.cpu().tolist()[:8]} differing tiles {int((maps.cpu().numpy() != want).sum())}")
        return ok

    # plain
    hb.greedy_scan_device_ex(recs, dec, formats, metric, thr, numel, sd, maps, status, scratch, counts_out=cnt)
    ok0 = check("plain")
    # shared orders
    orders = hb.scan_orders_device(seed, T, 2)
    maps.fill_(-1); cnt.zero_()
    hb.greedy_scan_device_ex(recs, dec, formats, metric, thr, numel, sd, maps, status, scratch, counts_out=cnt, orders=orders)
    ok1 = check("shared orders")
    ok2 = True
    if len(formats) >= 3 and metric == "pcc" and ident:
        # the split search on partial records: every format but the last in K1 (the one before it without Σ|x−y|, max), the rest listed
        lay = k1
        last, prev = hb.fmt_mask([formats[-1]]), hb.fmt_mask([formats[-2]])
        full = lay & ~last & ~prev
        part = hb.tile_stats_partial(x, lay, full, prev)
        listed = torch.empty((count * T,), dtype=torch.int32, device='cuda'); nl = torch.zeros((1,), dtype=torch.int32, device='cuda')
        carry = torch.empty((int(L.mtq_scan_carry_bytes(count)),), dtype=torch.uint8, device='cuda')
        maps.fill_(-1); cnt.zero_()
        hb.greedy_scan_device_ex(part, dec, formats, metric, thr, numel, sd, maps, status, scratch, counts_out=cnt, orders=orders, phase=1, listed=listed, n_listed=nl, carry=carry)
        hb.tile_stats_listed(x, lay, last, prev, listed, nl, part, scratch=torch.empty((count * T + 1,), dtype=torch.int32, device='cuda'))
        hb.greedy_scan_device_ex(part, dec, formats, metric, thr, numel, sd, maps, status, scratch, counts_out=cnt, phase=2, carry=carry)
        ok2 = check("split / lazy")
        # the columns of the final map from the lazily filled records == from the full records
        ns = int(L.mtq_columns_scratch_doubles())
        s_full = torch.empty((count, ns), dtype=torch.float64, device='cuda'); s_lazy = torch.empty_like(s_full)
        hb.check(L.mtq_column_sums_device_batched(recs.data_ptr(), count, T, dec, maps.data_ptr(), s_full.data_ptr(), hb._stream_ptr()))
        hb.check(L.mtq_column_sums_device_batched(part.data_ptr(), count, T, dec, maps.data_ptr(), s_lazy.data_ptr(), hb._stream_ptr()))
        torch.cuda.synchronize()
        if not torch.equal(s_full[:, :7].view(torch.int64), s_lazy[:, :7].view(torch.int64)):
            bad += 1
            print(f"  MISMATCH {label} lazy column sums")
            ok2 = False
        print(f"  {label}: listed {int(nl.item())} of {count * T} tiles ({100.0 * int(nl.item()) / (count * T):.1f} %)")
    print(f"{label}: plain {'ok' if ok0 else 'BAD'}, shared {'ok' if ok1 else 'BAD'}, split {'ok' if ok2 else 'BAD'}; counts[0] {wcounts[0].tolist()}")


g = torch.Generator(device='cuda'); g.manual_seed(0)
small = (torch.randn((6, 1024, 2048), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
for thr in (0.9, 0.99, 0.999, 0.9999, 0.99999, 0.9999999):
    run_case(small, ALL, thr, 123, label=f"6x1024x2048 pcc {thr}")
run_case(small, ["bf16", "bfp8", "bfp4"], 0.999, 7, label="3 formats")
run_case(small, ["bf16", "bfp4", "bfp2"], 0.99, 9, label="bf16,bfp4,bfp2")
run_case(small, ["bf16", "bfp2"], 0.9, 11, label="2 formats")
run_case(small, ["bfp8", "bfp4", "bfp2"], 0.995, 5, label="base bfp8")
run_case(small, ALL, 2e-4, 31, metric="mae", label="mae 2e-4")
run_case(small, ALL, 1e-3, 31, metric="mae", label="mae 1e-3")
tiny = (torch.randn((4, 32, 128), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
run_case(tiny, ALL, 0.999, 51, label="4 tiles")
run_case(tiny[:, :, :32].contiguous(), ALL, 0.999, 61, label="1 tile")
heavy = (torch.randn((3, 512, 512), generator=g, device='cuda') * 0.02 * torch.exp(1.5 * torch.randn((3, 512, 512), generator=g, device='cuda'))).to(torch.bfloat16)
for thr in (0.9, 0.99, 0.999):
    run_case(heavy, ALL, thr, 41, label=f"heavy {thr}")
big = (torch.randn((2, 14336, 4096), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
run_case(big, ALL, 0.999, 123, label="2x14336x4096 (global order)")
del big

# ---- timing on the headline shape
x = (torch.randn((n, 4096, 4096), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
run_case(x, ALL, 0.999, 123, label=f"{n}x4096x4096")
def timing(x):
    n = x.shape[0]
    rows_, cols_ = x.shape[1], x.shape[2]
    recs = hb.tile_stats_batched(x, 0xE)
    T = recs.shape[1]
    numel = float(rows_ * cols_)
    sd = torch.full((n,), 123, dtype=torch.int64, device='cuda')
    scratch = torch.empty((int(L.mtq_greedy_scan_scratch_bytes(n, T)),), dtype=torch.uint8, device='cuda')
    maps = torch.empty((n, T), dtype=torch.int8, device='cuda'); status = torch.empty((n,), dtype=torch.int32, device='cuda')
    orders = hb.scan_orders_device(123, T, 2)
    listed = torch.empty((n * T,), dtype=torch.int32, device='cuda'); nl = torch.zeros((1,), dtype=torch.int32, device='cuda')
    carry = torch.empty((int(L.mtq_scan_carry_bytes(n)),), dtype=torch.uint8, device='cuda')
    part = hb.tile_stats_partial(x, 0xE, 0x2, 0x4)
    lscr = torch.empty((n * T + 1,), dtype=torch.int32, device='cuda')


    def timeit(fn):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]


    print(f"timing, {n} tensors x {T} tiles (ms per launch, median of {reps}):")
    print(f"  orders kernel (once per launch, beside K1)      {timeit(lambda: hb.scan_orders_device(123, T, 2, out=orders)):.3f}")
    print(f"  search, own shuffles (round-2 form)             {timeit(lambda: hb.greedy_scan_device_ex(recs, DEC, ALL, 'pcc', 0.999, numel, sd, maps, status, scratch)):.3f}")
    print(f"  search, shared orders + helper wave             {timeit(lambda: hb.greedy_scan_device_ex(recs, DEC, ALL, 'pcc', 0.999, numel, sd, maps, status, scratch, orders=orders)):.3f}")


    def phase1():
        nl.zero_()
        hb.greedy_scan_device_ex(part, DEC, ALL, 'pcc', 0.999, numel, sd, maps, status, scratch, orders=orders, phase=1, listed=listed, n_listed=nl, carry=carry)


    print(f"  split: phase 1 (shared orders)                  {timeit(phase1):.3f}")
    phase1()
    print(f"  split: listed K1 ({int(nl.item())} tiles)                {timeit(lambda: hb.tile_stats_listed(x, 0xE, 0x8, 0x4, listed, nl, part, scratch=lscr)):.3f}")


    def chain():
        phase1()
        hb.tile_stats_listed(x, 0xE, 0x8, 0x4, listed, nl, part, scratch=lscr)
        hb.greedy_scan_device_ex(part, DEC, ALL, 'pcc', 0.999, numel, sd, maps, status, scratch, phase=2, carry=carry)


    print(f"  split: phase 1 + listed K1 + phase 2             {timeit(chain):.3f}")
    print(f"  split: listed K1, one wave per tile (no scratch) {timeit(lambda: hb.tile_stats_listed(x, 0xE, 0x8, 0x4, listed, nl, part)):.3f}")
    print(f"  K1 partial (bfp8 full, bfp4 sums)               {timeit(lambda: hb.tile_stats_partial(x, 0xE, 0x2, 0x4, out=part)):.3f}")
    print(f"  K1 full 0xE                                      {timeit(lambda: hb.tile_stats_batched(x, 0xE, out=recs)):.3f}")


timing(x)
del x
# the large shape of BASELINE configs[3] (Llama-3-8B gate / up projections: 57 344 tiles per tensor), one tensor and a shape group's worth
for cnt in (1, 32):
    timing((torch.randn((cnt, 14336, 4096), generator=g, device='cuda') * 0.02).to(torch.bfloat16))
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
