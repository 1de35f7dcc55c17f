"""Randomised GPU-vs-oracle parity sweep (not collected by pytest; run by hand on an MI355X box):
    python tests/fuzz_parity.py [cases] [seed]
Random shapes (ragged and aligned), storage types, value distributions (wide exponent spreads, zeros, sparse groups,
huge / tiny scales, ±Inf / NaN / denormals — two NaNs compare equal whatever their payload), format subsets: K1 records, K2 / K3 outputs and greedy / threshold maps against the CPU oracle, bit for bit."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
import torch  # noqa: E402

from oracle import mtq_oracle as orc  # noqa: E402
from quantization_analysis_amd import hip_backend as hb  # noqa: E402

ALL = ["bf16", "bfp8", "bfp4", "bfp2"]


def make(rng, shape, bf16):
    kind = rng.integers(0, 8)
    x = rng.standard_normal(shape)
    if kind == 1:
        x *= np.exp(rng.standard_normal(shape) * 2.0)
    elif kind == 2:
        x *= np.exp2(rng.integers(-30, 30, size=shape))
    elif kind == 3:
        x *= rng.random(shape) < 0.3
    elif kind == 4:
        x *= np.repeat(rng.random((shape[0], -(-shape[1] // 16))) < 0.5, 16, axis=1)[:, : shape[1]]
    elif kind == 5:
        x *= np.exp2(float(rng.integers(-60, 60)))
    x = (x * 0.02).astype(np.float32)
    if kind == 6:   # specials: ±Inf, NaN, ±0, float32 denormals, the largest finite values
        pool = np.array([np.inf, -np.inf, np.nan, 0.0, -0.0, 1e-40, -3e-39, 3.4e38, -3.3e38, 1.1754944e-38], dtype=np.float32)
        hit = rng.random(shape) < 0.02
        x = np.where(hit, pool[rng.integers(0, pool.size, size=shape)], x).astype(np.float32)
    elif kind == 7:  # everything tiny or everything huge (exponents outside the exact route's window)
        x = (x * np.float32(2.0) ** float(rng.choice([-120, -100, 100, 110]))).astype(np.float32)
    if bf16:
        x = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    return x


def kind_ok(x):
    return bool(np.isfinite(x).all() and np.abs(x).max() > 0 and np.isfinite(np.float64(x).__pow__(2).sum()))


def eq(a, b):
    both = np.isnan(a) & np.isnan(b)
    return np.array_equal(np.where(both, 0, a.view(np.uint64 if a.dtype == np.float64 else np.uint32)),
                          np.where(both, 0, b.view(np.uint64 if b.dtype == np.float64 else np.uint32)))


def zero_denominator(full, formats, thr, numel, seed) -> bool:
    """Does the host scan meet its zero-denominator branch (mixed_tile_greedy.py:186-189) on these records?  The slim records (no Σ|x−y|)
    refuse exactly such a search."""
    keep = [0, 1] + [2 + 5 * s_ + k for s_ in range(4) for k in range(3)]
    try:
        hb.greedy_run(np.ascontiguousarray(full[:, keep]), 0xF | hb.MASK_SLIM, formats, "pcc", thr, numel, seed)
        return False
    except hb.MtqError as exc:
        if "zero-variance" not in str(exc):
            raise
        return True


def search_variants(xd, a, thr, seed) -> bool:
    """The device search with the launch's shared orders, and (bf16 storage, whole 32x128 units) split in phases on partial records
    with the listed completion in between, against the oracle's map `a` (status 0 expected: the caller has excluded degenerate tensors)."""
    L = hb.lib()
    x3 = xd[None].contiguous()
    rows, cols = xd.shape
    ident = xd.dtype == torch.bfloat16
    k1 = 0xE if ident else 0xF
    dec = k1 | hb.MASK_BF16_IDENTITY if ident else k1
    recs = hb.tile_stats_batched(x3, k1)
    T = recs.shape[1]
    sd = torch.tensor([seed], dtype=torch.int64, device="cuda")
    scratch = torch.empty((int(L.mtq_greedy_scan_scratch_bytes(1, T)),), dtype=torch.uint8, device="cuda")
    maps = torch.empty((1, T), dtype=torch.int8, device="cuda")
    status = torch.empty((1,), dtype=torch.int32, device="cuda")
    orders = hb.scan_orders_device(seed, T, 2)
    hb.greedy_scan_device_ex(recs, dec, ALL, "pcc", thr, float(rows * cols), sd, maps, status, scratch, orders=orders)
    ok = int(status.cpu()[0]) == 0 and np.array_equal(maps.cpu().numpy().reshape(a.shape), a)
    if ident and rows % 32 == 0 and cols % 128 == 0:
        part = hb.tile_stats_partial(x3, 0xE, 0x2, 0x4)
        listed = torch.empty((T,), dtype=torch.int32, device="cuda")
        nl = torch.zeros((1,), dtype=torch.int32, device="cuda")
        carry = torch.empty((int(L.mtq_scan_carry_bytes(1)),), dtype=torch.uint8, device="cuda")
        maps.fill_(-1)
        hb.greedy_scan_device_ex(part, dec, ALL, "pcc", thr, float(rows * cols), sd, maps, status, scratch, orders=orders, phase=1, listed=listed, n_listed=nl, carry=carry)
        hb.tile_stats_listed(x3, 0xE, 0x8, 0x4, listed, nl, part, scratch=torch.empty((T + 1,), dtype=torch.int32, device="cuda"))
        hb.greedy_scan_device_ex(part, dec, ALL, "pcc", thr, float(rows * cols), sd, maps, status, scratch, phase=2, carry=carry)
        ok &= int(status.cpu()[0]) == 0 and np.array_equal(maps.cpu().numpy().reshape(a.shape), a)
    return bool(ok)


def run(cases: int, seed: int) -> int:
    """→ number of mismatching cases (each one is printed)."""
    rng = np.random.default_rng(seed)
    hb.require_gpu()
    t0 = time.time()
    bad = 0
    compared = handed_back = specials = specials_back = 0
    for c in range(cases):
        aligned = rng.random() < 0.4
        rows = int(rng.integers(1, 6)) * 32 if aligned else int(rng.integers(1, 200))
        cols = int(rng.integers(1, 5)) * 128 if aligned else int(rng.integers(1, 300))
        bf16 = bool(rng.random() < 0.5)
        x = make(rng, (rows, cols), bf16)
        mask = int(rng.integers(1, 16))
        fm = [f for i, f in enumerate(ALL) if mask >> i & 1]
        xd = torch.from_numpy(x).to(torch.bfloat16).cuda() if bf16 else torch.from_numpy(x).cuda()
        with np.errstate(all="ignore"):
            want = orc.tile_stats(x, fm)
            got = hb.tile_stats(xd, mask).cpu().numpy()
            ok = eq(got, want)
            fmt = ALL[int(rng.integers(0, 4))]
            ok_q = eq(hb.quantize(xd, fmt).cpu().numpy(), orc.quantize_weight_values(x, fmt))
            th, tw = -(-rows // 32), -(-cols // 32)
            amap = rng.integers(0, 4, size=(th, tw)).astype(np.int8)
            ok_a = eq(hb.apply_assignment(xd, amap).cpu().numpy(), orc.apply_assignment(x, amap))
            ok_g = ok_t = True
            if c % 4 == 0 and kind_ok(x):
                thr = float(rng.choice([0.99, 0.999, 0.9]))
                a, _counts, _st = orc.greedy(x, ALL, "pcc", thr, 7)
                full = hb.tile_stats(xd, 0xF).cpu().numpy()
                g, _c2, _o = hb.greedy_run(full, 0xF, ALL, "pcc", thr, float(x.size), 7)
                ok_g = np.array_equal(g.reshape(a.shape), a)
                # H1 on the device (csrc/mtq_scan.hip) on the same full records, and on a random format order / subset
                recs_d = torch.from_numpy(full).cuda()[None]
                sd = torch.tensor([7], dtype=torch.int64, device="cuda")
                cnt = torch.zeros((1, 4), dtype=torch.int32, device="cuda")
                dm, ds = hb.greedy_scan_device(recs_d, 0xF, ALL, "pcc", thr, float(x.size), sd, counts_out=cnt)
                # status 1 (zero denominator met) hands the tensor to the host scan — exactly when the host scan itself meets that branch
                # (slim records refuse such a tensor): a device scan that handed everything back would otherwise pass unnoticed
                degenerate = zero_denominator(full, ALL, thr, float(x.size), 7)
                compared += 1
                handed_back += int(ds.cpu()[0]) != 0
                ok_g &= (int(ds.cpu()[0]) != 0) == degenerate
                if int(ds.cpu()[0]) == 0:
                    ok_g &= np.array_equal(dm.cpu().numpy().reshape(a.shape), a)
                    ok_g &= cnt.cpu().numpy()[0].tolist() == [int(np.sum(a == k)) for k in range(4)]
                # round 3: the launch's shared visiting orders (helper wave), and — bf16 storage in whole 32x128 units — the search in
                # phases on partial records with the listed completion in between (what the streamed driver runs)
                if not degenerate:
                    ok_g &= search_variants(xd, a, thr, 7)
                # the mae search on the device against the host scan on the same records
                thr_m = float(rng.choice([1e-5, 1e-4, 1e-3])) * max(float(np.abs(x).mean()) / 0.016, 1e-6)
                gm, _cm, _om = hb.greedy_run(full, 0xF, ALL, "mae", thr_m, float(x.size), 7)
                dm, ds = hb.greedy_scan_device(recs_d, 0xF, ALL, "mae", thr_m, float(x.size), sd)
                ok_g &= int(ds.cpu()[0]) == 0 and np.array_equal(dm.cpu().numpy()[0], gm)
                thr_a = float(rng.choice([1e-3, 1e-2, 0.1])) * max(float(np.abs(x).max()), 1e-30)
                ga, _ca, _oa = hb.greedy_run(full, 0xF, ALL, "atol", thr_a, float(x.size), 7)
                dm, ds = hb.greedy_scan_device(recs_d, 0xF, ALL, "atol", thr_a, float(x.size), sd)
                ok_g &= int(ds.cpu()[0]) == 0 and np.array_equal(dm.cpu().numpy()[0], ga)
                order = [ALL[k] for k in rng.permutation(4)[: int(rng.integers(1, 5))]]
                s2 = int(rng.integers(1, 2**31))
                gh, _ch, _oh = hb.greedy_run(full, 0xF, order, "pcc", thr, float(x.size), s2)
                sd[0] = s2
                dm, ds = hb.greedy_scan_device(recs_d, 0xF, order, "pcc", thr, float(x.size), sd)
                degenerate = zero_denominator(full, order, thr, float(x.size), s2)
                compared += 1
                handed_back += int(ds.cpu()[0]) != 0
                ok_g &= (int(ds.cpu()[0]) != 0) == degenerate
                if int(ds.cpu()[0]) == 0:
                    ok_g &= np.array_equal(dm.cpu().numpy()[0], gh)
                # the record layouts of the streamed driver: slim (3 doubles per format), slim + identity bf16
                keep = [0, 1] + [2 + 5 * s_ + k for s_ in range(4) for k in range(3)]
                try:
                    g2, _c, _o = hb.greedy_run(np.ascontiguousarray(full[:, keep]), 0xF | hb.MASK_SLIM, ALL, "pcc", thr, float(x.size), 7)
                    ok_g &= np.array_equal(g2.reshape(a.shape), a)
                    if bf16:
                        ident = np.ascontiguousarray(full[:, [0, 1] + [2 + 5 * s_ + k for s_ in (1, 2, 3) for k in range(3)]])
                        g4, _c, _o = hb.greedy_run(ident, 0xE | hb.MASK_BF16_IDENTITY | hb.MASK_SLIM, ALL, "pcc", thr, float(x.size), 7)
                        ok_g &= np.array_equal(g4.reshape(a.shape), a)
                except hb.MtqError as exc:   # slim records refuse a zero-variance tensor (the driver then takes the full records)
                    if "zero-variance" not in str(exc):
                        raise
            elif c % 4 == 0:   # ±Inf / NaN / all-zero inputs: no oracle search to compare with, but the device scan must do what the host scan does
                thr = float(rng.choice([0.99, 0.999, 0.9]))
                full = hb.tile_stats(xd, 0xF).cpu().numpy()
                gh, _ch, _oh = hb.greedy_run(full, 0xF, ALL, "pcc", thr, float(x.size), 7)
                sd = torch.tensor([7], dtype=torch.int64, device="cuda")
                dm, ds = hb.greedy_scan_device(torch.from_numpy(full).cuda()[None], 0xF, ALL, "pcc", thr, float(x.size), sd)
                specials += 1
                specials_back += int(ds.cpu()[0]) != 0
                if int(ds.cpu()[0]) == 0:
                    ok_g = np.array_equal(dm.cpu().numpy()[0], gh)
                for met, th2 in (("mae", 1e-4), ("atol", 1e-2)):   # status 0 must mean the host scan's map, whatever the input holds
                    gh2, _c2h, _o2h = hb.greedy_run(full, 0xF, ALL, met, th2, float(x.size), 7)
                    dm2, ds2 = hb.greedy_scan_device(torch.from_numpy(full).cuda()[None], 0xF, ALL, met, th2, float(x.size), sd)
                    if int(ds2.cpu()[0]) == 0:
                        ok_g &= np.array_equal(dm2.cpu().numpy()[0], gh2)
        if not (ok and ok_q and ok_a and ok_g and ok_t):
            bad += 1
            print(f"MISMATCH case {c}: shape {(rows, cols)} bf16 {bf16} mask {mask:#x} fmt {fmt}: stats {ok} quantize {ok_q} apply {ok_a} greedy {ok_g}", flush=True)
    print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.1f} s; device pcc searches compared {compared}, handed back {handed_back} "
          f"(each checked against the host scan's zero-denominator branch); special-value inputs {specials}, handed back {specials_back}")
    if compared and handed_back > compared // 2:
        print("more than half of the device searches were handed back: the comparison is not comparing")
        bad += 1
    return bad


if __name__ == "__main__":
    raise SystemExit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
