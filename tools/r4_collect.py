"""Copies the judged summaries of a `tools/r4_final.sh <tag>` run (gpurun_out/<tag>/) into profiles/ (tracked) as r4_<suffix>_*, and
recomputes profiles/k1_traffic.json and profiles/k1_valu.json from the PMC passes of that run (the launch bench.py issues per step on the
lazy route: mtq_tile_stats_partial, layout 0xE, bfp8 whole, bfp4 sums).  usage: python tools/r4_collect.py [tag] [suffix]"""
import csv, glob, json, os, shutil, sys, collections
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r4f"
suf = sys.argv[2] if len(sys.argv) > 2 else "a"
O, P = ROOT / "gpurun_out" / tag, ROOT / "profiles"


def newest(pattern, must=True):
    files = sorted(glob.glob(str(O / pattern)), key=os.path.getmtime)
    if not files:
        if must:
            sys.exit(f"nothing matches {pattern}")
        return None
    return files[-1]


def per_dispatch(path, kernel_part):
    """{counter: [sum over dimension instances per dispatch, in dispatch order]} for kernels whose name contains kernel_part."""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if kernel_part in r["Kernel_Name"]:
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    return {c: [v for _k, v in sorted(d.items())] for c, d in per.items()}


def steady(vals):
    big = [v for v in vals if v > 0.5 * max(vals)]
    return sum(big) / len(big)


def slim_counter_csv(src, dst):
    agg = {}
    for r in csv.DictReader(open(src)):
        if "mtq::" not in r["Kernel_Name"]:
            continue
        k = (r["Dispatch_Id"], r["Kernel_Name"][:80], r["Grid_Size"], r["Counter_Name"])
        agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
        for (d, n, g, c), v in agg.items():
            w.writerow([d, n, g, c, f"{v:.6f}"])


def copy(src, name):
    if src and os.path.exists(src):
        shutil.copy(src, P / f"r4_{suf}_{name}")
        return True
    return False


for src, name in (("bench.json", "bench.json"), ("bench_steps20.json", "bench_steps20.json"), ("bench_whole_records.json", "bench_whole_records.json"),
                  ("bench_r2_form.json", "bench_r2_form.json"), ("bench_llama3_8b.json", "bench_llama3_8b.json"), ("bench_under_rocprof.json", "bench_under_rocprofv3.json"),
                  ("scan_v3_check.txt", "scan_v3_check.txt"), ("threshold_pipeline.txt", "threshold_pipeline.txt"), ("exit_check.txt", "exit_check.txt"),
                  ("fuzz_2000.txt", "fuzz_2000.txt"), ("k1_only.log", "k1_only.txt"), ("k1_f32.log", "k1_f32.txt"), ("k1_listed.txt", "k1_listed.txt"),
                  ("k1_f32dom_vs_intdom.txt", "k1_f32dom_vs_intdom.txt"), ("ubench_f32.txt", "ubench_f32.txt"), ("gpu_tests.log", "gpu_tests.txt")):
    copy(str(O / src), name)
for pat, name in (("prof_bench/*/*_kernel_stats.csv", "bench_kernel_stats.csv"), ("prof_k1/*/*_kernel_stats.csv", "k1_only_kernel_stats.csv"),
                  ("prof_f32/*/*_kernel_stats.csv", "k1_f32_kernel_stats.csv")):
    copy(newest(pat, must=False), name)
for pat, name in (("pmc_fetch/*/*_counter_collection.csv", "pmc_fetch_size.csv"), ("pmc_write/*/*_counter_collection.csv", "pmc_write_size.csv"),
                  ("pmc_sq/*/*_counter_collection.csv", "pmc_sq_k1.csv"), ("pmc_sq_f32/*/*_counter_collection.csv", "pmc_sq_k1_f32.csv"),
                  ("pmc_fetch_f32/*/*_counter_collection.csv", "pmc_fetch_size_f32.csv"), ("pmc_write_f32/*/*_counter_collection.csv", "pmc_write_size_f32.csv")):
    f = newest(pat, must=False)
    if f:
        slim_counter_csv(f, P / f"r4_{suf}_{name}")

fetch, write, sq = newest("pmc_fetch/*/*_counter_collection.csv", False), newest("pmc_write/*/*_counter_collection.csv", False), newest("pmc_sq/*/*_counter_collection.csv", False)
lazy = "tile_stats_bf16_rolled<3u, 1u"
if fetch and write:
    tiles = 128 * 128 * 128
    fk, wk = steady(per_dispatch(fetch, lazy)["FETCH_SIZE"]), steady(per_dispatch(write, lazy)["WRITE_SIZE"])
    t = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on tools/k1_partial_bench.py 128 3 0xE 0x2 0x4 — the K1 launch bench.py issues per step "
                   f"on the lazy route (128 tensors, layout bfp8|bfp4|bfp2, bfp8 whole + bfp4 sums evaluated, the rest of every record NaN) — MI355X, round 4 (profiles/r4_{suf}_pmc_*.csv)",
         "tiles_per_launch": tiles, "layout_mask": "0xE", "evaluated": "bfp8 (5 statistics), bfp4 (3 sums)",
         "tile_stats_bf16_rolled<3,1>": {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "read_bytes_corrected": 2.0 * 1024.0 * fk,
                                         "note": "gfx950 FETCH_SIZE counts 1/2 of wide (16 B/lane) streaming reads incl. LDS-DMA: doubled (MI355X_MICROARCH.md HBM section); WRITE_SIZE exact"},
         "algorithmic_bytes_per_tile": {"read": 2048, "write": 136},
         "hbm_bytes_per_launch": 2.0 * 1024.0 * fk + 1024.0 * wk}
    t["hbm_bytes_per_tile"] = t["hbm_bytes_per_launch"] / tiles
    full = per_dispatch(fetch, "tile_stats_bf16_rolled<7u, 7u")
    if full.get("FETCH_SIZE"):
        fw, ww = steady(full["FETCH_SIZE"]), steady(per_dispatch(write, "tile_stats_bf16_rolled<7u, 7u")["WRITE_SIZE"])
        t["whole_records"] = {"kernel": "tile_stats_bf16_rolled<7,7> (MTQ_LAZY=0: every statistic of every format; same tool, same passes)", "FETCH_SIZE_KB": fw, "WRITE_SIZE_KB": ww,
                              "hbm_bytes_per_tile": (2.0 * 1024.0 * fw + 1024.0 * ww) / tiles}
    (P / "k1_traffic.json").write_text(json.dumps(t, indent=1) + "\n")
    print(f"traffic: {t['hbm_bytes_per_tile']:.1f} B/tile (2048 + 136 algorithmic)")
if sq:
    tiles = 32 * 128 * 128
    out = {}
    for name, part in (("lazy", lazy), ("whole", "tile_stats_bf16_rolled<7u, 7u")):
        c = per_dispatch(sq, part)
        if not c.get("SQ_INSTS_VALU"):
            continue
        wc = steady(c["SQ_WAVE_CYCLES"])
        out[name] = {"valu_insts_per_tile": steady(c["SQ_INSTS_VALU"]) / tiles, "issuing": steady(c["SQ_ACTIVE_INST_ANY"]) / wc,
                     "waiting_for_issue": steady(c["SQ_WAIT_INST_ANY"]) / wc, "waiting_for_memory": steady(c["SQ_WAIT_ANY"]) / wc}
    if "lazy" in out:
        # issue cost of the lazy kernel's static mix (tools/isa_mix.py on the listing of the round-4 float-domain tile_stats_bf16_rolled<3,1>: 31 % plain
        # 32-bit / fp32 add-mul forms at ~2.7 cycles, 69 % packed-fp32 / med3 / max3 / fp64 / convert forms at ~4.4 -> 3.87; the whole-record kernel <7,7>: 3.88)
        v = {"valu_insts_per_tile": round(out["lazy"]["valu_insts_per_tile"], 1), "avg_issue_cycles_per_inst": 3.87, "simds": 1024, "clock_hz": 2400000000.0,
             "wave_cycle_shares": {k: round(x, 3) for k, x in out["lazy"].items() if k != "valu_insts_per_tile"},
             "whole_records": dict({k: round(x, 3) for k, x in out.get("whole", {}).items()}, avg_issue_cycles_per_inst=3.88),
             "source": f"rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES on tools/k1_partial_bench.py 32 3 0xE 0x2 0x4 "
                       f"(profiles/r4_{suf}_pmc_sq_k1.csv): the lazy route's K1, tile_stats_bf16_rolled<3,1>; issue cost = the kernel's static mix (tools/isa_mix.py; issue rates profiles/r1_f_valu_issue_rates.txt)"}
        (P / "k1_valu.json").write_text(json.dumps(v, indent=1) + "\n")
        print("valu:", json.dumps(out))
sqf = newest("pmc_sq_f32/*/*_counter_collection.csv", False)
if sqf:
    c = per_dispatch(sqf, "tile_stats_direct<float, 15u>")
    ff, wf = newest("pmc_fetch_f32/*/*_counter_collection.csv", False), newest("pmc_write_f32/*/*_counter_collection.csv", False)
    tiles = 8 * 128 * 128
    wc = steady(c["SQ_WAVE_CYCLES"])
    d = {"kernel": "tile_stats_direct<float, 15> (K1, float32 storage, all four formats), tools/k1_f32_bench.py 8 3 normal", "tiles_per_launch": tiles,
         "valu_insts_per_tile": steady(c["SQ_INSTS_VALU"]) / tiles, "issuing": steady(c["SQ_ACTIVE_INST_ANY"]) / wc, "waiting_for_issue": steady(c["SQ_WAIT_INST_ANY"]) / wc,
         "waiting_for_memory": steady(c["SQ_WAIT_ANY"]) / wc}
    if ff and wf:
        d["FETCH_SIZE_KB"] = steady(per_dispatch(ff, "tile_stats_direct<float, 15u>")["FETCH_SIZE"])
        d["WRITE_SIZE_KB"] = steady(per_dispatch(wf, "tile_stats_direct<float, 15u>")["WRITE_SIZE"])
        d["hbm_bytes_per_tile"] = (2.0 * 1024.0 * d["FETCH_SIZE_KB"] + 1024.0 * d["WRITE_SIZE_KB"]) / tiles
        d["algorithmic_bytes_per_tile"] = {"read": 4096, "write": 176}
    (P / "k1_f32_counters.json").write_text(json.dumps(d, indent=1) + "\n")
    print("f32:", json.dumps(d))
b = O / "bench.json"
if b.exists():
    b = json.loads(b.read_text())
    print(f"bench {b['value'] / 1e6:.1f} M tiles/s, {b['ms_per_step']:.3f} ms/step, K1 {b['roofline']['launch_ms']:.3f} ms ({b['roofline']['frac']:.3f}), "
          f"alone {b['roofline']['kernel_alone']['launch_ms']:.3f} ms ({b['roofline']['kernel_alone']['frac']:.3f})")
