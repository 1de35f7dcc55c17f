#!/bin/bash
# Kernel timeline of the bench (rocprofv3 --kernel-trace): per kernel count / mean duration, and one steady-state step laid out on the
# clock of its K1 launch.  usage: bash tools/r3_trace.sh [outdir] [steps] ; environment (MTQ_LAZY=0 …) is passed on to bench.py
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r3t}; steps=${2:-30}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --cpu-sample 0 --steps $steps --warmup 3 > $O/bench.json 2> $O/bench.err; echo "rc=$?"
cd $O/trace/*/ && python3 - <<'PY'
import csv, glob, collections
f = glob.glob('*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
def tag(n):
    if 'tile_stats_bf16_rolled' in n: return 'K1-listed' if 'Lb0ELb1E' in n or 'false, true' in n else 'K1'
    if 'tile_stats_listed' in n: return 'listed-direct'
    if 'scan_orders' in n: return 'orders'
    if 'greedy_scan' in n: return 'scan'
    if 'redo_flagged' in n: return 'redo'
    if 'column' in n.lower() or 'colsum' in n.lower(): return 'colsum'
    if 'copy_rows' in n: return 'copy'
    return 'other:' + n[:40]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), tag(r['Kernel_Name']), r['Kernel_Name'][:70]) for r in rows)
agg = collections.defaultdict(list)
for s, e, t, n in ev: agg[t].append((e - s) / 1e6)
for t, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{t:28s} n={len(d):5d} mean {sum(d)/len(d):8.4f} ms  total {sum(d):9.3f} ms")
k1 = [(s, e) for s, e, t, n in ev if t == 'K1' and e - s > 1e6]
print('big K1 launches', len(k1))
if len(k1) > 16:
    t0 = k1[12][0]
    for i in range(12, 16):
        s, e = k1[i]; print(f"K1[{i}] start {(s-t0)/1e6:8.3f} ms dur {(e-s)/1e6:.3f} gap-from-prev-end {(s-k1[i-1][1])/1e6:.3f}")
    # what runs beside one K1 launch, as a share of that launch's duration
    for i in (13, 14):
        ks, ke = k1[i]
        over = collections.defaultdict(float)
        for s, e, t, n in ev:
            if (s, e) == (ks, ke): continue
            o = min(e, ke) - max(s, ks)
            if o > 0: over[t] += o / 1e6
        print(f"beside K1[{i}] ({(ke-ks)/1e6:.3f} ms): " + ", ".join(f"{t} {v:.3f}" for t, v in sorted(over.items(), key=lambda kv: -kv[1])))
    a, b = k1[12][0], k1[15][1]
    print("everything else between K1[12] start and K1[15] end (ms from K1[12] start):")
    for s, e, t, n in ev:
        if a <= s <= b and not (t == 'K1' and e - s > 1e6) and (e - s) > 20000:
            print(f"   {t:14s} start {(s-a)/1e6:8.3f} end {(e-a)/1e6:8.3f} dur {(e-s)/1e6:7.3f}")
PY
