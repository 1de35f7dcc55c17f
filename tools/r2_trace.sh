set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2t; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --cpu-sample 0 --steps 30 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "rc=$?"
cd $O/trace/*/ && python - <<'PY'
import csv, glob
f = glob.glob('*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    n = r['Kernel_Name']
    tag = 'K1' if 'tile_stats_bf16' in n else 'scan' if 'greedy_scan' in n else 'redo' if 'redo_flagged' in n else 'colsum' if 'column' in n or 'colsum' in n.lower() else 'other'
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), tag, n[:60]))
ev.sort()
k1 = [(s, e) for s, e, t, n in ev if t == 'K1' and e - s > 1e6]
t0 = k1[10][0]
print('big K1 launches', len(k1))
for i in range(12, 18):
    s, e = k1[i]
    print(f"K1[{i}] start {(s-t0)/1e6:8.3f} ms dur {(e-s)/1e6:.3f} gap-from-prev-end {(s-k1[i-1][1])/1e6:.3f}")
# everything between the end of K1[13] and the start of K1[14]
a, b = k1[13][1], k1[14][0]
print("between K1[13] end and K1[14] start:")
for s, e, t, n in ev:
    if s >= a - 20000 and s <= b + 20000 and not (t == 'K1' and e - s > 1e6):
        print(f"   {t:7s} start {(s-a)/1e3:8.1f} us dur {(e-s)/1e3:8.1f} us  {n}")
PY
