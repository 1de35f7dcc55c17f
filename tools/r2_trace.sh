set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2t; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --cpu-sample 0 --scan device --chunk 128 --steps 30 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "rc=$?"
cd $O/trace/*/ && python - <<'PY'
import csv, glob
f = glob.glob('*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    n = r['Kernel_Name']
    tag = 'K1' if 'tile_stats_bf16' in n else 'scan' if 'greedy_scan' in n else 'redo' if 'redo_flagged' in n else 'colsum' if 'column' in n or 'colsum' in n.lower() else None
    if tag: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), tag))
ev.sort()
t0 = ev[0][0]
k1 = [(s, e) for s, e, t in ev if t == 'K1']
sc = [(s, e) for s, e, t in ev if t == 'scan']
print('K1 launches', len(k1), 'scan launches', len(sc))
for i in range(8, 16):
    s, e = k1[i]
    print(f"K1[{i}] start {(s-t0)/1e6:8.3f} ms dur {(e-s)/1e6:.3f} gap-from-prev-end {(s-k1[i-1][1])/1e6:.3f}")
for i in range(8, 16):
    s, e = sc[i]
    print(f"scan[{i}] start {(s-t0)/1e6:8.3f} ms dur {(e-s)/1e6:.3f}")
PY
