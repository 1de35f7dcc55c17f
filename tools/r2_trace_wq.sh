set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2tw; rm -rf $O; mkdir -p $O/wq
cd /tmp && export TMPDIR=/tmp
cd $O/wq && rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/wq synthetic:llama3-8b model.layers --backend hip --no-plots --compression-config $R/compression_configs/greedy_seed123.json > $O/wq.log 2>&1; echo "rc=$?"; grep streamed $O/wq.log
cd $O/trace/*/ && python - <<'PY'
import csv, glob
f = glob.glob('*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    n = r['Kernel_Name']
    tag = 'K1' if 'tile_stats_bf16' in n else 'scan' if 'greedy_scan' in n else 'redo' if 'redo_flagged' in n else 'colsum' if 'column_sums' in n else None
    if tag: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), tag, r['Grid_Size']))
ev.sort()
t0 = ev[0][0]
for s, e, t, g in ev:
    if t in ('K1', 'scan'): print(f"{t:5s} start {(s-t0)/1e6:9.3f} ms  dur {(e-s)/1e6:8.3f} ms  grid {g}")
PY
