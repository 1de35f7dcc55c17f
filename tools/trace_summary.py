"""Per-kernel summary and a timeline of a rocprofv3 --kernel-trace CSV: python tools/trace_summary.py dir [t_from_ms t_to_ms]
(kernel count / mean / total; then every dispatch longer than 20 us whose start lies in the window, on the clock of the trace's first dispatch)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def tag(n):
    if 'tile_stats_bf16_rolled' in n: return 'K1-listed' if 'false, true' in n else 'K1'
    if 'tile_stats_listed' in n: return 'listed-direct'
    if 'tile_stats_direct' in n: return 'K1-direct'
    if 'scan_orders' in n: return 'orders'
    if 'greedy_scan' in n or 'greedy_atol' in n: return 'scan'
    if 'redo' in n: return 'redo'
    if 'columns_' in n: return 'colsum'
    if 'copy_rows' in n: return 'copy'
    if 'copyBuffer' in n: return 'memcpy'
    if 'fillBuffer' in n: return 'memset'
    if 'threshold_assign' in n: return 'K4'
    if 'knife' in n: return 'knife'
    return 'o:' + n[:36]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), tag(r['Kernel_Name']), r['Queue_Id'], r['Grid_Size_X'], r['Workgroup_Size_X']) for r in rows)
agg = collections.defaultdict(list)
for s, e, t, q, g, w in ev: agg[t].append((e - s) / 1e6)
for t, d in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:18]:
    print(f"{t:38s} n={len(d):5d} mean {sum(d)/len(d):8.4f} ms  total {sum(d):9.3f} ms")
if len(sys.argv) > 3:
    t0 = ev[0][0]; a, b = float(sys.argv[2]) * 1e6, float(sys.argv[3]) * 1e6
    for s, e, t, q, g, w in ev:
        if a <= s - t0 <= b and e - s > 20000:
            print(f"{(s-t0)/1e6:9.3f} {(e-t0)/1e6:9.3f} dur {(e-s)/1e6:7.3f} q{q:>3} {t:16s} grid {g:>9} wg {w:>4}")
