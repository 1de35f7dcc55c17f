"""Dispatches of the last run_batches call in a rocprofv3 --kernel-trace of tools/thr_trace.py: python tools/thr_timeline.py dir [n_last]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r['Queue_Id'], r['Grid_Size_X']) for r in rows)[-n:]
t0 = ev[0][0]
prev = t0
for s, e, k, q, g in ev:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev) / 1e3:7.1f} gap  dur {(e - s) / 1e3:8.1f} us  q{q:>2} grid {g:>8}  {k}")
    prev = e
