"""What runs after the last K1 launch of a traced bench region (rocprofv3 --kernel-trace csv): the drain every timed region pays once.
usage: python tools/r3_tail.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
k1 = [r for r in rows if "rolled<3u, 1u" in r["Kernel_Name"] or "rolled<7u, 7u" in r["Kernel_Name"]]
t0 = int(k1[-1]["End_Timestamp"])
print(f"last K1: {(t0 - int(k1[-1]['Start_Timestamp'])) / 1e3:.0f} us; kernels from 2.5 ms before its end (start, end in us relative to it):")
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if s > -2500000 and "mtq" in r["Kernel_Name"]:
        print(f"  {r['Kernel_Name'].replace('mtq::(anonymous namespace)::', '').replace('void mtq::', '')[:50]:50s} {s / 1e3:8.0f} {e / 1e3:8.0f}")
