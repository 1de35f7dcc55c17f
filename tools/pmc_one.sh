#!/bin/bash
# one rocprofv3 --pmc pass (counters given as $2, kernel-trace only) over a tool command; prints the per-dispatch sums of the counter for
# kernels whose name contains $3.  usage: bash tools/pmc_one.sh outdir "COUNTER [COUNTER…]" kernel-name-part -- python3 tools/x.py args…
R=$GRAFT_REPO_ROOT; O=$R/$1; C=$2; K=$3; shift 4
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -- "$@" > $O/run.log 2>&1; echo "rc=$?"
python3 - "$O" "$K" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        per[(r["Kernel_Name"][:70], r["Counter_Name"])][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
for (n, c), v in per.items():
    print(n, c, [round(x) for _k, x in sorted(v.items())])
PY
