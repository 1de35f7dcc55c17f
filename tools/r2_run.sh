# Round-2 measurement run (one gpurun call): GPU tests, smoke, bench (device scan, and --scan host for comparison), K1 / K2 / K3 / K5
# under rocprofv3, K1's VALU / wait counters.  Outputs under gpurun_out/r2/ (copy what is judged into profiles/).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=15 > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
python bench.py --scan host --cpu-sample 0 > $O/bench_host_scan.json 2> $O/bench_host_scan.err; echo "bench host-scan rc=$?"; cut -c1-300 $O/bench_host_scan.json
