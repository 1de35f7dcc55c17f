# Round-2 measurement run (one gpurun call): GPU tests, smoke, bench, the wq Llama run, K1 / K2 / K3 / K5 under rocprofv3,
# K1's VALU / wait counters.  Outputs under gpurun_out/r2/ (copy what is judged into profiles/).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=15 > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k1 -- python $R/tools/k1_bench.py 32 10 0xE > $O/k1_only.log 2>&1; echo "prof k1 rc=$?"; tail -1 $O/k1_only.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k23 -- python $R/tools/k23_bench.py > $O/k23.log 2>&1; echo "prof k23 rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -- python $R/tools/k1_bench.py 32 3 0xE > $O/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
