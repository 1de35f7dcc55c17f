# Round-2 final measurement run (one gpurun call).  Outputs under gpurun_out/r2f/; tools/r2_collect.py copies the judged summaries to profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --cpu-sample 0 --steps 20 --warmup 3 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench 20 steps rc=$?"
python bench.py --scan host --cpu-sample 0 > $O/bench_host_scan.json 2> $O/bench_host_scan.err; echo "bench host-scan rc=$?"
python tools/scan_device_bench.py 128 3 > $O/scan_device_bench.txt 2>&1; echo "scan bench rc=$?"
mkdir -p $O/wq && ( cd $O/wq && python $R/wq synthetic:llama3-8b model.layers --backend hip --no-plots --compression-config $R/compression_configs/greedy_seed123.json > $O/wq_llama.log 2>&1; echo "wq llama rc=$?" ); grep streamed $O/wq_llama.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $R/bench.py --cpu-sample 0 --steps 50 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "prof bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k1 -- python $R/tools/k1_bench.py 128 10 0xE > $O/k1_only.log 2>&1; echo "prof k1 rc=$?"; grep "K1 n=" $O/k1_only.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/tools/k1_bench.py 128 3 0xE > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/tools/k1_bench.py 128 3 0xE > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
