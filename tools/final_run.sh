set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-sample 0 > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "torchrun rc=$?"
HSA_ENABLE_SDMA=0 python bench.py --cpu-sample 0 > $O/bench_sdma_off.json 2>/dev/null; echo "sdma-off rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $R/bench.py --cpu-sample 0 > $O/bench_under_rocprof.json 2>&1; echo "prof bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k1 -- python $R/tools/k1_bench.py 32 10 0xE > $O/k1_only.log 2>&1; echo "prof k1 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k1f32 -- python $R/tools/k1_f32_bench.py 8 10 normal > $O/k1_f32.log 2>&1; echo "prof k1 f32 rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/tools/k1_bench.py 32 3 0xE > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/tools/k1_bench.py 32 3 0xE > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
python $R/tools/k23_bench.py > $O/k23.log 2>&1; echo "k23 rc=$?"
