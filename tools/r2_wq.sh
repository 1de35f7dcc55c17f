set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2w; rm -rf $O; mkdir -p $O/wq
cd $R && timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "device_scan" 2>&1 | tail -2
cd $O/wq
run() { tag=$1; shift; env "$@" python $R/wq synthetic:llama3-8b model.layers --backend hip --no-plots --compression-config $R/compression_configs/greedy_seed123.json > $O/wq_$tag.log 2>&1; echo "$tag rc=$? $(grep streamed $O/wq_$tag.log | cut -c60-160)"; rm -rf results; }
run slots3 MTQ_WQ_MAX_SLOTS=3
run slots5 MTQ_WQ_MAX_SLOTS=5
cd $R && python tools/scan_device_bench.py 8 3 14336 4096
