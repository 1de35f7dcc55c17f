set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2w; rm -rf $O; mkdir -p $O/wq
cd $R && timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_cli.py tests/test_configs_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
cd $O/wq
run() { tag=$1; shift; env "$@" python $R/wq synthetic:llama3-8b model.layers --backend hip --no-plots --compression-config $R/compression_configs/greedy_seed123.json > $O/wq_$tag.log 2>&1; echo "$tag rc=$? $(grep streamed $O/wq_$tag.log | cut -c60-160)"; rm -rf results; }
run hyb_a X=1
run dev_a MTQ_WQ_DEVICE_SCAN_MAX_TILES=4194304
run hyb_b X=1
run hyb16k MTQ_WQ_DEVICE_SCAN_MAX_TILES=16383
run host MTQ_DEVICE_SCAN=0
run hyb_trace MTQ_PIPE_TRACE=1
