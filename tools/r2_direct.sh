set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_algorithms.py tests/test_fuzz_parity.py tests/test_loader_fp8.py -x -q -m gpu 2>&1 | tail -3
python tools/k1_f32_bench.py 8 10 normal 2>&1 | tail -5
