set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ab; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_algorithms.py tests/test_cli.py tests/test_configs_gpu.py tests/test_sweep.py -x -q -m gpu 2>&1 | tail -2
for t in $(ls tools | grep -i thr); do echo "== $t"; timeout -k 10 300 python tools/$t 2>&1 | grep -v amdgpu.ids | tail -6; done
