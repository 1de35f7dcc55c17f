set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2s; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python tools/scan_device_bench.py 128 3 4096 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/scan.txt
timeout -k 10 300 python tools/scan_device_bench.py 32 3 14336 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/scan.txt
MTQ_LIB=$R/quantization_analysis_amd/libmtq_hip_prof.so timeout -k 10 300 python tools/scan_device_bench.py 128 3 4096 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/scan.txt
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_fuzz_parity.py -x -q -m gpu -k "scan or fuzz or pipeline" 2>&1 | tail -2
timeout -k 10 300 python tests/fuzz_parity.py 1500 77 2>&1 | tail -1
