set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2z; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 700 python tests/fuzz_parity.py 2000 20261004 > $O/fuzz_2000.txt 2>&1; echo "fuzz rc=$? $(tail -1 $O/fuzz_2000.txt)"
timeout -k 10 300 python tools/scan_device_bench.py 256 1 4096 4096 2>&1 | grep -v amdgpu.ids | tee $O/scan_256_distinct_seeds.txt
timeout -k 10 300 python tools/scan_device_bench.py 24 1 14336 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/scan_256_distinct_seeds.txt
timeout -k 10 300 python tools/scan_device_bench.py 64 1 1024 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/scan_256_distinct_seeds.txt
