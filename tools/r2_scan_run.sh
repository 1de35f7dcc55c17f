set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2s; rm -rf $O; mkdir -p $O
cd $R
for n in 1 8 32 96; do timeout -k 10 300 python tools/scan_device_bench.py $n 3 14336 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/scan_big.txt || exit 1; done
for n in 64 256; do timeout -k 10 300 python tools/scan_device_bench.py $n 3 4096 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/scan_big.txt || exit 1; done
