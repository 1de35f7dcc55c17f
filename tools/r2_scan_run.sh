set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2s; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "streamed_pipeline" 2>&1 | tail -2
show() { python -c "
import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], '%.1f M tiles/s' % (d['value']/1e6), '%.3f ms/step' % d['ms_per_step'], 'K1 %.3f ms' % d['roofline']['launch_ms'], 'host cpu %.1f ms/step' % d['config']['host_cpu_ms_per_step'], d['config'].get('driver_thread_ms_per_step'))" $1 $2; }
for c in 128 64 32; do python bench.py --cpu-sample 0 --scan device --chunk $c > $O/b_c$c.json 2>/dev/null; show $O/b_c$c.json chunk$c; done
MTQ_K1_UNITS_PER_WAVE=0 python bench.py --cpu-sample 0 --scan device --chunk 128 > $O/b_u0.json 2>/dev/null; show $O/b_u0.json chunk128_persistentK1
python bench.py --cpu-sample 0 --scan host > $O/b_host.json 2>/dev/null; show $O/b_host.json host
