set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ab; rm -rf $O; mkdir -p $O/wq
cd $R
show() { python -c "
import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], '%.1f M tiles/s' % (d['value']/1e6), '%.3f ms/step' % d['ms_per_step'], 'K1 %.3f ms' % d['roofline']['launch_ms'], 'alone %.3f' % d['roofline']['kernel_alone']['launch_ms'], 'cpu %.1f' % d['config']['host_cpu_ms_per_step'])" $1 $2; }
run() { tag=$1; shift; env "$@" python bench.py --cpu-sample 0 > $O/b_$tag.json 2>/dev/null; show $O/b_$tag.json $tag; }
wq() { tag=$1; shift; (cd $O/wq && env "$@" python $R/wq synthetic:llama3-8b model.layers --backend hip --no-plots --compression-config $R/compression_configs/greedy_seed123.json > $O/wq_$tag.log 2>&1; echo "wq $tag rc=$? $(grep streamed $O/wq_$tag.log | cut -c60-160)"; rm -rf results); }
run p0a MTQ_SCAN_WAVE_PRIO=0
run p3a MTQ_SCAN_WAVE_PRIO=3
run p0b MTQ_SCAN_WAVE_PRIO=0
run p3b MTQ_SCAN_WAVE_PRIO=3
run p1 MTQ_SCAN_WAVE_PRIO=1
wq p0 MTQ_SCAN_WAVE_PRIO=0 MTQ_WQ_DEVICE_SCAN_MAX_TILES=4194304
wq p3 MTQ_SCAN_WAVE_PRIO=3 MTQ_WQ_DEVICE_SCAN_MAX_TILES=4194304
wq p3b MTQ_SCAN_WAVE_PRIO=3 MTQ_WQ_DEVICE_SCAN_MAX_TILES=4194304
