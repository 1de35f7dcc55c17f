# Interleaved A/B runs of bench.py on ONE box (boxes differ by a few per cent: only runs of the same call compare).
# Edit the `run` lines; the variants of round 2 are listed in DESIGN.md §4 ("Living beside the scan kernels"), §5 and §6b.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ab; rm -rf $O; mkdir -p $O
cd $R
show() { python -c "
import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], '%.1f M tiles/s' % (d['value']/1e6), '%.3f ms/step' % d['ms_per_step'], 'K1 %.3f ms' % d['roofline']['launch_ms'], 'alone %.3f' % d['roofline']['kernel_alone']['launch_ms'], 'cpu %.1f' % d['config']['host_cpu_ms_per_step'])" $1 $2; }
run() { tag=$1; shift; env "$@" python bench.py --cpu-sample 0 > $O/b_$tag.json 2>/dev/null; show $O/b_$tag.json $tag; }
run upw8_a MTQ_K1_UNITS_PER_WAVE=8
run upw12_a MTQ_K1_UNITS_PER_WAVE=12
run upw16_a MTQ_K1_UNITS_PER_WAVE=16
run upw32_a MTQ_K1_UNITS_PER_WAVE=32
run upw8_b MTQ_K1_UNITS_PER_WAVE=8
run upw12_b MTQ_K1_UNITS_PER_WAVE=12
run upw16_b MTQ_K1_UNITS_PER_WAVE=16
run upw32_b MTQ_K1_UNITS_PER_WAVE=32
