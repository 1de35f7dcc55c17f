set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ab; rm -rf $O; mkdir -p $O
cd $R
show() { python -c "
import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], '%.1f M tiles/s' % (d['value']/1e6), '%.3f ms/step' % d['ms_per_step'], 'K1 %.3f ms' % d['roofline']['launch_ms'], 'alone %.3f' % d['roofline']['kernel_alone']['launch_ms'])" $1 $2; }
for w in 2 1; do for u in 8 16 32 64; do MTQ_K1_UNITS_PER_WAVE=$u MTQ_LIB=$R/quantization_analysis_amd/libmtq_hip_w$w.so python bench.py --cpu-sample 0 > $O/b_w${w}_u$u.json 2>/dev/null; show $O/b_w${w}_u$u.json waves${w}_upw$u; done; done
