set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2w; rm -rf $O; mkdir -p $O/wq
cd $O/wq
run() { tag=$1; shift; env "$@" python $R/wq synthetic:llama3-8b model.layers --backend hip --no-plots --compression-config $R/compression_configs/greedy_seed123.json > $O/wq_$tag.log 2>&1; echo "$tag rc=$? $(grep streamed $O/wq_$tag.log | cut -c60-130)"; rm -rf results; }
run s3a MTQ_WQ_MAX_SLOTS=3
run s8a MTQ_WQ_MAX_SLOTS=8
run s3b MTQ_WQ_MAX_SLOTS=3
run s8b MTQ_WQ_MAX_SLOTS=8
