set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2t; rm -rf $O; mkdir -p $O/wq
cd $O/wq
run() { tag=$1; shift; env "$@" python $R/wq synthetic:llama3-8b model.layers --backend hip --no-plots --compression-config $R/compression_configs/greedy_seed123.json > $O/wq_$tag.log 2>&1; echo "$tag rc=$? $(grep streamed $O/wq_$tag.log | cut -c60-160)"; rm -rf results; }
run hyb_t MTQ_PIPE_TRACE=1
grep "\[pipe\]" $O/wq_hyb_t.log | cut -c1-200
