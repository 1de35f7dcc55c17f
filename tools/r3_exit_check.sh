#!/bin/bash
# The command that ended in a SIGSEGV inside __cxa_finalize in round 2 (gpurun_out/r2w/wq.log), once, in its rocprofv3 form (the
# program directly behind `--`), and once without the profiler: exit codes, and whether the log holds a signal report.
# usage: bash tools/r3_exit_check.sh [outdir]
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r3x}; rm -rf $O; mkdir -p $O/run
cd /tmp && export TMPDIR=/tmp
cd $O/run
rocprofv3 --kernel-trace --stats -d $O/prof -- python3 $R/wq synthetic:llama3-8b model.layers --backend hip --no-plots --compression-config $R/compression_configs/greedy_seed123.json > $O/wq_rocprof.log 2>&1
echo "under rocprofv3: rc=$? results: $(grep -c '^results:' $O/wq_rocprof.log) signal reports: $(grep -c -E 'SIGSEGV|Aborted at|core dumped' $O/wq_rocprof.log)"
grep -E "streamed|wall:" $O/wq_rocprof.log | cut -c1-260
rm -rf results
python3 $R/wq synthetic:llama3-8b model.layers --backend hip --no-plots --compression-config $R/compression_configs/greedy_seed123.json > $O/wq_plain.log 2>&1
echo "without profiler: rc=$? results: $(grep -c '^results:' $O/wq_plain.log) signal reports: $(grep -c -E 'SIGSEGV|Aborted at|core dumped' $O/wq_plain.log)"
grep -E "streamed|wall:" $O/wq_plain.log | cut -c1-260
rm -rf results $O/prof/*/*.db 2>/dev/null
ls $O/prof/* 2>/dev/null | head -5
