#!/bin/bash
# Round-4 measurement run (one gpurun call, ~6 min).  Outputs under gpurun_out/<tag>/ (default r4f); tools/r4_collect.py <tag> copies the
# judged summaries into profiles/ and recomputes profiles/k1_traffic.json and k1_valu.json from the PMC passes.
#   usage: bash tools/r4_final.sh [tag] [what: all|bench|prof|pmc|wq|fuzz ...]
set -o pipefail
R=$GRAFT_REPO_ROOT; tag=${1:-r4f}; shift; what=${*:-all}; O=$R/gpurun_out/$tag; mkdir -p $O
has() { [[ " $what " == *" all "* || " $what " == *" $1 "* ]]; }
cd $R
if has tests; then timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log; python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log)"; fi
if has bench; then
  python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  python bench.py --cpu-sample 0 --legs none --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench 20 steps rc=$?"
  MTQ_LAZY=0 python bench.py --cpu-sample 0 --legs none --steps 20 --warmup 5 > $O/bench_whole_records.json 2> $O/bench_whole.err; echo "bench whole records rc=$?"
  MTQ_LAZY=0 MTQ_SHARED_ORDERS=0 python bench.py --cpu-sample 0 --legs none --steps 20 --warmup 5 > $O/bench_r2_form.json 2>> $O/bench_whole.err; echo "bench round-2 form rc=$?"
  python bench.py --workload llama3-8b --steps 10 --warmup 2 > $O/bench_llama3_8b.json 2> $O/bench_llama.err; echo "bench llama3-8b rc=$?"
  python tools/scan_v3_check.py 128 5 > $O/scan_v3_check.txt 2>&1; echo "scan check rc=$?"
  python tools/threshold_pipeline_bench.py 64 > $O/threshold_pipeline.txt 2>&1; echo "threshold pipeline rc=$?"
  python tools/k1_listed_bench.py 128 11 > $O/k1_listed.txt 2>&1; echo "listed K1 rc=$? $(tail -1 $O/k1_listed.txt)"
  for m in "0xE 0x2 0x4" "0xE 0xE 0x0"; do python tools/k1_partial_bench.py 128 11 $m 2>&1 | grep "K1 partial"; MTQ_LIB=$R/build/libmtq_intdom.so python tools/k1_partial_bench.py 128 11 $m 2>&1 | grep "K1 partial" | sed "s/^/[round-3 packed-integer form, -DMTQ_K1_INTDOM] /"; done > $O/k1_f32dom_vs_intdom.txt 2>&1; cat $O/k1_f32dom_vs_intdom.txt
  [ -x build/ubench_f32 ] && build/ubench_f32 > $O/ubench_f32.txt 2>&1
fi
if has wq; then bash tools/r3_exit_check.sh gpurun_out/$tag/exit > $O/exit_check.txt 2>&1; echo "exit check: $(grep -c 'signal reports: 0' $O/exit_check.txt) of 2 clean"; fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --cpu-sample 0 --legs none --steps 50 --regions 1 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "prof bench rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k1 -- python3 $R/tools/k1_partial_bench.py 128 10 0xE 0x2 0x4 > $O/k1_only.log 2>&1; echo "prof k1 rc=$?"; grep "K1 " $O/k1_only.log
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python3 $R/tools/k1_f32_bench.py 8 10 normal > $O/k1_f32.log 2>&1; echo "prof k1 f32 rc=$?"; grep "mask=0xf" $O/k1_f32.log
fi
if has pmc; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/k1_partial_bench.py 128 3 0xE 0x2 0x4 > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/k1_partial_bench.py 128 3 0xE 0x2 0x4 > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/tools/k1_partial_bench.py 32 3 0xE 0x2 0x4 > $O/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq_f32 -- python3 $R/tools/k1_f32_bench.py 8 3 normal > $O/pmc_sq_f32.log 2>&1; echo "pmc sq f32 rc=$?"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_f32 -- python3 $R/tools/k1_f32_bench.py 8 3 normal > $O/pmc_fetch_f32.log 2>&1; echo "pmc fetch f32 rc=$?"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_f32 -- python3 $R/tools/k1_f32_bench.py 8 3 normal > $O/pmc_write_f32.log 2>&1; echo "pmc write f32 rc=$?"
fi
if has fuzz; then cd $R && timeout -k 10 900 python tests/fuzz_parity.py 2000 20261005 > $O/fuzz_2000.txt 2>&1; echo "fuzz rc=$? $(tail -1 $O/fuzz_2000.txt)"; fi
find $O -name "*.db" -delete 2>/dev/null; du -sh $O | cut -f1
