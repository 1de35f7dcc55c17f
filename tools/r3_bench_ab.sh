#!/bin/bash
# Round 3: the bench line on the lazy route, on the whole-record route with shared orders, and in the round-2 form (own shuffles), one box.
# usage: bash tools/r3_bench_ab.sh [outdir] [steps]
out=${1:-gpurun_out/r3d}; steps=${2:-20}
mkdir -p "$out"
python bench.py --steps "$steps" --warmup 5 --cpu-sample 0 > "$out/bench_lazy.json" 2> "$out/bench.err" || exit 1
python bench.py --steps 100 --warmup 5 --cpu-sample 0 > "$out/bench_lazy100.json" 2>> "$out/bench.err" || exit 1
MTQ_LAZY=0 python bench.py --steps "$steps" --warmup 5 --cpu-sample 0 > "$out/bench_full_shared.json" 2>> "$out/bench.err" || exit 1
MTQ_LAZY=0 MTQ_SHARED_ORDERS=0 python bench.py --steps "$steps" --warmup 5 --cpu-sample 0 > "$out/bench_r2form.json" 2>> "$out/bench.err" || exit 1
python bench.py --steps "$steps" --warmup 5 --cpu-sample 0 > "$out/bench_lazy_b.json" 2>> "$out/bench.err" || exit 1
for f in bench_lazy bench_lazy100 bench_full_shared bench_r2form bench_lazy_b; do
python - "$out/$f.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"{sys.argv[1]}: {d['value']/1e6:.1f} M tiles/s, {d['ms_per_step']:.3f} ms/step, K1 {r['launch_ms']:.3f} ms in the bench (frac {r['frac']:.4f}), alone {r['kernel_alone']['launch_ms']:.3f} ms (frac {r['kernel_alone']['frac']:.4f}), handed back {d['config']['host_fallbacks']}, counts {d['summary']['counts_bf16_bfp8_bfp4_bfp2']}")
PY
done
