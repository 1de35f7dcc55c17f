#!/bin/bash
# bench.py under several environments, interleaved twice on one box.  usage: [BENCH_ARGS="--workload llama3-8b"] bash tools/r3_env_ab.sh outdir steps "ENV1=.. ENV2=.." "ENV.." ...
out=$1; steps=$2; shift 2
mkdir -p "$out"
for round in 1 2; do
  i=0
  for envs in "$@"; do
    i=$((i+1))
    env $envs python bench.py --steps "$steps" --warmup 5 --cpu-sample 0 --legs none $BENCH_ARGS > "$out/v${i}_r${round}.json" 2>> "$out/err.txt" || { echo "variant $i failed"; tail -3 "$out/err.txt"; }
    python - "$out/v${i}_r${round}.json" "$envs" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
k1 = f", K1 {r['launch_ms']:.3f} ms in the bench, alone {r['kernel_alone']['launch_ms']:.3f}" if "kernel_alone" in r else f", K1 frac {r['frac']:.3f}"
print(f"[{sys.argv[2]}] {d['value']/1e6:.1f} M tiles/s, {d['ms_per_step']:.3f} ms/step{k1}")
PY
  done
done
