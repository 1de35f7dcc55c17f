# bench.py at several K1 chunk sizes on one box, interleaved: bash tools/ab_chunk.sh [rounds]
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
N=${1:-2}
for i in $(seq 1 $N); do
  for c in 16 32 64 8; do
    out=$(MTQ_CHAIN_RECORDS=0 python bench.py --cpu-sample 0 --chunk $c 2>/dev/null)
    echo "chunk $c: $(echo "$out" | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1e6,1),'M tiles/s', round(d['ms_per_step'],3),'ms/step  K1 launch', round(d['roofline']['launch_ms'],4), 'ms x', d['roofline']['launches'])")"
  done
done
