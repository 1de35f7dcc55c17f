# A/B of an environment switch on one box, interleaved: bash tools/ab_env.sh VAR A B [rounds] [bench flags...]
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
V=$1; A=$2; B=$3; N=${4:-3}; shift 4
for i in $(seq 1 $N); do
  for val in $A $B; do
    out=$(env $V=$val python bench.py --cpu-sample 0 "$@" 2>/dev/null)
    echo "$V=$val $(echo "$out" | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1e6,1),'M tiles/s', round(d['ms_per_step'],3),'ms/step  K1', round(d['roofline']['launch_ms'],4))")"
  done
done
uptime
