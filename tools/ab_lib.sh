# bench.py with two builds of the library, interleaved on one box: bash tools/ab_lib.sh path/to/other.so [rounds]
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
OTHER=$(realpath $1); N=${2:-4}
for i in $(seq 1 $N); do
  for lib in default $OTHER; do
    if [ $lib = default ]; then out=$(python bench.py --cpu-sample 0 2>/dev/null); else out=$(MTQ_LIB=$lib python bench.py --cpu-sample 0 2>/dev/null); fi
    echo "$(basename $lib): $(echo "$out" | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1e6,1),'M tiles/s', round(d['ms_per_step'],3),'ms/step')")"
  done
done
uptime
