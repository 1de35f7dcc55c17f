cd $GRAFT_REPO_ROOT
st() { grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
for w in 16 14 13 12 11 16 13; do
  echo "workers $w"; st
  python bench.py --cpu-sample 0 --steps 200 --workers $w 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1e6,1),'M tiles/s', round(d['ms_per_step'],3))"
  st
done
uptime
