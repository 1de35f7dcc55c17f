set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ab; rm -rf $O; mkdir -p $O/wq
cd $R
show() { python -c "
import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], '%.1f M tiles/s' % (d['value']/1e6), '%.3f ms/step' % d['ms_per_step'], 'K1 %.3f ms' % d['roofline']['launch_ms'], 'alone %.3f' % d['roofline']['kernel_alone']['launch_ms'], 'cpu %.1f' % d['config']['host_cpu_ms_per_step'])" $1 $2; }
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_cli.py -x -q -m gpu 2>&1 | tail -1
python bench.py --cpu-sample 0 > $O/b.json 2>/dev/null; show $O/b.json default
python bench.py --cpu-sample 0 --steps 20 --warmup 3 > $O/b20.json 2>/dev/null; show $O/b20.json s20
python bench.py --cpu-sample 0 --steps 20 --warmup 3 > $O/b20b.json 2>/dev/null; show $O/b20b.json s20
MTQ_PIPE_SLOTS=2 python bench.py --cpu-sample 0 > $O/bs2.json 2>/dev/null; show $O/bs2.json slots2
MTQ_PIPE_SLOTS=4 python bench.py --cpu-sample 0 > $O/bs4.json 2>/dev/null; show $O/bs4.json slots4
(cd $O/wq && python $R/wq synthetic:llama3-8b model.layers --backend hip --no-plots --compression-config $R/compression_configs/greedy_seed123.json > $O/wq.log 2>&1; echo "wq rc=$? $(grep streamed $O/wq.log | cut -c60-160)")
