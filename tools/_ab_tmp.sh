cd $GRAFT_REPO_ROOT
for g in on off; do
python bench.py --config BENCHMARK1 --steps 30 --warmup 3 --no-cpu-baseline --loopback --graph-exchanges $g 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B1 loopback graph=$g', d['ms_per_step'], sum(d['kernel_ms'].values()), d['kernel_ms']['step2d_loop'], d['config'].get('graph_exchanges'))"
done
cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/lb_graph -o lb --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config BENCHMARK1 --steps 30 --warmup 3 --no-cpu-baseline --loopback --graph-exchanges on > /dev/null 2>&1
head -12 $GRAFT_REPO_ROOT/gpurun_out/lb_graph/*/lb_kernel_stats.csv 2>/dev/null | cut -c1-150 || head -12 $GRAFT_REPO_ROOT/gpurun_out/lb_graph/lb_kernel_stats.csv | cut -c1-150
