cd $GRAFT_REPO_ROOT
run() { echo "== $*"; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('contexts_per_gpu'), d['config'].get('host_threads_per_gpu'))"; }
B="python bench.py --workload config3 --no-cpu-baseline --steps 6 --warmup 2"
run $B
run $B --contexts 4 --threads 4
run $B --contexts 6 --threads 2
run $B --contexts 8 --threads 2
run $B --contexts 8 --threads 4
run env GPU_MAX_HW_QUEUES=8 $B --contexts 8 --threads 2
run env GPU_MAX_HW_QUEUES=8 $B --contexts 8 --threads 4
run $B --contexts 2 --threads 2
