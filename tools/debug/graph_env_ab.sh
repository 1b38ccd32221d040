run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 60 --warmup 6 --no-secondary --no-roofline --no-cpu-baseline --no-eager-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
run A=1
run SAM_NO_TB_OVERLAP=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run SAM_NO_TB_OVERLAP=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run A=1
