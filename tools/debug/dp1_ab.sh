export SAM_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
for late in 1 0; do
for i in 1 2; do SAM_WGRAD_LATE=$late python bench.py --gpus 1 --steps 60 --warmup 8 --no-eager-baseline --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('late=$late', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_p10_p90_max'], d.get('exposed_comm_ms'), d.get('rccl_ranks_seen'))"; done; done
