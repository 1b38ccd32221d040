cd $GRAFT_REPO_ROOT
F="--no-secondary --no-cpu-baseline --no-eager-baseline --no-roofline"
for a in "--steps 60 --warmup 15" "--steps 120 --warmup 10" "--steps 400 --warmup 10" "--steps 60 --warmup 15"; do python bench.py $a $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RUN','$a', d['ms_per_step'], d['value'])"; done
python bench.py --no-cpu-baseline --no-eager-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RUN default+roofline+secondary', d['ms_per_step'], d['value'])"
