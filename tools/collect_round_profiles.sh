TAG=${1:-r4p}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/profile_bench.sh ${TAG} --no-secondary
cd $GRAFT_REPO_ROOT
python tools/bench_wgrad.py 11648 > gpurun_out/${TAG}_micro_wgrad.txt 2>&1
python tools/bench_wgrad.py 1280 >> gpurun_out/${TAG}_micro_wgrad.txt 2>&1
python tools/bench_attn.py > gpurun_out/${TAG}_micro_attn.txt 2>&1
python tools/bench_gemm8.py > gpurun_out/${TAG}_micro_gemm.txt 2>&1
python tools/bench_adam.py > gpurun_out/${TAG}_micro_adam.txt 2>&1
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json
python tools/bench_eval.py > gpurun_out/${TAG}_eval.txt 2>&1
