cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r01_v2.json
python bench.py --frames-per-launch 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
