rm -rf gpurun_out/prof_v7
bash tools/profile_bench.sh gpurun_out/prof_v7 --steps 192 --warmup 32 | tail -1 | cut -c1-100
cd $GRAFT_REPO_ROOT; python bench.py --latency-probe 2>&1 | tail -1 > gpurun_out/bench_r01_final.json
