rm -rf gpurun_out/prof_v3
bash tools/profile_bench.sh gpurun_out/prof_v3
cd $GRAFT_REPO_ROOT; python bench.py --latency-probe --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
