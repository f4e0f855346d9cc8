rm -rf gpurun_out/prof_v6
bash tools/profile_bench.sh gpurun_out/prof_v6 | tail -1 | cut -c1-100
