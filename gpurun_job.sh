cd $GRAFT_REPO_ROOT
python tools/kbench.py --iters 5 --ex 1,0,1 2>&1 | grep -v "^Loading\|^OBJ\|^Loaded\|^scene"
python tools/kbench.py --iters 3 --ex 64,8,1 2>&1 | grep -v "^Loading\|^OBJ\|^Loaded\|^scene"
python tools/kbench.py --iters 3 --ex 16,0,0 --scene atrium --width 3840 --height 2160 2>&1 | grep -v "^Loading\|^OBJ\|^Loaded\|^scene"
rm -rf gpurun_out/prof_v5; bash tools/profile_bench.sh gpurun_out/prof_v5 | tail -2
python bench.py --latency-probe 2>&1 | tail -1 > gpurun_out/bench_r01_final.json; cut -c1-300 gpurun_out/bench_r01_final.json
