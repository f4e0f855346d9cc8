#!/bin/bash
# The round's final bench lines on the GPU box, without a profiler: one JSON line per file under <outdir>.
#   bash tools/final_lines.sh <outdir>       then, in the build container:  cp <outdir>/*.json profiles/ (named r06_final_bench_*.json there)
out=$1
mkdir -p $out
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; echo "== $name: bench.py $*"; timeout -k 10 500 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "FAILED $name"; tail -5 $out/$name.err; return 1; }; cut -c1-160 $out/$name.json; }
run driver --gpus 1 --steps 20 --warmup 5 && \
run default && \
run c2_far --camera far --no-cpu-baseline && \
run c2_near --camera near --no-cpu-baseline && \
run demo --workload demo && \
run demo_baked --workload demo --baked --no-cpu-baseline && \
run c3 --workload c3 && \
run c4 --workload c4 && \
run c4_1spp --workload c4 --spp 1 --no-cpu-baseline && \
run c5 --workload c5 --steps 5 --warmup 1 --no-cpu-baseline && \
run c6 --workload c6 --no-cpu-baseline && \
run c6_incoherent --workload c6 --bounces 2 --metallic 1 --roughness 0.3 --steps 6 --warmup 2 --no-cpu-baseline && \
run tiled_driver_shape_forced_collective --force-collective --steps 20 --warmup 5 --no-cpu-baseline && \
run tiled_driver_shape_forced_collective_one_launch --force-collective --steps 20 --warmup 5 --no-cpu-baseline --subgroups 1 && \
run predicted_scaling --predict-scaling 2,4,8 && \
run predicted_scaling_c5 --workload c5 --predict-scaling 8
