#!/bin/bash
# The round's profile campaign on the GPU box, in parts that fit one gpurun call each:
#   bash tools/profile_campaign.sh <outdir> <part>     part = c2 | demo | ex | big | aux
# Every workload: tools/profile_bench.sh (one rocprofv3 --kernel-trace --stats run of the bench command + separate PMC passes).
# Afterwards, in the build container:  python tools/summarize_profile.py <outdir>/<name> r06_<name>   for every <name>.
out=$1; part=$2
mkdir -p $out
cd $GRAFT_REPO_ROOT
run() { name=$1; ps=$2; pw=$3; shift 3; echo "=== $name: $*"; bash tools/profile_bench.sh $out/$name $ps $pw "$@" > $out/$name.log 2>&1; tail -2 $out/$name.log | cut -c1-300; }
case $part in
c2)  run c2_mid 96 32
     run c2_far 96 32 --camera far
     run c2_near 96 32 --camera near
     run c2_mid_f1 48 16 --frames-per-launch 1 --steps 400 ;;
demo) run demo 96 32 --workload demo
     run demo_baked 96 32 --workload demo --baked ;;
ex)  run c3 3 1 --workload c3 --steps 20 --warmup 3
     run c4_16spp 3 1 --workload c4 --steps 20 --warmup 3
     run c4_1spp 64 32 --workload c4 --spp 1 --steps 256 --warmup 32 ;;
big) run c5 1 1 --workload c5 --steps 4 --warmup 1
     run c6 32 32 --workload c6 --steps 128 --warmup 32
     run c6_incoherent 2 1 --workload c6 --bounces 2 --metallic 1 --roughness 0.3 --steps 6 --warmup 2 ;;
aux) bash tools/profile_aux.sh $out/aux > $out/aux.log 2>&1; tail -3 $out/aux.log
     export TMPDIR=/tmp
     mkdir -p $out/driver
     timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/driver/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/driver/driver_line.json 2> $out/driver/driver.err; echo "driver rc $?" ;;
esac
