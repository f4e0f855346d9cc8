#!/bin/bash
# rocprofv3 kernel-trace summaries of the kernels beside the two render kernels: the GPU BVH build, the un-stripe pass
# (+ RCCL gather / all-to-all with the one rank of a one-GPU box), the resolve pass.  Usage: bash tools/profile_aux.sh <outdir>
out=$1
export TMPDIR=/tmp
mkdir -p $out
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; echo "== $name: $*" | tee -a $out/progress.log
        timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- "$@" > $out/$name.log 2> $out/$name.err; echo "rc=$?" | tee -a $out/progress.log; }
run bvh python3 tools/bvh_build_bench.py
run tiled_rotate python3 bench.py --force-collective --no-cpu-baseline --no-latency --steps 320 --warmup 64
run tiled_root0 python3 bench.py --force-collective --no-cpu-baseline --no-latency --steps 320 --warmup 64 --gather root0
run tiled_c5 python3 bench.py --force-collective --no-cpu-baseline --workload c5 --steps 2 --warmup 1
for n in bvh tiled_rotate tiled_root0 tiled_c5; do f=$(ls $out/$n/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/${n}_kernel_stats.csv; done
ls $out/*_kernel_stats.csv
