#!/bin/bash
# A/B of the bounce kernel's secondary rays through the hand-written loop (RT_EX_SECONDARY_ASM, rt_kernels.hip): variants built with
# tools/build_variant.sh s<k> -DRT_EX_SECONDARY_ASM=<k>;  on the GPU box:  bash tools/ab_ex_secondary.sh <outdir>
out=$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
bash tools/ab_variants.sh $out/c3 python3 tools/kbench.py --ex 64,8,1 --iters 8 && cp $out/c3/ab.log $out/ab_c3.log && \
bash tools/ab_variants.sh $out/atrium python3 tools/kbench.py --scene atrium --width 3840 --height 2160 --ex 16,2,1 --iters 3 && cp $out/atrium/ab.log $out/ab_atrium_bounces.log
rm -f $out/*/librt_hip_saved.so
