#!/bin/bash
# A/B of the hand-written loop's round-6 forms (RT_ASM_V2 / RT_ASM_LAYOUT / RT_ASM_ALIGN / RT_SLAB_MASKS, rt_kernels.hip) against round 5's:
# build the variants first (tools/build_variant.sh <name> <flags>: profiles/r06_experiments/asm_loop_v2.md lists them), then on the GPU box
#   bash tools/ab_asm_v2.sh <outdir> [quick]
out=$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
bash tools/ab_variants.sh $out/f32 python3 tools/kbench.py --batch 32 --iters 40 --check && cp $out/f32/ab.log $out/ab_batch32.log && \
bash tools/ab_variants.sh $out/c4 python3 tools/kbench.py --scene atrium --width 3840 --height 2160 --ex 16,0,0 --iters 20 && cp $out/c4/ab.log $out/ab_c4_16spp.log || exit 1
if [ "$2" != quick ]; then
bash tools/ab_variants.sh $out/f1 python3 tools/kbench.py --batch 1 --iters 300 --path 20 && cp $out/f1/ab.log $out/ab_single_frames.log && \
bash tools/ab_variants.sh $out/c3 python3 tools/kbench.py --ex 64,8,1 --iters 8 && cp $out/c3/ab.log $out/ab_c3.log
fi
rm -f $out/*/librt_hip_saved.so
