#!/bin/bash
# A/B of kernel build variants on the GPU box: for every cuda-raytracing_amd/_variants/librt_hip_<name>.so (built locally with
# RT_HIPCC_EXTRA=... python cuda-raytracing_amd/_build.py --force), copy it over librt_hip.so and run the given command.
#   bash tools/ab_variants.sh <outdir> <command...>
out=$1; shift
mkdir -p $out
cp cuda-raytracing_amd/librt_hip.so $out/librt_hip_saved.so
# whatever ends this script (a failing command, a signal, the last line) puts the shipped library back: bench.py's code hash is
# computed from the sources, so a variant left installed would be priced as the shipped kernel
trap 'cp $out/librt_hip_saved.so cuda-raytracing_amd/librt_hip.so; touch cuda-raytracing_amd/librt_hip.so cuda-raytracing_amd/librt_host.so' EXIT
for lib in cuda-raytracing_amd/_variants/librt_hip_*.so; do
    name=$(basename $lib .so); name=${name#librt_hip_}
    cp $lib cuda-raytracing_amd/librt_hip.so
    touch cuda-raytracing_amd/librt_hip.so cuda-raytracing_amd/librt_host.so     # (newer than the sources: no rebuild)
    for rep in 1 2; do
        echo "== variant $name (run $rep)" | tee -a $out/ab.log
        "$@" 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.log
    done
done
