#!/bin/bash
# A/B of kernel build variants on the GPU box: for every cuda-raytracing_amd/_variants/librt_hip_<name>.so (built locally with
# RT_HIPCC_EXTRA=... python cuda-raytracing_amd/_build.py --force), copy it over librt_hip.so and run the given command.
#   bash tools/ab_variants.sh <outdir> <command...>
out=$1; shift
mkdir -p $out
cp cuda-raytracing_amd/librt_hip.so $out/librt_hip_saved.so
# whatever ends this script (a failing command, a signal, the last line) puts the shipped library back.  Every variant carries
# the hash of the sources + flags it was built from (rt_build_info); the loader refuses a library that is not the build of the
# tree unless RT_ALLOW_VARIANT_LIB=1, and bench.py prices its line with the hash of the library that ran -- so a variant's
# line says profile_stale instead of borrowing the shipped kernel's counters, and the log names each variant's hash.
export RT_ALLOW_VARIANT_LIB=1
trap 'cp $out/librt_hip_saved.so cuda-raytracing_amd/librt_hip.so; touch cuda-raytracing_amd/librt_hip.so cuda-raytracing_amd/librt_host.so' EXIT
for lib in cuda-raytracing_amd/_variants/librt_hip_*.so; do
    name=$(basename $lib .so); name=${name#librt_hip_}
    cp $lib cuda-raytracing_amd/librt_hip.so
    touch cuda-raytracing_amd/librt_hip.so cuda-raytracing_amd/librt_host.so     # (newer than the sources: no rebuild)
    for rep in 1 2; do
        echo "== variant $name (run $rep) code hash $(python3 -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('cuda-raytracing_amd._build').library_code_hash())")" | tee -a $out/ab.log
        "$@" 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.log
    done
done
