#!/bin/bash
# Build one kernel variant for tools/ab_variants.sh:   bash tools/build_variant.sh <name> [extra hipcc flags...]
# -> cuda-raytracing_amd/_variants/librt_hip_<name>.so (carrying the hash of sources + flags it was built from); the shipped
# library is left as it was.  Only librt_hip.so differs between variants: the C-ABI does not change with these switches.
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p cuda-raytracing_amd/_variants
hash=$(RT_HIPCC_EXTRA="$*" python3 cuda-raytracing_amd/_build.py --print-hash)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared "$@" -DRT_CODE_HASH="\"$hash\"" \
    -o cuda-raytracing_amd/_variants/librt_hip_$name.so cuda-raytracing_amd/csrc/rt_kernels.hip cuda-raytracing_amd/csrc/rt_bvh_build.hip cuda-raytracing_amd/csrc/rt_comm.hip
echo "built variant $name ($hash): $*"
