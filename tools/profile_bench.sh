#!/bin/bash
# Profiles `python3 bench.py` on the GPU box: kernel-trace stats + separate PMC passes (one counter group per pass,
# as MI355X_MICROARCH.md prescribes).  Usage: bash tools/profile_bench.sh <outdir> [bench args...]
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
cd $GRAFT_REPO_ROOT
echo "== stats" | tee -a $out/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline "$@" > $out/stats.log 2>&1
echo "rc=$?" | tee -a $out/progress.log
pmc() { name=$1; shift; echo "== $name" | tee -a $out/progress.log; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --steps 64 --warmup 32 --no-cpu-baseline $BENCH_ARGS > $out/$name.log 2>&1; echo "rc=$?" | tee -a $out/progress.log; }
BENCH_ARGS="$*"
pmc pmc_fetch FETCH_SIZE && pmc pmc_write WRITE_SIZE && pmc pmc_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum \
 && pmc pmc_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
 && pmc pmc_sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
 && pmc pmc_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
tail -1 $out/stats.log | cut -c1-300
