#!/bin/bash
# Profiles `python3 bench.py <args>` on the GPU box: one rocprofv3 --kernel-trace --stats run of the command as given, then
# separate PMC passes (one counter group per pass, as MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes; never combined
# with a trace) of a shorter run of the same workload.  Summarise with tools/summarize_profile.py.
# Usage: bash tools/profile_bench.sh <outdir> <pmc_steps> <pmc_warmup> [bench args...]
out=$1; psteps=$2; pwarm=$3; shift 3
export TMPDIR=/tmp
mkdir -p $out
cd $GRAFT_REPO_ROOT
echo "== stats: bench.py $*" | tee -a $out/progress.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-latency "$@" > $out/stats.log 2>$out/stats.err
echo "rc=$?" | tee -a $out/progress.log
BENCH_ARGS="$*"
pmc() { name=$1; shift; echo "== $name" | tee -a $out/progress.log; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py $BENCH_ARGS --steps $psteps --warmup $pwarm --no-cpu-baseline --no-latency > $out/$name.log 2>$out/$name.err; rc=$?; echo "rc=$rc" | tee -a $out/progress.log; return $rc; }
pmc pmc_fetch FETCH_SIZE && pmc pmc_write WRITE_SIZE \
 && pmc pmc_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
 && pmc pmc_sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH \
 && pmc pmc_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
tail -1 $out/stats.log | cut -c1-400
