#!/bin/bash
# Profiling session for one BASELINE config: rocprofv3 kernel stats of a bench run (program directly after `--`), three PMC
# passes (each its own run: FETCH_SIZE and WRITE_SIZE need separate TCC slots), the traffic file keyed by build + workload.
#   tools/gpu_profile_cfg.sh TAG CONFIG NAME [extra bench flags]      -> gpurun_out/TAG/NAME/*
set -o pipefail
TAG=${1:-r03prof}; CFG=${2:-2}; NAME=${3:-cfg$CFG}; shift; shift; shift
O=gpurun_out/$TAG/$NAME
mkdir -p $O
export TMPDIR=/tmp
case $CFG in 2) ST=16;; 3) ST=8;; 5) ST=8;; *) ST=8;; esac
rocprofv3 --kernel-trace --stats -d $O/stats -o run -- python bench.py --config $CFG --steps $ST --warmup 2 --repeats 3 --no-cpu-baseline --no-secondary "$@" > $O/bench_under_rocprof.json 2> $O/stats.log; echo "cfg$CFG stats rc=$?"
python tools/pmc_report.py --stats $O/stats/run_results.db > $O/kernel_stats.csv 2> $O/kernel_stats.err; head -7 $O/kernel_stats.csv
B="bench.py --config $CFG --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-profile $@"
pass() { n=$1; shift; rocprofv3 --pmc "$@" -d $O/pmc_$n -o runc -- python $B > $O/pmc_$n.json 2> $O/pmc_$n.log; echo "cfg$CFG pmc $n rc=$?"; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
python tools/pmc_report.py $O/pmc_sq1 $O/pmc_fetch $O/pmc_write > $O/pmc_counters.json 2> $O/pmc_report.err; echo "report rc=$?"
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write --config $CFG "$@" > $O/pmc_traffic.json 2> $O/pmc_traffic.err; echo "traffic rc=$?"; head -c 600 $O/pmc_traffic.json
rm -rf $O/stats $O/pmc_sq1 $O/pmc_fetch $O/pmc_write   # the raw databases stay on the box: the summaries above are what is kept
du -sh $O | cut -f1
