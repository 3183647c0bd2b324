#!/bin/bash
# rocprofv3 kernel stats of a short bench run: tools/gpu_kstats.sh TAG CONFIG [bench flags] -> gpurun_out/TAG/cfgN_kernel_stats.csv
TAG=$1; CFG=$2; shift; shift
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/st$CFG -o run -- python bench.py --config $CFG --steps 8 --warmup 2 --repeats 3 --no-cpu-baseline --no-secondary "$@" > $O/cfg${CFG}_bench.json 2> $O/cfg${CFG}_stats.log
python tools/pmc_report.py --stats $O/st$CFG/run_results.db > $O/cfg${CFG}_kernel_stats.csv 2> $O/cfg${CFG}_kernel_stats.err
rm -rf $O/st$CFG
head -9 $O/cfg${CFG}_kernel_stats.csv | cut -c1-150
