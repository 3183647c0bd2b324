#!/bin/bash
# rocprofv3 kernel stats of the default bench run alone (no secondary figures), into gpurun_out/$1/
set -o pipefail
O=gpurun_out/${1:-r02stats}
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o run -- python bench.py --steps 16 --warmup 2 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/stats.log; echo "stats rc=$?"
python tools/pmc_report.py --stats $O/stats/run_results.db > $O/kernel_stats.csv
head -8 $O/kernel_stats.csv
