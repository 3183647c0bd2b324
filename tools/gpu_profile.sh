#!/bin/bash
# Profiling session (config 2): rocprofv3 kernel stats of a bench run, PMC passes (each its own run, program
# directly after `--`), in-kernel phase stamps.  Summaries land in gpurun_out/$TAG/ and are copied to profiles/.
set -o pipefail
TAG=${1:-r02prof}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
B="bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-profile"
rocprofv3 -L > $O/counters_available.txt 2>&1
grep -c . $O/counters_available.txt
rocprofv3 --kernel-trace --stats -d $O/stats -o run -- python bench.py --steps 16 --warmup 2 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/stats.log; echo "stats rc=$?"
pass() { n=$1; shift; rocprofv3 --pmc "$@" -d $O/pmc_$n -o runc -- python $B > $O/pmc_$n.json 2> $O/pmc_$n.log; echo "pmc $n rc=$?"; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_F32
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
python tools/pmc_report.py $O/pmc_* > $O/pmc_counters.json 2> $O/pmc_report.err; echo "report rc=$?"
python tools/stamps.py --config 2 --out $O/stamps_cfg2 > $O/stamps_cfg2.stdout 2>&1; echo "stamps rc=$?"
cat $O/stamps_cfg2.stdout | tail -25
find $O/stats -name "*kernel_stats.csv" | head -2
python tools/pmc_report.py --stats $O/stats/run_results.db | head -6
python - <<PY
import json
d=json.load(open("$O/pmc_counters.json"))
for k,v in d["kernels"].items():
    print(k, json.dumps(v.get("derived")), v.get("registers"))
PY
tail -3 $O/pmc_sq1.log $O/pmc_sq2.log $O/pmc_tcc.log | cut -c1-300
