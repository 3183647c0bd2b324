#!/bin/bash
# A/B of environment switches on the config-2 bench, same box:  gpu_ab_env.sh <outdir> "<VAR=val ...>" ...
O=gpurun_out/${1:-abenv}; shift
mkdir -p $O
i=0
run() { name=$1; shift; env "$@" python bench.py --repeats 3 --no-cpu-baseline --no-secondary > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/$name.json"))
    print("%-28s"%"$*", round(d["value"],1), d["repeats"]["ms_per_step_all"], {k:round(v["avg_launch_us"],1) for k,v in d["roofline_kernels"].items()})
except Exception as e:
    print("$name FAILED", e); print(open("$O/$name.err").read()[-600:])
PY
}
run base DSAT_NOP=1
for v in "$@"; do i=$((i+1)); run v$i $v; done
run base2 DSAT_NOP=1
