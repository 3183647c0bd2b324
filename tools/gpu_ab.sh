#!/bin/bash
# A/B of environment switches on the config-2 bench: prints value and per-kernel times
O=gpurun_out/${1:-r02ab}
mkdir -p $O
run() { name=$1; shift; env "$@" python bench.py --repeats 3 --no-cpu-baseline > $O/$name.json 2> $O/$name.err; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("%-14s"%"$name", round(d["value"],1), {k:round(v["avg_launch_us"],1) for k,v in d["roofline_kernels"].items()}, d.get("throughput_mode",{}).get("value"))
PY
}
run base DSAT_X=0
run nopairb DSAT_NO_PAIRB=1
run nocn DSAT_NO_CN=1
run nopairb_nocn DSAT_NO_PAIRB=1 DSAT_NO_CN=1
