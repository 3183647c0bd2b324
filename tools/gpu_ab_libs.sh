#!/bin/bash
# A/B of library builds (DSAT_LIB=tools/libs/libdsat_<name>.so) on the config-2 bench
O=gpurun_out/${1:-r02abl}; shift
mkdir -p $O
run() { name=$1; lib=$2; DSAT_LIB=$lib python bench.py --repeats 3 --no-cpu-baseline > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/$name.json"))
    print("%-14s"%"$name", round(d["value"],1), d["repeats"]["ms_per_step_all"], {k:round(v["avg_launch_us"],1) for k,v in d["roofline_kernels"].items()}, round(d.get("throughput_mode",{}).get("value",0),1))
except Exception as e:
    print("$name FAILED", e); print(open("$O/$name.err").read()[-600:])
PY
}
run base "$PWD/diffusionsat_amd/libdsat.so"
for v in "$@"; do run $v "$PWD/tools/libs/libdsat_$v.so"; done
