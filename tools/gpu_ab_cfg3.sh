#!/bin/bash
# A/B of library builds on the config-3 bench (large batch) and config 2 two-in-flight
O=gpurun_out/${1:-r02ab3}; shift
mkdir -p $O
run() { name=$1; lib=$2; DSAT_LIB=$lib python bench.py --config 3 --steps 16 --repeats 3 --no-cpu-baseline > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/$name.json"))
    print("%-10s cfg3"%"$name", round(d["value"],2), {k:(round(v["avg_launch_us"],1),round(v["frac"],3)) for k,v in d["roofline_kernels"].items()})
except Exception as e:
    print("$name FAILED", e); print(open("$O/$name.err").read()[-800:])
PY
DSAT_LIB=$lib python bench.py --batches-in-flight 2 --steps 32 --repeats 3 --no-cpu-baseline --no-profile > $O/${name}_2f.json 2> $O/${name}_2f.err; python - <<PY
import json
try:
    d=json.load(open("$O/${name}_2f.json")); print("%-10s cfg2 x2 in flight"%"$name", round(d["value"],1))
except Exception as e:
    print("$name 2f FAILED", e); print(open("$O/${name}_2f.err").read()[-500:])
PY
}
run base "$PWD/diffusionsat_amd/libdsat.so"
for v in "$@"; do run $v "$PWD/tools/libs/libdsat_$v.so"; done
