#!/bin/bash
# A/B of environment switches on the config-4 stream (2048-instance sample) and on config 2's throughput_mode figure
O=gpurun_out/$1; mkdir -p $O
for spec in $2; do
  n=${spec%%=*}; e=${spec#*=}; [ "$e" = "-" ] && e=""
  env ${e//+/ } python bench.py --config 4 --stream-instances 2048 --repeats 5 --no-cpu-baseline --no-profile > $O/cfg4_$n.json 2>&1
  env ${e//+/ } python bench.py --no-cpu-baseline > $O/cfg2_$n.json 2>&1
done
python - "$O" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "cfg*.json"))):
    t = open(f).read()
    try:
        b = json.loads(t[t.index('{"metric"'):].splitlines()[0])
        print(os.path.basename(f), round(b["value"], 2), b["repeats"]["ms_per_step_all"], "tp", b.get("throughput_mode", {}).get("value"))
    except Exception as ex:
        print(os.path.basename(f), "FAILED", t[-300:])
PY
