#!/bin/bash
# A/B of environment switches on bench.py: tools/gpu_ab_env.sh OUTDIR "CONFIGS" "NAME=ENV ..." (ENV: K=V joined by +, or "-")
O=gpurun_out/$1; mkdir -p $O
for c in $2; do for spec in $3; do
  n=${spec%%=*}; e=${spec#*=}; [ "$e" = "-" ] && e=""
  env ${e//+/ } python bench.py --config $c --no-cpu-baseline --no-secondary > $O/cfg${c}_$n.json 2>&1
done; done
python - "$O" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "cfg*.json"))):
    t = open(f).read()
    try:
        b = json.loads(t[t.index('{"metric"'):].splitlines()[0])
        print(os.path.basename(f), round(b["value"], 2), {k: round(x["avg_launch_us"], 1) for k, x in b.get("roofline_kernels", {}).items()})
    except Exception as ex:
        print(os.path.basename(f), "FAILED", t[-300:])
PY
