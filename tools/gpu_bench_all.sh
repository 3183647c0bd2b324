#!/bin/bash
# All bench lines of a round: config 2 (the default run, with CPU baseline), 3, 4, 5 and the launcher
# rehearsal (2 gloo ranks sharing the one GPU).  Lines land in gpurun_out/$1/ (copied to profiles/).
O=gpurun_out/${1:-r02bench}
mkdir -p $O
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "cfg2 rc=$?"
python bench.py --config 3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?"
python bench.py --config 4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"
python bench.py --config 5 --steps 32 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?"
python bench.py --config 5 --zipf 1.1 --steps 32 --no-cpu-baseline > $O/bench_cfg5_zipf11.json 2> $O/bench_cfg5_zipf11.err; echo "cfg5z rc=$?"
DSAT_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 16 --warmup 2 --repeats 3 --no-cpu-baseline --no-profile > $O/bench_launcher_gloo2.json 2> $O/bench_launcher_gloo2.err; echo "gloo2 rc=$?"
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f))
        print(f.split("/")[-1], round(d["value"],2), d["unit"], "n_gpus",d["n_gpus"], "roofline", d.get("roofline",{}).get("kernel"), round(d.get("roofline",{}).get("frac",0),3), {k:round(v["frac"],3) for k,v in d.get("roofline_kernels",{}).items()}, "cpu", d.get("cpu_baseline",{}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
