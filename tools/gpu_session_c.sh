#!/bin/bash
# GPU session C: full GPU suite + config-2 bench + phase stamps (after a kernel change)
set -o pipefail
O=gpurun_out/${1:-r02c}
mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -6 $O/pytest.log
python bench.py --repeats 3 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench cfg2 rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_cfg2.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["repeats"]["ms_per_step_all"])
print({k:(round(v["frac"],3),round(v["avg_launch_us"],1)) for k,v in d["roofline_kernels"].items()})
print(d.get("throughput_mode"))
PY
python tools/stamps.py --config 2 --out $O/stamps_cfg2 2>&1 | tail -22
