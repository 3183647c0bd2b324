#!/bin/bash
# GPU session A (round 2): full GPU suite, default bench, launcher rehearsal (2 gloo ranks on the one GPU),
# configs 3/4/5 bench lines.  Outputs under gpurun_out/r02a/.
set -o pipefail
O=gpurun_out/r02a
mkdir -p $O
python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -25 $O/pytest.log
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench cfg2 rc=$?"
DSAT_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 8 --warmup 2 --repeats 3 --no-cpu-baseline --no-profile > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "bench gloo2 rc=$?"
python bench.py --config 3 --repeats 3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench cfg3 rc=$?"
python bench.py --config 4 --repeats 3 --stream-instances 1024 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "bench cfg4 rc=$?"
python bench.py --config 5 --steps 16 --repeats 3 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "bench cfg5 rc=$?"
for f in cfg2 gloo2 cfg3 cfg4 cfg5; do echo "== $f"; head -c 600 $O/bench_$f.json; echo; tail -3 $O/bench_$f.err; done
