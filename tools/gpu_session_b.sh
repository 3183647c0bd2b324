#!/bin/bash
# GPU session B: config-5 machinery (heavy rows, level-2 sums, select parts) -- tests + bench lines
set -o pipefail
O=gpurun_out/r02b
mkdir -p $O
python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_hot_kernels.py -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -20 $O/pytest.log
python bench.py --config 5 --steps 16 --repeats 3 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "bench cfg5 rc=$?"
python bench.py --config 5 --zipf 1.1 --steps 16 --repeats 3 --no-cpu-baseline > $O/bench_cfg5z.json 2> $O/bench_cfg5z.err; echo "bench cfg5 zipf1.1 rc=$?"
python bench.py --repeats 3 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench cfg2 rc=$?"
for f in cfg5 cfg5z cfg2; do echo "== $f"; head -c 400 $O/bench_$f.json; echo; tail -3 $O/bench_$f.err; done
