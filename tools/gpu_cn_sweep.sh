mkdir -p gpurun_out/r03f
for c in 3 5; do for f in 0 0.5 0.75 1.0; do
  if [ "$f" = "0" ]; then E="DSAT_CN_P3=0"; else E="DSAT_CN_P3_FRAC=$f"; fi
  env $E python bench.py --config $c --no-cpu-baseline --no-secondary > gpurun_out/r03f/bench_cfg${c}_$f.json 2>&1
done; done
python - <<PY
import json
for c in (3,5):
    for f in ("0","0.5","0.75","1.0"):
        t=open("gpurun_out/r03f/bench_cfg%d_%s.json"%(c,f)).read(); t=t[t.index('{"metric"'):]
        b=json.loads(t.splitlines()[0]); print(c, f, round(b["value"],2), {k:round(x["avg_launch_us"],1) for k,x in b.get("roofline_kernels",{}).items()})
PY
