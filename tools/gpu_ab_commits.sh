# same-box comparison of checked-out builds (ab_*/ directories: `git archive <commit> | tar -x -C ab_<commit>` + build) with the working tree
one() { (cd $1 && python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(b['value'],2), {k:round(x['avg_launch_us'],1) for k,x in b['roofline_kernels'].items()})"); }
for i in 1 2 3; do for d in ${AB_DIRS:-ab_old .}; do one $d; done; done
