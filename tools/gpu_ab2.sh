# same-box A/B: checked-out builds in ab_<name>/ against the working tree; optional per-dir environment after ':'
one() { d=${1%%:*}; e=${1#*:}; [ "$e" = "$1" ] && e="A=1"; (cd $d && env $e python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', round(b['value'],2), {k:round(x['avg_launch_us'],1) for k,x in b['roofline_kernels'].items()})"); }
for i in 1 2 3 4; do for d in ${AB_DIRS:-ab_old .}; do one $d; done; done
