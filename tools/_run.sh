python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k in ('value', 'ms_per_step', 'steps', 'kernel_ms', 'arith', 'sustained', 'h2d_inclusive'): print(k, d.get(k))"
