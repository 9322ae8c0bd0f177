for rep in 1 2; do for n0 in 0 130 132 134 136; do
NWW_X3_N0=$n0 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('n0=$n0', d['ms_per_step'], d['kernel_ms']['trunk_x3:conv1+pool+conv2+pool'])"; done; done
