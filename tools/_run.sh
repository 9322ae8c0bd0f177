python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for v in 1 0; do NWW_X3S=$v python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('X3S=$v', d['ms_per_step'], d['kernel_ms'])"; done
for v in 1 0; do NWW_X3S=$v python tools/bench_configs.py C5 gru 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'kernel_ms' in d:
        print('X3S=$v', d['config'], d['ms_per_step'], d['max_abs_dlogit_vs_oracle']); print('   ', [round(v, 3) for k, v in d['kernel_ms'].items() if k.startswith('gemm')])"; done
