python -m pytest tests -m gpu -x -q -k "conformer" 2>&1 | tail -3
python tools/bench_configs.py C5 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['max_abs_dlogit_vs_oracle'], [v for k, v in d['kernel_ms'].items() if 'mha' in k])"
