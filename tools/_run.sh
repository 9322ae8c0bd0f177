for kb in 160 80 100 56; do NWW_BC_FRONT_LDS_KB=$kb python tools/bench_configs.py C3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('lds_kb=$kb', d['ms_per_step'], [v for k, v in d['kernel_ms'].items() if 'conv1_dw' in k], d['max_abs_dlogit_vs_oracle'])"; done
