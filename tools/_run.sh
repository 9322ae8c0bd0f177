for d in 0 1 2 3 4; do NWW_FFN_DBG=$d python tools/bench_configs.py C5 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('dbg=$d', [v for k, v in d['kernel_ms'].items() if 'ffn' in k])"; done
