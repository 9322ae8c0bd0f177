python -m pytest tests -m gpu -x -q -k "e2e or crnn or onnx" 2>&1 | tail -3
for d in 0 8; do NWW_C3_DBG=$d python tools/bench_configs.py e2e_dnn 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('dbg=$d', d['kernel_ms'])"; done
