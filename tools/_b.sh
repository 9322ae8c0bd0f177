#!/bin/bash
# usage: tools/_b.sh <conv_arith> -> prints value, ms, kernel_ms
python bench.py --conv-arith $1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['kernel_ms'])"
