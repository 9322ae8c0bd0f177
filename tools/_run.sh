for v in 1 0; do NWW_TAIL_REDUCE=$v python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('tail_reduce=$v', d['ms_per_step'], d['kernel_ms'])"; done
python tools/latency_breakdown.py 2>&1 | grep "^cnn B=1" | cut -c1-330
NWW_TAIL_REDUCE=0 python tools/latency_breakdown.py 2>&1 | grep "^cnn B=1" | cut -c1-330
