for i in 1 2; do
for lib in "" /root/repo/ab_old/libnwwhip_old.so; do
NWW_LIB_PATH=$lib python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('lib=${lib:-HEAD}', d['ms_per_step'], d['kernel_ms'])"
done; done
