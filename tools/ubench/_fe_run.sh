cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "frontend or logmel or mel or fe_ or parity" 2>&1 | grep -E "passed|failed|error" | tail -3
python tools/pcm_rotation_probe.py 4096 2>&1 | grep buffer | head -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcfe
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d /tmp/pmcfe -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmcfe.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmcfe/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'fe2_wave' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: f"{max(v):.3e}" for k, v in acc.items()})
PY
