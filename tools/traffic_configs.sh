#!/bin/bash
# HBM bytes per launch of every kernel of the non-headline BASELINE configs: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they
# cannot share a pass; no trace options beside --pmc) over tools/bench_configs.py -> gpurun_out/<tag>_traffic_configs.json.
# FETCH_SIZE is doubled for gfx950 as MI355X_MICROARCH.md prescribes; WRITE_SIZE 1:1; both are KB counters.
# usage (through gpurun): bash tools/traffic_configs.sh r04 [configs...]
R=${1:-r04}; shift
CFG=${@:-C3 C5 C4 e2e gru}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/${R}_traffic_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${R}_traffic_$c -o t -- python $GRAFT_REPO_ROOT/tools/bench_configs.py $CFG > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - $R <<'PY'
import csv, glob, collections, json, sys
R = sys.argv[1]
def med(counter):
    f = glob.glob(f'gpurun_out/{R}_traffic_{counter}/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    # the big launches only (warm-up and the 8-clip checks are tiny): median of the upper half
    return {k: sorted(v)[len(v) // 2:][len(sorted(v)[len(v) // 2:]) // 2] for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}
fe, n = med('FETCH_SIZE')
wr, _ = med('WRITE_SIZE')
out = {}
for k in sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, 0) + wr.get(k, 0))):
    b = 2 * fe.get(k, 0) * 1024 + wr.get(k, 0) * 1024
    if b < 4e6: continue
    short = k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    out[short] = {"launches_sampled": n.get(k, 0), "fetch_size_kb": fe.get(k, 0), "write_size_kb": wr.get(k, 0), "hbm_bytes_per_launch": int(b)}
json.dump(out, open(f'gpurun_out/{R}_traffic_configs.json', 'w'), indent=1)
for k, v in out.items():
    print(f"{v['hbm_bytes_per_launch'] / 1e6:9.1f} MB  {k[:100]}")
PY
