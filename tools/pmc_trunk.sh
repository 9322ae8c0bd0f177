#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_trunk
mkdir -p $OUT
for d in 0 1; do
NWW_TRUNK_DBG=$d rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/d$d -o sq -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/d$d.log 2>&1
done
python - <<'PY'
import csv, collections, glob, os
for d in (0,1):
    f=glob.glob(os.environ['GRAFT_REPO_ROOT']+f'/gpurun_out/pmc_trunk/d{d}/*counter_collection.csv')[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'cnn_trunk' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print("DBG",d,{k:round(sorted(v)[len(v)//2]/1e6,2) for k,v in agg.items()})
PY
