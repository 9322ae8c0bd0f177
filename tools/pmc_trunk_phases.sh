#!/bin/bash
# PMC counters of the trunk kernel with conv2 (dbg=2) or conv1 (dbg=1) skipped
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trunk_phases
mkdir -p $OUT
for d in 0 1 2; do
  NWW_TRUNK_DBG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/d$d -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/d$d.log 2>&1
  python - <<PY
import csv
from collections import defaultdict
per = defaultdict(list)
for row in csv.DictReader(open("$OUT/d$d/p_counter_collection.csv")):
    if "trunk" in row["Kernel_Name"]:
        per[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("dbg=$d", {k: f"{max(v):.4g}" for k, v in per.items()})
PY
done
