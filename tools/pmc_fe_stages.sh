#!/bin/bash
# VALU / LDS instruction counts of the frontend kernel per stage: PMC pass with each stage skipped (NWW_FE_DBG bits).
# Needs an ablation build of the library: NWW_HIPCC_FLAGS=-DNWW_ABLATION python -c "from nanowakeword_amd import build; build.build_hip(force=True, out='nanowakeword_amd/libnwwhip_abl.so')" and NWW_LIB_PATH pointing at it.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fe_stages
mkdir -p $OUT
for d in 0 1 2 4 8 15; do
  NWW_FE_DBG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/d$d -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/d$d.log 2>&1
  python - <<PY
import csv, statistics
from collections import defaultdict
per = defaultdict(list)
for row in csv.DictReader(open("$OUT/d$d/p_counter_collection.csv")):
    if row["Kernel_Name"].startswith("fe_stft"):
        per[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("dbg=$d", {k: max(v) for k, v in per.items()})
PY
done
