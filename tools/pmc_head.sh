#!/bin/bash
# PMC counters per kernel for one head: bash tools/pmc_head.sh conformer 2048
cd /tmp && export TMPDIR=/tmp
H=${1:-conformer}; B=${2:-2048}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$H
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq -o p -- python $GRAFT_REPO_ROOT/tools/profile_head.py $H $B > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -o p -- python $GRAFT_REPO_ROOT/tools/profile_head.py $H $B > $OUT/st.log 2>&1
python - <<PY
import csv, statistics
from collections import defaultdict
per = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open("$OUT/sq/p_counter_collection.csv")):
    per[(row["Kernel_Name"][:40], row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(per.items(), key=lambda kv: -max(kv[1].get("SQ_BUSY_CYCLES", [0]))):
    print(k, {n: f"{statistics.median(x):.3g}" for n, x in v.items()})
PY
head -12 $OUT/st/p_kernel_stats.csv
