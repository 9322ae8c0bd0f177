#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_fc1
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/bench.py --conv-arith f16x3 --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-extras"
run() {
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o p -- $CMD > $OUT/$n.log 2>&1
  python - <<PY
import csv
from collections import defaultdict
per = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open("$OUT/$n/p_counter_collection.csv")):
    for k in ("gemm_x3", "cnn_trunk"):
        if k in row["Kernel_Name"]:
            per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in per.items():
    print(k, "$n", {c: (max(v), len(v)) for c, v in d.items()})
PY
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run stall SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run mix SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
