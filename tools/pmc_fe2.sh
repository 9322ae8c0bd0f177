#!/bin/bash
# PMC passes over the wave-private frontend kernel (fe2_wave_kernel): instruction mix, stall breakdown, LDS.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-fe2_pmc}
mkdir -p $OUT
run() {  # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$n.log 2>&1
  python - <<PY
import csv
from collections import defaultdict
per = defaultdict(list)
for row in csv.DictReader(open("$OUT/$n/p_counter_collection.csv")):
    if "fe2_wave" in row["Kernel_Name"] or row["Kernel_Name"].startswith("fe_stft"):
        per[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("$n", {k: max(v) for k, v in per.items()})
PY
}
run mix SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_BRANCH
run stall SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
