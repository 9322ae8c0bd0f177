#!/bin/bash
# PMC passes (instruction mix, stall breakdown, LDS / MFMA busy) for every kernel whose name contains $1, over bench.py.
# usage (GPU box): [PMC_CMD='python tools/bench_configs.py C5'] bash tools/pmc_kernel.sh <kernel-substring> [outdir-tag]
K=$1; TAG=${2:-pmc_$1}; shift; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
run() {
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o p -- ${PMC_CMD:-python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-extras} > $OUT/$n.log 2>&1
  python - <<PY
import csv
from collections import defaultdict
per = defaultdict(list)
for row in csv.DictReader(open("$OUT/$n/p_counter_collection.csv")):
    if "$K" in row["Kernel_Name"]:
        per[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("$K $n", {k: (max(v), len(v)) for k, v in per.items()})
PY
}
run mix SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_BRANCH
run stall SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
