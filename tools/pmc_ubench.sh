#!/bin/bash
# LDS / VALU counters of ONE kernel of a micro-benchmark binary (a --pmc pass of its own: no trace domains beside it).
# usage (on the GPU box, through gpurun): bash tools/pmc_ubench.sh <kernel name substring> <binary> [args...]
#   e.g. bash tools/pmc_ubench.sh bc_front_b_kernel tools/ubench/front_ab 8192
# prints the per-launch averages of SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE, SQ_INSTS_LDS, SQ_INSTS_VALU
K=$1; shift
BIN=$(realpath $1); shift
cd /tmp && export TMPDIR=/tmp
D=/tmp/pmc_ubench_$$
rm -rf $D
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $D -o x -- $BIN "$@" > $D.log 2>&1
grep -E "ms per launch" $D.log | tail -2
python3 - $D "$K" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r['Kernel_Name']:
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print({k: f"{v[0] / v[1]:.3e}" for k, v in acc.items()})
PY
