cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in front_ab_before front_ab; do
  rm -rf /tmp/pmc_$v
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d /tmp/pmc_$v -o x -- $R/tools/ubench/$v 8192 > /tmp/pmc_$v.log 2>&1
  echo "== $v: $(grep 'ms per launch' /tmp/pmc_$v.log)"
  python3 - /tmp/pmc_$v <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if 'bc_front_b_kernel' in r['Kernel_Name']:
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print({k: f"{v[0] / v[1]:.3e}" for k, v in acc.items()})
PY
done
