#!/bin/bash
# SQ counters of every kernel of the non-headline BASELINE configs (one rocprofv3 --pmc pass over tools/bench_configs.py; no
# trace options beside --pmc) -> gpurun_out/<tag>_pmc_configs.csv: per kernel the median over its launches.
# usage (through gpurun): bash tools/pmc_configs.sh r03
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/${R}_cfg_pmc
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${R}_cfg_pmc -o sq -- python $GRAFT_REPO_ROOT/tools/bench_configs.py C3 C5 e2e gru C4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - $R <<'PY'
import csv, glob, collections, sys
R = sys.argv[1]
f = glob.glob(f'gpurun_out/{R}_cfg_pmc/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
names = ['GRBM_GUI_ACTIVE', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_LDS', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT']
rows = []
for k, d in acc.items():
    med = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    rows.append((med.get('GRBM_GUI_ACTIVE', 0), k, len(next(iter(d.values()))), med))
with open(f'gpurun_out/{R}_pmc_configs.csv', 'w') as o:
    w = csv.writer(o)
    w.writerow(['kernel', 'launches'] + names + ['mfma_busy_frac_of_simd_cycles', 'lds_conflict_frac'])
    for g, k, n, m in sorted(rows, reverse=True):
        if g < 20000: continue
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs (0.575 ms of bc_front_b reads 1.01e7); MFMA_BUSY over the 1024 SIMDs
        mf = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / (g / 8) if g else 0
        lf = m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE'] if m.get('SQ_LDS_IDX_ACTIVE') else 0
        w.writerow([k[:110], n] + ['%.4g' % m.get(c, 0) for c in names] + ['%.3f' % mf, '%.3f' % lf])
print(open(f'gpurun_out/{R}_pmc_configs.csv').read()[:3000])
PY
