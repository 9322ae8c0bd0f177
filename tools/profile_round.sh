#!/bin/bash
# Round profile on the GPU box: parity tests, bench, rocprofv3 kernel stats, and separate PMC passes
# (FETCH_SIZE / WRITE_SIZE cannot share a pass; no --stats/trace combined with --pmc).
# usage: bash tools/profile_round.sh r03   (run through gpurun; outputs under gpurun_out/<tag>/)
set -u
R=${1:-r03}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $OUT/pytest_gpu.txt
python bench.py 2>/dev/null | tail -1 | tee $OUT/bench.json
python tools/tolerance_audit.py > $OUT/tolerance_audit.json 2>$OUT/tolerance_audit.err
python tools/bench_configs.py 2>/dev/null > $OUT/configs.jsonl
python tools/latency_b1.py > $OUT/latency_b1.txt 2>&1
python tools/latency_breakdown.py > $OUT/latency_breakdown.txt 2>&1
B="python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --prewarm-seconds 0.3 --profile-every 1 --no-cpu-baseline --no-extras"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $B > $OUT/stats.log 2>&1
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --prewarm-seconds 0 --no-cpu-baseline --no-extras"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $B > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o sq -- $B > $OUT/pmc_sq.log 2>&1
# every BASELINE config (tools/bench_configs.py) under the kernel trace: rocprofv3's own per-kernel averages beside the tool's events
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_configs -o configs -- python $GRAFT_REPO_ROOT/tools/bench_configs.py > $OUT/stats_configs.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/c4_breakdown.py 1024 > $OUT/c4_breakdown.json 2>/dev/null
NWW_STREAM_INC=0 python tools/c4_breakdown.py 1024 > $OUT/c4_breakdown_full_window.json 2>/dev/null
# HBM bytes and SQ counters per kernel of the non-headline configs (separate --pmc passes)
bash tools/traffic_configs.sh $R C3 C5 C4 e2e gru > $OUT/traffic_configs.txt 2>&1
cp gpurun_out/${R}_traffic_configs.json $OUT/traffic_configs.json
bash tools/pmc_configs.sh $R > $OUT/pmc_configs.txt 2>&1
cp gpurun_out/${R}_pmc_configs.csv $OUT/pmc_all_configs.csv
find $OUT -name "*.csv" | head -30
