cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for cob in 16 8; do echo "== COB=$cob"; NWW_CONV_COB=$cob python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernel_ms'])"; done
for cfg in "16 256 3" "16 128 3" "16 128 6" "8 128 8" "26 256 2" "32 256 1" "16 256 6"; do set -- $cfg; echo "== FE fc=$1 block=$2 wgs/cu=$3"; NWW_FE_FC=$1 NWW_FE_BLOCK=$2 NWW_FE_WGS_PER_CU=$3 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernel_ms']['frontend:fe_stft_mel_db_kernel'], d['stft_stage'])"; done
