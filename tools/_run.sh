for cfg in "4 3 1 0 0" "4 3 1 0 1" "4 3 1 0 2" "4 3 1 0 3" "4 3 0 0 0" "4 3 0 0 1" "4 3 0 0 2"; do set -- $cfg
NWW_X3_WAVES=$1 NWW_TRUNK_STRIPS=$2 NWW_X3_V1=$3 NWW_X3_SKEW=$4 NWW_TRUNK_DBG=$5 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('waves=$1 strips=$2 v1=$3 skew=$4 dbg=$5', d['kernel_ms']['trunk_x3:conv1+pool+conv2+pool'])"; done
