#!/bin/bash
# Register / spill / LDS / scratch metadata of every kernel in one csrc file (device-only -S build).
# usage: tools/kernel_regs.sh trunk_b.hip [extra hipcc flags]
src=$1; shift
out=/tmp/isa/$(basename $src .hip).s
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S "$@" -I /root/repo/nanowakeword_amd/csrc -I /root/repo/include -x hip /root/repo/nanowakeword_amd/csrc/$src -o $out 2>/dev/null
python3 - $out <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    ag, name, scr, sg, vg, sp = m.groups()
    import subprocess
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(f"vgpr {vg:>4} agpr {ag:>4} spill {sp:>3} scratch {scr:>4}  {dem[:150]}")
PY
