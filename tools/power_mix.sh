#!/bin/bash
# Power and clock of pure instruction streams (tools/ubench/power_mix.hip) sampled with rocm-smi: energy per wave-instruction.
cd $GRAFT_REPO_ROOT
echo -n "idle: "; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo
for mode in ${MODES:-0 3 1 2 4 5}; do
  tools/ubench/power_mix $mode 7 &
  sleep 3.5
  for i in 1 2 3; do echo -n "mode $mode: "; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/=*//g; s/GPU\[0\]//g' | tr '\n' ' '; echo; sleep 1; done
  wait
  sleep 2
done
