#!/bin/bash
# tools/exp/ab_run.sh OUT CONFIGS... -- LIBS...: kernel_time.py for every (config, scratch library) pair; "product" = the shipped library
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
CFGS=(); while [ "$1" != "--" ]; do CFGS+=("$1"); shift; done; shift
: > $R/gpurun_out/$OUT
for c in "${CFGS[@]}"; do for l in "$@"; do
  if [ "$l" = product ]; then python3 $R/tools/exp/kernel_time.py $c >> $R/gpurun_out/$OUT 2>&1
  else python3 $R/tools/exp/kernel_time.py $c tools/exp/scratch/libmipgen_accel_$l.so >> $R/gpurun_out/$OUT 2>&1; fi
done; done
cat $R/gpurun_out/$OUT
