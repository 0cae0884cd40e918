#!/bin/bash
# Same-box A/B of several builds of libpfhip.so on the strict-fp32 leg (bench.py --fp32-mfma-only); runs ON the GPU box via gpurun.
#   tools/ab_fp32.sh <runs> <libA.so> <libB.so> [...]
RUNS=${1:-2}; shift
OUT=gpurun_out/abf_$(date +%H%M%S).txt
for i in $(seq 1 $RUNS); do
  for L in "$@"; do
    PF_LIBPFHIP=$PWD/$L python bench.py --fp32-mfma-only --steps 10 --warmup 3 --no-legs --no-cpu-baseline --verbose --profile-steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-44s' % '$L'[-44:], round(d['value'],1), 'fps  convs', round(d['roofline']['step']['stages']['convs']['ms'],3), 'kernel sum', round(d['roofline']['kernel_ms_per_step'],3))" >> $OUT
  done
done
cat $OUT
