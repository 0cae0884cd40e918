#!/bin/bash
# Same-box A/B of several builds of libpfhip.so on the DEFAULT (headline) workload of bench.py (runs ON the GPU box via gpurun).
#   tools/ab_headline.sh <runs> <libA.so> <libB.so> [...]      (paths relative to the repo root)
RUNS=${1:-3}; shift
OUT=gpurun_out/abh_$(date +%H%M%S).txt
mkdir -p gpurun_out
for i in $(seq 1 $RUNS); do
  for L in "$@"; do
    PF_LIBPFHIP=$PWD/$L python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --verbose --profile-steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-44s' % '$L'[-44:], round(d['value'],1), 'fps  convs', round(d['roofline']['step']['stages']['convs']['ms'],3), 'kernel sum', round(d['roofline']['kernel_ms_per_step'],3), 'overflow', d.get('range_overflow'))" >> $OUT
  done
done
cat $OUT
