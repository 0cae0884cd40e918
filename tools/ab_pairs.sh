#!/bin/bash
# Same-box A/B of conv_pair.hip (plan option fuse_pairs, process-wide through PF_OPTS); runs ON the GPU box via gpurun.
#   tools/ab_pairs.sh <out-dir> [runs]
OUT=${1:-gpurun_out/pairs}; RUNS=${2:-2}; mkdir -p $OUT
for b in 32 4 1; do
  for o in 0 2; do
    PF_OPTS=fuse_pairs=$o python tools/layer_profile.py --batch $b --steps 5 > $OUT/layers_b${b}_fuse$o.txt 2>&1
    head -1 $OUT/layers_b${b}_fuse$o.txt | sed "s/^/fuse_pairs=$o: /"
  done
done
for i in $(seq 1 $RUNS); do
  for o in 0 2; do
    PF_OPTS=fuse_pairs=$o python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --verbose --profile-steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline fuse_pairs=$o', round(d['value'],1), 'fps  convs', round(d['roofline']['step']['stages']['convs']['ms'],3), 'kernel sum', round(d['roofline']['kernel_ms_per_step'],3), 'overflow', d.get('range_overflow'))"
  done
done
