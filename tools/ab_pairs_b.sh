#!/bin/bash
# per-layer timings with and without conv_pair at the batch sizes given: tools/ab_pairs_b.sh <out-dir> <B> [<B> ...]
OUT=$1; shift; mkdir -p $OUT
for b in "$@"; do for o in 0 2; do PF_OPTS=fuse_pairs=$o python tools/layer_profile.py --batch $b --steps 5 > $OUT/layers_b${b}_fuse$o.txt 2>&1; done; done
