#!/bin/bash
# conv_pair MERGED A/B: tests, per-layer timings (fuse_pairs 0 / 3 / 4 at B = 32), headline with fuse_pairs = 1 vs 4
OUT=${1:-gpurun_out/merged}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py -q -x -k "conv_pair or packed_activation_block" > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
for o in 0 3 4; do PF_OPTS=fuse_pairs=$o timeout 600 python tools/layer_profile.py --batch 32 --steps 5 > $OUT/layers_b32_fuse$o.txt 2>&1; done
for o in 1 4 1 4; do PF_OPTS=fuse_pairs=$o timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse', $o, d['value'], d['ms_per_step'])" >> $OUT/headline.txt; done
cat $OUT/headline.txt
