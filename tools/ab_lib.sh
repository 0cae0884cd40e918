#!/bin/bash
# A/B of alternative builds of libpfhip.so: tools/ab_lib.sh <out-dir> <lib> [<lib> ...]   ("default" = the in-tree build)
# per build: the packed-pair conv tests, per-layer timings at B = 32, two headline runs
OUT=$1; shift; mkdir -p $OUT
for L in "$@"; do
  if [ "$L" = default ]; then unset PF_LIBPFHIP; tag=default; else export PF_LIBPFHIP=$PWD/$L; tag=$(basename $L .so); fi
  timeout 900 python -m pytest tests/test_gpu_conv.py -q -x -k "s4 or packed or pair" > $OUT/tests_$tag.txt 2>&1; tail -1 $OUT/tests_$tag.txt
  timeout 600 python tools/layer_profile.py --batch 32 --steps 5 > $OUT/layers_b32_$tag.txt 2>&1
done
for rep in 1 2; do for L in "$@"; do
  if [ "$L" = default ]; then unset PF_LIBPFHIP; tag=default; else export PF_LIBPFHIP=$PWD/$L; tag=$(basename $L .so); fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'])" >> $OUT/headline.txt
done; done
cat $OUT/headline.txt
