#!/bin/bash
# by_batch A/B of two shape tables (table b = $1)
tb=$1
d=panoptic-forecasting_amd/csrc
cp $d/conv_s4_tuned.inc /tmp/table_a.inc
cp $tb $d/conv_s4_tuned.inc
(cd $d && make -j16 2>&1 | grep -E "error" ; cp libpfhip.so /tmp/libpfhip_b.so)
cp /tmp/table_a.inc $d/conv_s4_tuned.inc
(cd $d && make -j16 2>&1 | grep -E "error")
for i in 1 2; do
  for v in a b; do
    if [ $v = a ]; then unset PF_LIBPFHIP; else export PF_LIBPFHIP=/tmp/libpfhip_b.so; fi
    python bench.py --verbose --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), {k: round(v['value'],1) for k,v in d['by_batch'].items()})"
  done
done
