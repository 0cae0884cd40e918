#!/bin/bash
# Same-box A/B of two conv_s4 shape tables: builds libpfhip_b.so with table B, alternates the headline bench N times.
#   tools/ab_tables.sh <table_b.inc> [N]
set -u
tb=$1; n=${2:-3}
d=panoptic-forecasting_amd/csrc
cp $d/conv_s4_tuned.inc /tmp/table_a.inc
cp $tb $d/conv_s4_tuned.inc
(cd $d && make -j16 2>&1 | grep -E "error" ; cp libpfhip.so /tmp/libpfhip_b.so)
cp /tmp/table_a.inc $d/conv_s4_tuned.inc
(cd $d && make -j16 2>&1 | grep -E "error")
for i in $(seq $n); do
  for v in a b; do
    if [ $v = a ]; then unset PF_LIBPFHIP; else export PF_LIBPFHIP=/tmp/libpfhip_b.so; fi
    python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --verbose --profile-steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), 'convs', round(d['roofline']['step']['stages']['convs']['ms'],3), 'sum', round(d['roofline']['kernel_ms_per_step'],3))"
  done
done
