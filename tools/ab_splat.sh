#!/bin/bash
# Same-box A/B of builds of libpfhip.so on the warp/splat pair (tools/bench_splat.py, 32 frames): tools/ab_splat.sh <runs> <lib> [<lib> ...]
RUNS=$1; shift
for i in $(seq 1 $RUNS); do for L in "$@"; do
  PF_LIBPFHIP=$PWD/$L python tools/bench_splat.py --batch 32 --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-44s' % '$L'[-44:], {k.replace('pf::','').replace('(pf::SplatArgs)',''): v for k, v in d.items() if 'kernel' in k or k == 'pair_us'})"
done; done
