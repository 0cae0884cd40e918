#!/bin/bash
# headline bench under larger resident batches on one box (round 5)
for cfg in "64 4" "128 4" "192 4" "256 4" "128 8" "192 6" "128 2" "64 4"; do
  set -- $cfg
  python bench.py --steps 10 --warmup 3 --no-legs --no-cpu-baseline --profile-steps 1 --batch $1 --streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch $1 streams $2:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms/step')"
done
