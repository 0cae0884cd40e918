#!/bin/bash
# headline bench under several (resident frames, streams) schedules on one box
for cfg in "64 4" "128 4" "96 3" "64 2" "80 5" "96 6" "64 4"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --profile-steps 1 --batch $1 --streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch $1 streams $2:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms/step')"
done
