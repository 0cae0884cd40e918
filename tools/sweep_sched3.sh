#!/bin/bash
# headline bench: passes per hipGraph and streams x sub-batch, with the B = 32 table rows (round 5; runs ON the GPU box)
for cfg in "128 4 2" "128 4 4" "128 4 1" "160 5 2" "96 3 2" "128 4 3" "256 8 2" "128 4 2"; do
  set -- $cfg
  python bench.py --steps 10 --warmup 3 --no-legs --no-cpu-baseline --profile-steps 1 --batch $1 --streams $2 --replays $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch $1 streams $2 passes $3:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms/step')"
done
