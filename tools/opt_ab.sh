#!/bin/bash
# Same-library A/B of a process-wide option (runs ON the GPU box): tools/opt_ab.sh <runs> "<PF_OPTS a>" "<PF_OPTS b>" [bench args]
RUNS=${1:-3}; A=$2; B=$3; shift 3
for i in $(seq 1 $RUNS); do
  for O in "$A" "$B"; do
    PF_OPTS="$O" PF_BENCH_KERNELS=1 python bench.py --batch 16 --streams 1 --steps 10 --warmup 3 --replays 1 --no-cpu-baseline --no-legs --verbose --profile-steps 3 "$@" > /tmp/ab_line.json 2> /tmp/ab_err.txt
    python - "$O" <<'PY'
import json, sys
d = json.load(open('/tmp/ab_line.json'))
st = d['roofline']['step']['stages']
print('%-24s %8.1f fps  kernel_ms %.3f  ' % (sys.argv[1] or '(default)', d['value'], d['roofline']['kernel_ms_per_step']) + '  '.join('%s %.3f' % (k, v['ms']) for k, v in sorted(st.items())))
PY
    grep "^# " /tmp/ab_err.txt | grep -E "front|stem|split_kernel<2, 64>|conv_dma_kernel<3, 2" | sed "s/^/    /"
  done
done
