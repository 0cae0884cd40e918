#!/bin/bash
# Same-box A/B of several builds of libpfhip.so (runs ON the GPU box via gpurun): alternating runs of the one-sub-batch
# bench, per-stage kernel time (hipEvents) and frames/s of each.
#   tools/ab_bench.sh <runs> <libA.so> <libB.so> [<libC.so> ...] [-- bench args...]      (paths relative to the repo root)
# Output: gpurun_out/ab_<timestamp>.txt
RUNS=${1:-3}; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
OUT=gpurun_out/ab_$(date +%H%M%S).txt
mkdir -p gpurun_out
for i in $(seq 1 $RUNS); do
  for L in "${LIBS[@]}"; do
    PF_LIBPFHIP=$PWD/$L PF_BENCH_KERNELS=1 python bench.py --batch 16 --streams 1 --steps 10 --warmup 3 --replays 1 --no-cpu-baseline --no-legs --verbose --profile-steps 3 "$@" \
      > /tmp/ab_line.json 2> /tmp/ab_err.txt
    python - "$L" >> $OUT <<'PY'
import json, sys
d = json.load(open('/tmp/ab_line.json'))
st = d['roofline']['step']['stages']
print('%-50s %8.1f fps  kernel_ms %.3f  ' % (sys.argv[1][-50:], d['value'], d['roofline']['kernel_ms_per_step']) +
      '  '.join('%s %.3f' % (k, v['ms']) for k, v in sorted(st.items())))
PY
    grep "^# " /tmp/ab_err.txt | head -14 | sed "s/^/    /" >> $OUT
  done
done
cat $OUT | grep -v "^    "
