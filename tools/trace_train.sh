#!/bin/bash
# kernel-trace statistics of a training step, one stream (each kernel alone on the chip): tools/trace_train.sh <out-dir> [bench_train args]
OUT=$PWD/${1:-gpurun_out/trace_train}; shift || true
REPO=$PWD; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $REPO/tools/bench_train.py --steps 5 --no-side-stream "$@" > $OUT/bench.log 2>$OUT/trace.err
find $OUT -name '*.db' -delete
python - <<P
import csv, glob
f = glob.glob('$OUT/trace/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:40]:
    print('%-90s %6s %12s %10s %6s' % (r['Name'][:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'][:10], r['Percentage'][:5]))
P
