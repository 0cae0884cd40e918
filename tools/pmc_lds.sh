#!/bin/bash
# LDS-pipe counters of the one-sub-batch bench command (runs ON the GPU box via gpurun): is the 3x3 packed-pair kernel bound by
# LDS bandwidth, and do its fragment reads conflict?   tools/pmc_lds.sh <tag>
TAG=${1:-lds}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --batch 32 --streams 1 --steps 4 --warmup 2 --replays 1 --no-cpu-baseline --no-legs --no-graph --verbose --profile-steps 1"
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_lds" -o p -- $BENCH > "$OUT/bench.log" 2>"$OUT/err.log"
rocprofv3 --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_lds2" -o p -- $BENCH > "$OUT/bench2.log" 2>"$OUT/err2.log"
find "$OUT" -name '*.db' -delete
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in ('pmc_lds', 'pmc_lds2'):
    for f in glob.glob(out + '/' + d + '/**/*counter_collection.csv', recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            if d == 'pmc_lds' and r['Counter_Name'] == 'SQ_INSTS_LDS': n[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_ACTIVE_INST_LDS', 0))[:14]:
    c = max(n[k], 1)
    print('%-60s n=%4d ' % (k, c) + '  '.join('%s %.3g' % (a.replace('SQ_', ''), b / c) for a, b in sorted(v.items())))
PY
