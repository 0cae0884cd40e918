#!/bin/bash
# Runs ON the GPU box: SQ counters of one command (two passes of 7-8 counters), per-kernel averages printed.
#   tools/pmc_kernel.sh <tag> <command...>
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p1 -o p -- "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p -- "$@" > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'conv' in k or 'stem' in k or 'raster' in k or 'bin_' in k:
        print(k[:70])
        print('   ', {c: '%.3g' % (sum(v) / len(v)) for c, v in sorted(d.items())})
PY
