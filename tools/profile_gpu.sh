#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel trace + separate PMC passes of the bench command.
#   tools/profile_gpu.sh <tag> [bench args...]
# Outputs under gpurun_out/<tag>/ ; summarise locally with tools/profile_summarise.py <tag>.
set -u
TAG=${1:-prof}; shift || true
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# one sub-batch (32 frames) on one stream: the launches bench.py times for its roofline line; the headline runs 2 of
# these concurrently on 2 streams inside one hipGraph
BENCH="python $REPO/bench.py --batch 32 --streams 1 --steps 10 --warmup 3 --replays 1 --no-cpu-baseline --no-legs --no-graph --verbose --profile-steps 1 $*"
# another workload under the same passes (the training step: PF_PROFILE_CMD="python tools/bench_train.py --steps 5")
if [ -n "${PF_PROFILE_CMD:-}" ]; then BENCH="${PF_PROFILE_CMD/tools\//$REPO/tools/}"; fi
python -c "import sys; sys.path.insert(0, '$REPO'); import bench; print(bench.source_sha())" > "$OUT/source_sha.txt" 2>/dev/null
echo "$BENCH" | sed "s#$REPO/##g" > "$OUT/bench_cmd.txt"
cd /tmp
# 1. kernel trace + stats (timing)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $BENCH > "$OUT/bench_trace.log" 2>"$OUT/trace.err"
# 2. PMC passes, one counter set per run (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o p -- $BENCH > "$OUT/bench_pmc_$C.log" 2>"$OUT/pmc_$C.err"
  # calibration on a known byte count: a 256 MiB float4 device copy (reads 256 MiB, writes 256 MiB)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/cal_$C" -o c -- python $REPO/tools/pmc_calib.py > "$OUT/cal_$C.log" 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -o p -- $BENCH > "$OUT/bench_pmc_mfma.log" 2>"$OUT/pmc_mfma.err"
# issue/wait breakdown of the non-matrix kernels (quad-cycle units, MI355X_MICROARCH.md "rocprofv3 PMC slots")
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_sq" -o p -- $BENCH > "$OUT/bench_pmc_sq.log" 2>"$OUT/pmc_sq.err"
# keep only the CSVs (sizes are bounded by the 64 MiB pull limit)
find "$OUT" -name '*.db' -delete
ls -R "$OUT" | head -50
