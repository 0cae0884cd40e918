#!/bin/bash
# Re-measure the conv_s4 shape table on the GPU box: tools/tune_s4.py at B = 16, 8, 4, 2, 1 with the CURRENT table as "auto",
# winners into gpurun_out/<tag>_conv_s4_tuned.inc (not installed: compare with tools/ab_tables.sh, then copy it over csrc/conv_s4_tuned.inc).
#   tools/retune_s4.sh <tag>
set -u
tag=${1:-retune}
out=gpurun_out
mkdir -p $out
inc=panoptic-forecasting_amd/csrc/conv_s4_tuned.inc
new=$out/${tag}_conv_s4_tuned.inc
echo "    // rows appended by tools/tune_s4.py --emit (MI355X, 1024x2048 network, dense-tap conv_s4; p1: 0 = 8x32 tiles, 1 = 8x64, 2 = 16x32 on 8 waves, 3 / 4 = 8x32 with the rounds split over 2 / 4 wave groups)" > $new
for b in 16 8 4 2 1; do
  python tools/tune_s4.py --batch $b --emit $new > $out/${tag}_tune_s4_b$b.txt 2>&1 || echo "tune b=$b failed"
done
echo "new table: $new"
