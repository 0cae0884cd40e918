mkdir -p gpurun_out/s8e
D=$PWD/panoptic-forecasting_amd/csrc/ab
for i in 1 2; do
  for v in base v3 v4; do
    PF_LIBPFHIP=$D/libpfhip_$v.so python tools/bench_splat.py > gpurun_out/s8e/splat_${v}_$i.json 2>/dev/null
  done
  python tools/bench_splat.py > gpurun_out/s8e/splat_v2_$i.json 2>/dev/null
done
for f in gpurun_out/s8e/splat_*.json; do echo -n "$f "; cat $f; done
