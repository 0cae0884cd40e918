mkdir -p gpurun_out/s8f
D=$PWD/panoptic-forecasting_amd/csrc/ab
PF_LIBPFHIP=$D/libpfhip_w2.so python -m pytest tests/test_gpu_warp_splat.py -x -q > gpurun_out/s8f/splat_tests_w2.log 2>&1
PF_LIBPFHIP=$D/libpfhip_w1.so python -m pytest tests/test_gpu_warp_splat.py -x -q > gpurun_out/s8f/splat_tests_w1.log 2>&1
for i in 1 2 3; do
  for v in base w1 w2; do
    PF_LIBPFHIP=$D/libpfhip_$v.so python tools/bench_splat.py > gpurun_out/s8f/splat_${v}_$i.json 2>/dev/null
  done
done
tail -1 gpurun_out/s8f/splat_tests_w1.log gpurun_out/s8f/splat_tests_w2.log
for f in gpurun_out/s8f/splat_*.json; do echo -n "$f "; cat $f; done
