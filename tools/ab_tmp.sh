mkdir -p gpurun_out/s8c
python -m pytest tests/test_gpu_warp_splat.py -x -q > gpurun_out/s8c/splat_tests.log 2>&1
BASE=$PWD/panoptic-forecasting_amd/csrc/ab/libpfhip_base.so
for i in 1 2 3; do
  PF_LIBPFHIP=$BASE python tools/bench_splat.py > gpurun_out/s8c/splat_base_$i.json 2>/dev/null
  python tools/bench_splat.py > gpurun_out/s8c/splat_new_$i.json 2>/dev/null
done
tail -2 gpurun_out/s8c/splat_tests.log; cat gpurun_out/s8c/splat_*.json
