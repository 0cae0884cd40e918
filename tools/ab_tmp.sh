mkdir -p gpurun_out/s7c
for i in 1 2; do
for v in base new; do
  if [ $v = base ]; then export PF_LIBPFHIP=$PWD/panoptic-forecasting_amd/csrc/libpfhip_base.so; else unset PF_LIBPFHIP; fi
  PF_BENCH_KERNELS=1 python bench.py --no-cpu-baseline --no-legs > gpurun_out/s7c/bench_${v}_$i.json 2> gpurun_out/s7c/bench_${v}_$i.err
done; done
