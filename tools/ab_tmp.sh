mkdir -p gpurun_out/s7e
for i in 1 2; do
for v in base new; do
  if [ $v = base ]; then export PF_LIBPFHIP=$PWD/panoptic-forecasting_amd/csrc/libpfhip_base.so; else unset PF_LIBPFHIP; fi
  PF_BENCH_KERNELS=1 python bench.py --no-cpu-baseline --no-legs > gpurun_out/s7e/bench_${v}_$i.json 2> gpurun_out/s7e/bench_${v}_$i.err
done; done
unset PF_LIBPFHIP
python tools/tune_s4.py --batch 4 > gpurun_out/s7e/tune_b4.txt 2>&1
python tools/tune_s4.py --batch 1 > gpurun_out/s7e/tune_b1.txt 2>&1
