mkdir -p gpurun_out/s7p
for i in 1 2; do
for v in default fr0 fr2 fr4 fr6; do
  case $v in default) A="";; fr0) A="--free-run 0";; fr2) A="--free-run 2";; fr4) A="--free-run 4";; fr6) A="--free-run 6";; esac
  python bench.py --no-cpu-baseline --no-legs --profile-steps 1 $A > gpurun_out/s7p/bench_${v}_$i.json 2> gpurun_out/s7p/bench_${v}_$i.err
done; done
python bench.py --no-cpu-baseline --no-legs --profile-steps 1 --batch 48 --streams 3 > gpurun_out/s7p/bench_b48_default.json 2>/dev/null
python bench.py --no-cpu-baseline --no-legs --profile-steps 1 --batch 48 --streams 3 --free-run 2.7 > gpurun_out/s7p/bench_b48_fr.json 2>/dev/null
python bench.py --no-cpu-baseline --no-legs --profile-steps 1 --batch 32 --streams 4 --free-run 1 > gpurun_out/s7p/bench_b32s4_fr.json 2>/dev/null
