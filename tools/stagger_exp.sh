# bench.py schedule sweep (runs on the GPU box): frames/s for (batch, streams, stagger)
for i in 1 2; do
for cfg in "32 2 0" "32 2 1" "32 4 1" "48 3 1" "64 4 1" "64 4 0" "48 3 0"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --no-legs --profile-steps 1 --batch $1 --streams $2 --stagger $3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $1 streams $2 stagger $3: %.1f fps  %.3f ms/step'%(d['value'], d['ms_per_step']))"
done; done
