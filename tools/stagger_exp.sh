# bench.py schedule sweep (runs on the GPU box): frames/s for (batch, streams, stagger); optional argument: a list of configurations
CFGS=${1:-"32 2 0;32 2 1;32 4 1;48 3 1;64 4 1;64 4 0;48 3 0"}
for i in 1 2; do
echo "$CFGS" | tr ';' '\n' | while read b s g; do
  python bench.py --no-cpu-baseline --no-legs --profile-steps 1 --batch $b --streams $s --stagger $g 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b streams $s stagger $g: %.1f fps  %.3f ms/step'%(d['value'], d['ms_per_step']))"
done; done
