for w in cggi kms2party kms2_n1024; do for B in 1 16 256; do
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --workload $w --batch $B 2>/dev/null | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$w B=$B ms/step %.3f gates/s %.0f'%(d['ms_per_step'], d['value']))"
done; done
