# gates/s against batch size (one GPU): python bench.py --batch B for the headline shape and CGGIparam
for w in kms2_n1024 cggi; do for b in 1 16 64 256 512 1024 2048 4096 16384; do
 python bench.py --workload $w --batch $b --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-secondary 2>/dev/null | grep '"metric"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$w batch %6d  %8.0f gates/s  %8.2f ms/step' % ($b, d['value'], d['ms_per_step']))"
done; done
