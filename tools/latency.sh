# single-gate / small-batch latency of every workload with the latency variant off (1) and automatic (0)
for w in ${WORKLOADS:-cggi cggi_l2 kms2_n1024 kms2party}; do for b in ${BATCHES:-1 16 64 256}; do for m in 1 0; do
 MKT_ROT_WIDE=$m python3 bench.py --workload $w --batch $b --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w batch $b wide=$m', 'ms/step %.3f'%d['ms_per_step'], 'rot %.3f'%d['kernels_ms_per_step']['blindrotate'], 'ks %.3f'%d['kernels_ms_per_step']['keyswitch'], 'p2 %.3f'%d['kernels_ms_per_step']['kms_phase2'], 'errs', d['decrypt_errors'])
"
done; done; done
