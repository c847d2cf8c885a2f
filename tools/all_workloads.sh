# usage: TAG=r02c bash tools/all_workloads.sh  -- one bench line per workload + the batch curve, under gpurun_out/$TAG/
TAG=${TAG:-r02}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for w in kms2party cggi cggi_l2 lmss kms2partyblock kms4party ccs2party ccs8party ccs8_n2048; do
  python3 bench.py --steps 3 --warmup 1 --workload $w --no-roofline > $O/bench_$w.json 2>/dev/null
done
python3 bench.py --steps 10 --warmup 2 --workload cggi_l2 --batch 1 --no-roofline --no-cpu-baseline > $O/bench_cggi_l2_b1.json 2>/dev/null
python3 bench.py --steps 10 --warmup 2 --workload cggi --batch 1 --no-roofline --no-cpu-baseline > $O/bench_cggi_b1.json 2>/dev/null
python3 bench.py --steps 10 --warmup 2 --workload kms2party --batch 1 --no-roofline --no-cpu-baseline > $O/bench_kms2party_b1.json 2>/dev/null
python3 bench.py --steps 2 --warmup 1 --workload lmss --batch 16384 --no-roofline --no-cpu-baseline > $O/bench_lmss_16384.json 2>/dev/null
python3 bench.py --steps 2 --warmup 1 --workload lmss_k2 --batch 16384 --no-roofline --no-cpu-baseline > $O/bench_lmss_k2_16384.json 2>/dev/null
for w in kms2_n1024 kms2party kms2partyblock cggi; do python3 bench.py --steps 3 --warmup 1 --workload $w --arith exact --no-roofline --no-cpu-baseline --no-secondary > $O/bench_${w}_exact.json 2>/dev/null; done
: > $O/batch_curve.txt
for w in kms2_n1024 cggi; do for b in 1 16 64 256 512 1024 2048 4096 16384; do
  python3 bench.py --steps 3 --warmup 1 --workload $w --batch $b --no-roofline --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w', 'batch', $b, 'gates/s %.0f'%d['value'], 'ms/batch %.3f'%d['ms_per_step'], 'errs', d['decrypt_errors'])
" >> $O/batch_curve.txt
done; done
grep -h '"metric"' $O/bench_*.json | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config']['params'], d['config']['batch_per_gpu'], 'gates/s %.0f'%d['value'], 'rot %.2f'%d['kernels_ms_per_step']['blindrotate'], 'ks %.2f'%d['kernels_ms_per_step']['keyswitch'], 'errs', d['decrypt_errors'], 'bitexact', d.get('oracle_bitexact'), 'frac %.3f'%d['roofline']['frac'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))
"
cat $O/batch_curve.txt
