mkdir -p gpurun_out/${TAG:-r01b}
for w in lmss kms2partyblock ccs2party kms4party ccs8party; do
  B=1024; S=3; [ $w = ccs8party ] && S=1; [ $w = kms4party ] && S=2
  python bench.py --steps $S --warmup 1 --workload $w --batch $B --no-roofline 2>/dev/null | grep '"metric"' > gpurun_out/${TAG:-r01b}/bench_$w.json
  python3 -c "
import json; d=json.load(open('gpurun_out/${TAG:-r01b}/bench_$w.json')); print('$w', 'gates/s %.0f'%d['value'], d['kernels_ms_per_step'], 'cpu %.0f'%d['cpu_baseline']['value'], 'bitexact', d['oracle_bitexact'], 'dec', d['decrypt_ok'])"
done
