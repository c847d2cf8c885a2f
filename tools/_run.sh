mkdir -p gpurun_out/r03b
timeout 3000 python3 tools/noise_measure.py --batch 256 CGGIparam CGGI_N1024_l2 Blockparam Blockparam_k2 CCS2party CCS4party CCS8party CCS16party CCS8party_N2048 KMS2party KMS2party_N1024_l2 KMS4party KMS8party KMS2partyblock > gpurun_out/r03b/noise_all.jsonl 2> gpurun_out/r03b/noise_all.err
timeout 1200 python3 tools/noise_measure.py --batch 256 --variants Blockparam CCS8party KMS4party KMS2partyblock > gpurun_out/r03b/noise_var.jsonl 2>> gpurun_out/r03b/noise_all.err
tail -2 gpurun_out/r03b/noise_all.err
cat gpurun_out/r03b/noise_all.jsonl gpurun_out/r03b/noise_var.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['set'], d['variant'], 'sigma %.5f' % d['sigma'], 'mean %.5f' % d['mean'], 'max %.4f' % d['max'], 'fails', d['fails'], '/', d['gates'], [round(x,4) for x in d['sigma_after_parties']])
"
