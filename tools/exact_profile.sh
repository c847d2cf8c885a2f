R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03x
( echo "# EXACT mode after the 30-bit-prime / lazy-butterfly rebuild (round 3): python3 bench.py --arith exact --workload W --steps 3 --warmup 1, batch 1024"
for w in kms2_n1024 kms2party kms2partyblock ccs2party ccs8party cggi lmss; do python3 bench.py --arith exact --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep -a "\"metric\"" | tee $R/gpurun_out/r03x/bench_${w}_exact.json | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config']['params'], 'gates/s %.0f'%d['value'], 'rot ms %.2f'%d['kernels_ms_per_step']['blindrotate'], 'ks ms %.2f'%d['kernels_ms_per_step']['keyswitch'], 'decrypt_ok', d['decrypt_ok'])
"; done
python3 tools/exact_rate.py 2>&1 | grep -a -E "EXACT|F64REF" ) > $R/gpurun_out/r03x/exact_bench.txt 2>&1
( echo "# transform legs of one bench.py run (roofline_transform): Float64 and EXACT, 4 GiB per launch"; bash tools/ntt_legs.sh 2>&1 | grep -a -E "kernel" ) > $R/gpurun_out/r03x/transform_legs.txt 2>&1
( echo "# SQ counters of the EXACT batched transforms, N=1024 then N=2048 (tools/pmc_ntt.sh)"; N=1024 bash tools/pmc_ntt.sh; N=2048 bash tools/pmc_ntt.sh ) > $R/gpurun_out/r03x/ntt_pmc.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03x/stats -- python3 $R/bench.py --arith exact --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
cat $R/gpurun_out/r03x/exact_bench.txt $R/gpurun_out/r03x/transform_legs.txt $R/gpurun_out/r03x/ntt_pmc.txt | cut -c1-400
