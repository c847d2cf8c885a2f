# usage (GPU box, repo root): bash tools/round_profiles.sh <tag> <part>   -- the round's committed measurements -> gpurun_out/<tag>/
#   part 1: the driver's command (default bench) with kernel-trace stats and every PMC group; the launcher / circuit / MUX / EXACT lines
#   part 2, 3: one bench line + FETCH_SIZE / WRITE_SIZE / clock passes per workload (so that roofline.traffic is never null)
#   part 4: PMC passes of the EXACT workloads and of the 16 384-gate block batch; part 5: every bench line again, once the round's PMC
#   summaries are committed under profiles/ (a line reads its traffic from there)
# then: python3 tools/collect_profiles.py <tag> <name>  for every name (tools/collect_all.sh <tag>)
TAG=${1:-r04}; PART=${2:-1}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
line() { python3 bench.py "$@" 2>/dev/null | grep '"metric"'; }
if [ $PART = 1 ]; then
  line --steps 20 --warmup 5 > $O/bench_kms2_n1024.json
  NOROOF= bash tools/pmc_pass.sh $TAG kms2_n1024 kms2_n1024 full > /dev/null    # NOROOF=: the transform legs stay in the profiled command
  line --arith exact --no-roofline --no-secondary --no-cpu-baseline > $O/bench_kms2_n1024_exact.json
  bash tools/pmc_pass.sh $TAG kms2_n1024 kms2_n1024_exact full -- --arith exact > /dev/null
  MKT_BENCH_SHARE_GPU=1 line --gpus 2 --launcher inproc --no-roofline --no-secondary > $O/bench_kms2_n1024_inproc2.json
  MKT_BENCH_SHARE_GPU=1 line --gpus 2 --no-roofline --steps 3 > $O/bench_kms2_n1024_ranks2.json
  line --workload adder8 --instances 1024 --steps 2 --no-roofline --no-secondary --no-cpu-baseline > $O/bench_adder8.json
  line --op mux --no-roofline --no-secondary > $O/bench_kms2_n1024_mux.json
fi
if [ $PART = 4 ]; then      # PMC passes of the other EXACT workloads (the headline's are in part 1)
  for w in kms2party kms2partyblock cggi lmss ccs2party; do bash tools/pmc_pass.sh $TAG $w ${w}_exact full -- --arith exact > /dev/null; done
  bash tools/pmc_pass.sh $TAG lmss lmss_16384 -- --batch 16384 > /dev/null
  bash tools/pmc_pass.sh $TAG kms2_n1024 kms2_n1024_mux -- --op mux > /dev/null
fi
if [ $PART = 5 ]; then      # bench lines only, AFTER the PMC summaries of this round are in profiles/ (roofline.traffic reads them)
  line --steps 20 --warmup 5 > $O/bench_kms2_n1024.json
  line --arith exact --no-roofline --no-secondary --no-cpu-baseline > $O/bench_kms2_n1024_exact.json
  MKT_BENCH_SHARE_GPU=1 line --gpus 2 --launcher inproc --no-roofline --no-secondary > $O/bench_kms2_n1024_inproc2.json
  MKT_BENCH_SHARE_GPU=1 line --gpus 2 --no-roofline --steps 3 > $O/bench_kms2_n1024_ranks2.json
  line --workload adder8 --instances 1024 --steps 2 --no-roofline --no-secondary --no-cpu-baseline > $O/bench_adder8.json
  line --op mux --no-roofline --no-secondary > $O/bench_kms2_n1024_mux.json
  for w in kms2party cggi cggi_l2 lmss kms2partyblock kms4party; do line --steps 8 --warmup 4 --workload $w --no-roofline > $O/bench_$w.json; done     # (warm-up: the part needs tens of ms of work to reach its sustained clock)
  for w in ccs2party ccs8party ccs8_n2048; do line --steps 2 --warmup 1 --workload $w --no-roofline > $O/bench_$w.json; done
  line --steps 2 --warmup 1 --workload lmss --batch 16384 --no-roofline --no-cpu-baseline > $O/bench_lmss_16384.json
  line --steps 2 --warmup 1 --workload lmss_k2 --batch 16384 --no-roofline --no-cpu-baseline > $O/bench_lmss_k2_16384.json
  line --steps 2 --warmup 1 --workload kms2partyblock --batch 16384 --no-roofline > $O/bench_kms2partyblock_16384.json
  line --steps 2 --warmup 1 --workload kms4party --batch 8192 --no-roofline --no-cpu-baseline > $O/bench_kms4party_8192.json                 # config (3): one GPU's share of 65 536 gates
  for w in kms2party kms2partyblock cggi lmss lmss_k2 ccs2party; do line --steps 6 --warmup 3 --workload $w --arith exact --no-roofline --no-cpu-baseline --no-secondary > $O/bench_${w}_exact.json; done
fi
if [ $PART = 2 ]; then
  for w in kms2party cggi cggi_l2 lmss kms2partyblock kms4party; do
    line --steps 8 --warmup 4 --workload $w --no-roofline > $O/bench_$w.json
    bash tools/pmc_pass.sh $TAG $w $w full > /dev/null          # full: the SQ groups too (issue_roofline needs SQ_INSTS_VALU)
  done
fi
if [ $PART = 3 ]; then
  for w in ccs2party ccs8party ccs8_n2048; do
    line --steps 2 --warmup 1 --workload $w --no-roofline > $O/bench_$w.json
    bash tools/pmc_pass.sh $TAG $w $w full > /dev/null
  done
  line --steps 2 --warmup 1 --workload lmss --batch 16384 --no-roofline --no-cpu-baseline > $O/bench_lmss_16384.json
  line --steps 2 --warmup 1 --workload lmss_k2 --batch 16384 --no-roofline --no-cpu-baseline > $O/bench_lmss_k2_16384.json
  bash tools/pmc_pass.sh $TAG lmss_k2 lmss_k2_16384 full -- --batch 16384 > /dev/null
  for w in kms2party kms2partyblock cggi lmss lmss_k2 ccs2party; do line --steps 6 --warmup 3 --workload $w --arith exact --no-roofline --no-cpu-baseline --no-secondary > $O/bench_${w}_exact.json; done
fi
ls $O | tr '\n' ' '
