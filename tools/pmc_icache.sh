# usage (GPU box, repo root): bash tools/pmc_icache.sh <outdir-tag> [bench args...]  -- instruction-cache counters of one bench.py command
# (the EXACT gate kernels' main loops are 60-80 KB of code: do they run out of the instruction cache?)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES" "SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_ic_$i -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline "$@" > $O/pmc_ic_$i.log 2>&1
  tail -2 $O/pmc_ic_$i.log | cut -c1-300
done
python3 - <<PY
import csv, glob
agg = {}
for f in glob.glob('$O/pmc_ic_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void mktd::', '')[:48]
        if 'exact_' in k or 'blindrotate' in k or 'ccs' in k:
            agg.setdefault((k, r['Counter_Name']), []).append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()):
    print(k, c, '%.5g' % (sum(v) / len(v)), len(v))
PY
