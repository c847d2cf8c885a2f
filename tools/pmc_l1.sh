# usage: WLS="lmss cggi" bash tools/pmc_l1.sh -- vector L1 (TCP) / L2 (TCC) request counters of the rotation kernel of a workload
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TCP_[A-Z_]*(sum)?|TA_[A-Z_]*sum|TCC_(REQ|READ|HIT|MISS|EA_RDREQ)[A-Z_0-9]*sum)\b" | sort -u | tr '\n' ' ' | cut -c1-1500; echo
for WL in ${WLS:-lmss cggi}; do
D=$R/gpurun_out/pmcl1_$WL; rm -rf $D; mkdir -p $D
ARGS="--workload $WL --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline"
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $D/a -- python3 $R/bench.py $ARGS > $D/log_a.txt 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_BUSY_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $D/b -- python3 $R/bench.py $ARGS > $D/log_b.txt 2>&1
python3 - "$D" "$WL" <<'PY'
import csv, glob, sys
d, wl = sys.argv[1], sys.argv[2]
agg = {}
for f in glob.glob(d + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'blindrotate' not in r['Kernel_Name']: continue
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
print(wl, {c: '%.4g' % (sum(v) / len(v)) for c, v in sorted(agg.items())})
PY
tail -1 $D/log_a.txt | cut -c1-200
done
