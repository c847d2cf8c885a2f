# usage: WL=ccs2party bash tools/pmc_icache.sh -- instruction-cache / fetch counters of the rotation kernel of a workload
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQC_ICACHE[A-Z_]*|SQ_IFETCH[A-Z_]*|SQ_WAIT_IFETCH[A-Z_]*|SQ_INST_LEVEL[A-Z_]*|SQC_INST[A-Z_]*|SQ_INSTS_SALU|SQ_WAVE_DEP[A-Z_]*|SQ_ACTIVE_INST_MISC|SQ_ACTIVE_INST_SCA|SQ_INST_CYCLES_SALU|SQ_THREAD_CYCLES_VALU|SQ_VALU_MFMA_BUSY_CYCLES|SQ_BUSY_CU_CYCLES|SQ_INSTS_BRANCH|SQ_WAIT_INST_ANY)\b" | sort -u | tr '\n' ' '; echo
for WL in ${WLS:-ccs2party cggi}; do
D=$R/gpurun_out/pmcic_$WL; rm -rf $D; mkdir -p $D
ARGS="--workload $WL --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $D/a -- python3 $R/bench.py $ARGS > $D/log_a.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQC_ICACHE_MISSES_DUPLICATE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d $D/b -- python3 $R/bench.py $ARGS > $D/log_b.txt 2>&1
python3 - "$D" "$WL" <<'PY'
import csv, glob, sys
d, wl = sys.argv[1], sys.argv[2]
agg = {}
for f in glob.glob(d + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'blindrotate' not in k: continue
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
print(wl, {c: '%.4g' % (sum(v) / len(v)) for c, v in sorted(agg.items())})
PY
tail -2 $D/log_a.txt | cut -c1-300
done
