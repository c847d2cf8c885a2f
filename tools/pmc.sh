# usage (GPU box, repo root): [KERN=substr] [PMC_GROUPS="sq1 sq2 clk"] bash tools/pmc.sh <tag> -- <python program and args, e.g. bench.py --workload lmss --steps 2 --warmup 0 ...>
# Separate rocprofv3 --pmc passes (one counter group per run, only with --kernel-trace: the rule of MI355X_MICROARCH.md) of ONE python
# command, then per-kernel means with the ratios that get read off them.  Groups: sq1 (wave cycles, waits, active), sq2 (LDS, instruction
# counts), tcp (L1 <-> L2 requests), icache (instruction cache), clk (GRBM clock, L2 hit rate), fetch, write (fabric bytes, KiB).
# Replaces pmc_ntt.sh, pmc_ab.sh, pmc_blk.sh, pmc_l1.sh, pmc_blindrotate.sh, pmc_icache.sh, pmc_summary.py; alternative builds and switches:
# swap the library / export the MKT_* variable before the call (tools/sweep.sh shows how).  tools/pmc_pass.sh stays the per-round pass
# whose output tools/collect_profiles.py turns into profiles/<tag>_bench_<workload>_pmc.txt.
TAG=$1; shift; [ "$1" = -- ] && shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
declare -A G=(
 [sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
 [sq2]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS"
 [tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
 [icache]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"
 [clk]="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
 [fetch]="FETCH_SIZE" [write]="WRITE_SIZE")
for g in ${PMC_GROUPS:-sq1 sq2 clk}; do
  rm -rf $O/pmc_$g
  ( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc ${G[$g]} --output-format csv -d $O/pmc_$g -- python3 "$@" > $O/pmc_$g.log 2>&1 )
done
KERN=${KERN:-} python3 - "$O" <<'PY'
import csv, glob, os, re, sys
agg = {}
for f in glob.glob(sys.argv[1] + '/pmc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', ''))
        if 'mktd' not in k or os.environ['KERN'] not in k: continue
        agg.setdefault(k, {}).setdefault(r['Counter_Name'], []).append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r.get('VGPR_Count'), r.get('LDS_Block_Size')))
for k, cs in sorted(agg.items(), key=lambda kv: -max(sum(x[1] for x in v) for v in kv[1].values())):
    o, ms = {}, 0.0
    for c, v in cs.items():
        top = max(x[1] for x in v); v = [x for x in v if x[1] >= 0.85 * top]     # the full-work launches
        o[c] = sum(x[0] for x in v) / len(v); ms = max(ms, sum(x[1] for x in v) / len(v))
    wc = o.get('SQ_WAVE_CYCLES'); any1 = next(iter(cs.values()))[0]
    line = [k[:70], 'ms %.3f' % ms, 'vgpr %s lds %s' % (any1[2], any1[3])]
    if wc:
        line += ['%s/wave %.3f' % (n, o[c] / wc) for n, c in (('valu', 'SQ_ACTIVE_INST_VALU'), ('lds', 'SQ_ACTIVE_INST_LDS'), ('wait_any', 'SQ_WAIT_ANY'), ('wait_inst', 'SQ_WAIT_INST_ANY'), ('wait_lds', 'SQ_WAIT_INST_LDS')) if c in o]
    if 'GRBM_GUI_ACTIVE' in o: line.append('clock %.3f GHz' % (o['GRBM_GUI_ACTIVE'] / 8 / (ms * 1e-3) / 1e9))
    if 'TCC_HIT_sum' in o: line.append('L2 hit %.3f' % (o['TCC_HIT_sum'] / (o['TCC_HIT_sum'] + o.get('TCC_MISS_sum', 0))))
    if 'SQ_LDS_IDX_ACTIVE' in o: line.append('bank_conflict/lds_cycles %.3f' % (o.get('SQ_LDS_BANK_CONFLICT', 0) / max(o['SQ_LDS_IDX_ACTIVE'], 1)))
    if 'FETCH_SIZE' in o: line.append('fabric read %.3f GB (2 x FETCH_SIZE)' % (2 * o['FETCH_SIZE'] * 1024 / 1e9))
    if 'WRITE_SIZE' in o: line.append('written %.3f GB' % (o['WRITE_SIZE'] * 1024 / 1e9))
    print(' | '.join(line))
    print('    ' + '  '.join('%s %.5g' % (c, v) for c, v in sorted(o.items())))
PY
