# usage (GPU box): GS="1 2 4" WL=lmss [BATCH=1024] [XARGS="--arith exact"] [KERN=exact_kms_phase1] bash tools/pmc_blk.sh  -- SQ / TCP counters of the rotation kernel under each grouping (MKT_ROT_BLKG)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for g in ${GS:-1 2 4}; do
 export MKT_ROT_BLKG=$g
 D=$R/gpurun_out/pmcblk_${WL:-lmss}_$g; rm -rf $D; mkdir -p $D
 ARGS="--workload ${WL:-lmss} --batch ${BATCH:-1024} --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline ${XARGS:-}"
 timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $D/a -- python3 $R/bench.py $ARGS > /dev/null 2>&1
 timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $D/b -- python3 $R/bench.py $ARGS > /dev/null 2>&1
 timeout 150 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $D/c -- python3 $R/bench.py $ARGS > /dev/null 2>&1
 KERN=${KERN:-} python3 - "$D" "$g" <<'PY'
import csv, glob, os, sys
d, g = sys.argv[1], sys.argv[2]
kern = os.environ.get('KERN', '')
agg = {}
for f in glob.glob(d + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if (kern not in k) if kern else ('blindrotate' not in k and 'ccs_' not in k): continue
        agg.setdefault(r['Counter_Name'], []).append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r.get('VGPR_Count', '?'), r.get('LDS_Block_Size', '?')))
o = {c: sum(x[0] for x in v) / len(v) for c, v in agg.items()}
ms = {c: sum(x[1] for x in v) / len(v) for c, v in agg.items()}
wc = o.get('SQ_WAVE_CYCLES', 1)
G = lambda k: o.get(k, 0)
print('G=%s' % g, 'ms %.2f' % ms.get('SQ_WAVE_CYCLES', 0), 'vgpr/lds', agg.get('SQ_WAVE_CYCLES', [(0, 0, '?', '?')])[0][2:],
      'valu/wave %.3f' % (G('SQ_ACTIVE_INST_VALU') / wc), 'lds/wave %.3f' % (G('SQ_ACTIVE_INST_LDS') / wc), 'wait_any %.3f' % (G('SQ_WAIT_ANY') / wc),
      'wait_inst %.3f' % (G('SQ_WAIT_INST_ANY') / wc), 'insts_valu %.4g' % G('SQ_INSTS_VALU'), 'busy %.4g' % G('SQ_BUSY_CYCLES'), 'waves %.4g' % G('SQ_WAVES'),
      '| clock GHz %.3f' % (G('GRBM_GUI_ACTIVE') / 8 / (ms.get('GRBM_GUI_ACTIVE', 1) * 1e-3) / 1e9), 'bank_conf/lds_idx %.3f' % (G('SQ_LDS_BANK_CONFLICT') / max(G('SQ_LDS_IDX_ACTIVE'), 1)),
      'wait_lds/wave %.3f' % (G('SQ_WAIT_INST_LDS') / wc),
      '| L1 acc %.4g' % G('TCP_TOTAL_CACHE_ACCESSES_sum'), 'L1->L2 rd %.4g' % G('TCP_TCC_READ_REQ_sum'), 'L1 miss %.3f' % (G('TCP_TCC_READ_REQ_sum') / max(G('TCP_TOTAL_CACHE_ACCESSES_sum'), 1)),
      'tcp_pending_stall %.4g' % G('TCP_PENDING_STALL_CYCLES_sum'),
      '| L2 hit %.3f' % (G('TCC_HIT_sum') / max(G('TCC_HIT_sum') + G('TCC_MISS_sum'), 1)), 'L2 req %.4g' % G('TCC_REQ_sum'), 'fetch GB %.2f' % (G('FETCH_SIZE') * 2 * 1024 / 1e9))
PY
done
