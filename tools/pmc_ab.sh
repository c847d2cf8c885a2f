# usage: LIBS="base occ3" VARIANT=21 WL=cggi bash tools/pmc_ab.sh   -- SQ / GRBM counters of the rotation kernel for alternative builds
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
cp $R/mktfhe_amd/lib/libmktfhe_hip.so /tmp/orig.so
for sfx in ${LIBS:-base}; do
 if [ "$sfx" != base ]; then cp $R/mktfhe_amd/lib/libmktfhe_hip_$sfx.so $R/mktfhe_amd/lib/libmktfhe_hip.so; else cp /tmp/orig.so $R/mktfhe_amd/lib/libmktfhe_hip.so; fi
 export MKT_ROT_VARIANT=${VARIANT:-0}
 D=$R/gpurun_out/pmcab_${sfx}_${VARIANT:-0}; rm -rf $D; mkdir -p $D
 ARGS="--workload ${WL:-kms2_n1024} --batch ${BATCH:-1024} --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-roofline"
 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $D/a -- python3 $R/bench.py $ARGS > /dev/null 2>&1
 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $D/b -- python3 $R/bench.py $ARGS > /dev/null 2>&1
 python3 - "$D" "$sfx" <<'PY'
import csv, glob, sys
d, sfx = sys.argv[1], sys.argv[2]
agg = {}
for f in glob.glob(d + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'blindrotate' not in k and 'ccs_' not in k: continue
        agg.setdefault(r['Counter_Name'], []).append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r.get('VGPR_Count', '?'), r.get('LDS_Block_Size', '?')))
out = {c: sum(x[0] for x in v) / len(v) for c, v in agg.items()}
ms = {c: sum(x[1] for x in v) / len(v) for c, v in agg.items()}
wc = out.get('SQ_WAVE_CYCLES', 1)
print(sfx, 'ms %.2f' % ms.get('SQ_WAVE_CYCLES', 0), 'vgpr/lds', agg.get('SQ_WAVE_CYCLES', [(0, 0, '?', '?')])[0][2:],
      'valu/wave %.3f' % (out.get('SQ_ACTIVE_INST_VALU', 0) / wc), 'lds/wave %.3f' % (out.get('SQ_ACTIVE_INST_LDS', 0) / wc),
      'wait_any %.3f' % (out.get('SQ_WAIT_ANY', 0) / wc), 'wait_inst %.3f' % (out.get('SQ_WAIT_INST_ANY', 0) / wc), 'active_any %.3f' % (out.get('SQ_ACTIVE_INST_ANY', 0) / wc),
      'insts_valu %.4g' % out.get('SQ_INSTS_VALU', 0), 'busy %.4g' % out.get('SQ_BUSY_CYCLES', 0),
      '| clock GHz %.3f' % (out.get('GRBM_GUI_ACTIVE', 0) / 8 / (ms.get('GRBM_GUI_ACTIVE', 1) * 1e-3) / 1e9), 'lds_idx %.4g' % out.get('SQ_LDS_IDX_ACTIVE', 0), 'bank_conf %.4g' % out.get('SQ_LDS_BANK_CONFLICT', 0),
      'wait_lds %.4g' % out.get('SQ_WAIT_INST_LDS', 0), 'vmem_cyc %.4g' % out.get('SQ_INST_CYCLES_VMEM', 0), 'waves %.4g' % out.get('SQ_WAVES', 0))
PY
done
cp /tmp/orig.so $R/mktfhe_amd/lib/libmktfhe_hip.so
