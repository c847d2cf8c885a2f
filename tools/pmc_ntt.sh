# usage (GPU box): [N=1024] bash tools/pmc_ntt.sh  -- SQ counters of the EXACT batched transform kernels (separate --pmc passes, each under timeout)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/pmcntt_${N:-1024}; rm -rf $D; mkdir -p $D
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $D/a -- python3 $R/tools/ntt_only.py ${N:-1024} 2 > /dev/null 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $D/b -- python3 $R/tools/ntt_only.py ${N:-1024} 2 > /dev/null 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE32_INSTS --output-format csv -d $D/c -- python3 $R/tools/ntt_only.py ${N:-1024} 2 > /dev/null 2>&1
python3 - "$D" <<'PY'
import csv, glob, sys
d = sys.argv[1]
for kern in ('ntt_fwd_kernel', 'ntt_inv_kernel'):
    agg = {}
    for f in glob.glob(d + '/*/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if kern not in r['Kernel_Name']: continue
            agg.setdefault(r['Counter_Name'], []).append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
    o = {c: sum(x[0] for x in v) / len(v) for c, v in agg.items()}
    ms = {c: sum(x[1] for x in v) / len(v) for c, v in agg.items()}
    wc = o.get('SQ_WAVE_CYCLES', 1); G = lambda k: o.get(k, 0)
    print(kern, 'ms %.2f' % ms.get('SQ_WAVE_CYCLES', 0), 'valu/wave %.3f' % (G('SQ_ACTIVE_INST_VALU') / wc), 'lds/wave %.3f' % (G('SQ_ACTIVE_INST_LDS') / wc),
          'wait_any %.3f' % (G('SQ_WAIT_ANY') / wc), 'wait_inst %.3f' % (G('SQ_WAIT_INST_ANY') / wc), 'insts_valu %.4g' % G('SQ_INSTS_VALU'), 'busy %.4g' % G('SQ_BUSY_CYCLES'),
          'wave_cycles %.4g' % wc, 'waves %.4g' % G('SQ_WAVES'), '| clock GHz %.3f' % (G('GRBM_GUI_ACTIVE') / 8 / (ms.get('GRBM_GUI_ACTIVE', 1) * 1e-3) / 1e9),
          'bank_conf/lds_idx %.3f' % (G('SQ_LDS_BANK_CONFLICT') / max(G('SQ_LDS_IDX_ACTIVE'), 1)), 'lds_idx %.4g' % G('SQ_LDS_IDX_ACTIVE'), 'wait_lds/wave %.3f' % (G('SQ_WAIT_INST_LDS') / wc),
          'insts_lds %.4g' % G('SQ_INSTS_LDS'), 'vmem_cycles %.4g' % G('SQ_INST_CYCLES_VMEM'), '| salu %.4g' % G('SQ_INSTS_SALU'), 'vmem %.4g' % G('SQ_INSTS_VMEM'),
          'act_sca/wave %.3f' % (G('SQ_ACTIVE_INST_SCA') / wc), 'act_vmem/wave %.3f' % (G('SQ_ACTIVE_INST_VMEM') / wc), 'act_misc/wave %.3f' % (G('SQ_ACTIVE_INST_MISC') / wc))
PY
