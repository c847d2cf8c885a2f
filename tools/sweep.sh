# usage (GPU box, repo root):
#   bash tools/sweep.sh [VAR=v1,v2,...]... [--libs "base x y"] [--workloads "kms2_n1024 cggi"] [--batches "1024 16384"] [-- bench.py args]
# ONE parametrised sweep for what used to be a dozen one-off scripts: every combination of the given environment settings (kernel
# switches read at mkt_ctx_create: MKT_ROT_STAGGER, MKT_ROT_WIDE, MKT_ROT_VARIANT, MKT_ROT_BLKG, MKT_KS_G, MKT_KS_BLOCKS, MKT_KS_PAIR,
# MKT_CCS_STAGGER, MKT_EXACT_WIDE, MKT_FFT_GRID ...) x alternative builds mktfhe_amd/lib/libmktfhe_hip_<sfx>.so ("base" = the default;
# build them with tools/tu_variant.sh / tools/variant.sh / make SFX=) x workloads x batch sizes, one bench.py line each, printed as
# gates/s, rotation / key-switch / phase-2 ms, decrypt_ok.  Same device, same call: the only way to compare on a pool whose devices differ.
# examples:  bash tools/sweep.sh MKT_ROT_STAGGER=0,16,64 --workloads "kms2_n1024 kms2party cggi"
#            bash tools/sweep.sh MKT_ROT_WIDE=1,2 --workloads cggi --batches "96 128 256 512"
#            bash tools/sweep.sh --libs "base cab1 cab5" --workloads "ccs2party ccs8party"
#            bash tools/sweep.sh MKT_KS_G=8,16,32 MKT_KS_BLOCKS=1024,4096 -- --steps 5
LIBS=base; WORKLOADS=kms2_n1024; BATCHES=1024; VARS=(); EXTRA=()
while [ $# -gt 0 ]; do case "$1" in
  --libs) LIBS="$2"; shift 2;; --workloads) WORKLOADS="$2"; shift 2;; --batches) BATCHES="$2"; shift 2;;
  --) shift; EXTRA=("$@"); break;; *=*) VARS+=("$1"); shift;; *) echo "unknown argument $1"; exit 2;; esac; done
combos=("")
for v in "${VARS[@]}"; do name=${v%%=*}; next=(); for c in "${combos[@]}"; do for val in $(echo "${v#*=}" | tr ',' ' '); do next+=("$c $name=$val"); done; done; combos=("${next[@]}"); done
for sfx in $LIBS; do
  # the alternative build is SELECTED (MKT_LIB_PATH, mktfhe_amd/_lib.py), never copied over the default library: an interrupted sweep leaves nothing behind
  if [ "$sfx" != base ]; then export MKT_LIB_PATH=$PWD/mktfhe_amd/lib/libmktfhe_hip_$sfx.so; else unset MKT_LIB_PATH; fi
  for c in "${combos[@]}"; do for w in $WORKLOADS; do for b in $BATCHES; do
    env $c python3 bench.py --steps 3 --warmup 1 --workload $w --batch $b --no-cpu-baseline --no-roofline --no-secondary "${EXTRA[@]}" 2>/dev/null | grep -a '"metric"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d['kernels_ms_per_step']
    print('$sfx |$c |', d['config']['params'], 'batch $b | %.0f gates/s | rot %.3f ks %.3f p2 %.3f ms | %.3f ms/step | ok' % (d['value'], k['blindrotate'], k['keyswitch'], k['kms_phase2'], d['ms_per_step']), d['decrypt_ok'], flush=True)"
  done; done; done
done
unset MKT_LIB_PATH
