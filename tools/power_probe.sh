# usage (GPU box): WL=lmss BATCH=16384 bash tools/power_probe.sh  -- samples rocm-smi power / clocks while a workload runs (is the part power-limited under this kernel?)
R=$GRAFT_REPO_ROOT; cd $R
rocm-smi --showpower --showclocks --showtemp 2>&1 | head -40
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' ' | sed 's/=\+//g'; echo; sleep 0.25; done ) > /tmp/smi.log 2>&1 &
SMI=$!
python3 bench.py --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --workload ${WL:-lmss} --batch ${BATCH:-16384} ${ARGS:-} 2>&1 | grep '"metric"' | cut -c1-300
wait $SMI
cat /tmp/smi.log | cut -c1-260 | awk 'NR%2==0' | head -30
