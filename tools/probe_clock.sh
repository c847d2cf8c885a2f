# usage: bash tools/probe_clock.sh  -- tools/bin/fft_probe under rocprofv3: clock held, VALU-active fraction and L2 fill traffic per probe kernel
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pp
PROBE_ITERS=${PROBE_ITERS:-16000} rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU FETCH_SIZE --output-format csv -d /tmp/pp -- $R/tools/bin/fft_probe > /tmp/probe.out 2>&1
grep -a "1024:" /tmp/probe.out
python3 - <<'PY'
import csv, glob, re
rows = {}
for f in glob.glob("/tmp/pp/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"probe<(\d+), (\d+), (\d+)>", r["Kernel_Name"])
        k = (int(r["Dispatch_Id"]), m.group(0) if m else r["Kernel_Name"][:30], int(r["Grid_Size"]))
        rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"]); rows[k]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, v in sorted(rows.items()):
    if k[2] < 100000: continue
    print(k[1], "grid", k[2], "ms %.2f" % v["ms"], "clock GHz %.3f" % (v["GRBM_GUI_ACTIVE"] / 8 / (v["ms"] * 1e-3) / 1e9), "valu/wave %.3f" % (v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]), "L2 fill GB/s %.0f" % (v.get("FETCH_SIZE", 0) * 2 * 1024 / 1e9 / (v["ms"] * 1e-3)))
PY
