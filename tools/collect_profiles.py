"""Copy the judged summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/ (tracked).
usage: python tools/collect_profiles.py <tag> [workload]"""
import csv, glob, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
wl = sys.argv[2] if len(sys.argv) > 2 else "kms2_n1024"
src, dst = f"gpurun_out/{tag}", "profiles"
os.makedirs(dst, exist_ok=True)
for f in glob.glob(f"{src}/bench_*.json"):
    lines = [l for l in open(f) if l.startswith('{"metric"')]
    if lines:
        open(f"{dst}/{tag}_{os.path.basename(f)}", "w").write("".join(lines))
newest = lambda files: sorted(files, key=os.path.getmtime)[-1:]          # a directory keeps the files of earlier runs of the round: the last run counts
for f in newest(glob.glob(f"{src}/trace_{wl}/*/*kernel_stats.csv")):
    shutil.copy(f, f"{dst}/{tag}_bench_{wl}_kernel_stats.csv")
dirs = sorted(d for d in glob.glob(f"{src}/pmc_{wl}_*") if os.path.basename(d)[len(f"pmc_{wl}_"):] in ("fetch", "write", "clk", "sq1", "sq2"))   # exactly this name's passes (kms2_n1024 must not swallow kms2_n1024_exact)
if dirs:
    with open(f"{dst}/{tag}_bench_{wl}_pmc.txt", "w") as out:
        out.write(f"# rocprofv3 --kernel-trace --pmc <counters> (separate passes), python3 bench.py --steps 2 --warmup 0 --workload {wl} --no-cpu-baseline --no-secondary\n")
        meta = dict(l.strip().split(" ", 1) for l in open(f"{src}/pmc_{wl}_meta.txt") if " " in l.strip()) if os.path.exists(f"{src}/pmc_{wl}_meta.txt") else {}
        git = os.popen("git rev-parse --short HEAD 2>/dev/null").read().strip() + ("+dirty" if os.popen("git status --porcelain -- mktfhe_amd include 2>/dev/null").read().strip() else "")
        out.write(f"# build_id: {meta.get('build_id', 'unknown')}  so_sha256: {meta.get('so_sha256', 'unknown')}  device: {meta.get('device', 'unknown')}  git (at collection): {git}\n")
        out.write("# FETCH_SIZE / WRITE_SIZE unit: KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of a coalesced read stream -> x2.\n")
        out.write("# kernel, grid, counter, mean value over the full-work dispatches (within 15 % of the longest of that kernel and grid), mean duration ms, dispatches\n")
        for d in dirs:
            agg = {}
            for f in newest(glob.glob(f"{d}/*/*counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"]
                    if "mktd" not in k: continue
                    key = (k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0], r["Grid_Size"], r["Counter_Name"])   # kernels of unnamed namespaces keep their own rows
                    agg.setdefault(key, []).append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
            for (k, g, c), v in sorted(agg.items()):
                # the untimed input folds of bench.py launch the same kernel on sparser ciphertexts (fewer parties involved: shorter);
                # the summary is of the full-work launches: those within 15 % of the longest of this (kernel, grid)
                top = max(x[1] for x in v)
                v = [x for x in v if x[1] >= 0.85 * top]
                out.write(f"{k},{g},{c},{sum(x[0] for x in v)/len(v):.6g},{sum(x[1] for x in v)/len(v):.4f},{len(v)}\n")
print(sorted(os.listdir(dst)))
