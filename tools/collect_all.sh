# usage: bash tools/collect_all.sh <tag>  -- every gpurun_out/<tag>/pmc_<name>_* and trace_<name> -> profiles/<tag>_bench_<name>_{pmc.txt,kernel_stats.csv}
TAG=${1:-r04}
for n in $(ls gpurun_out/$TAG | sed -n 's/^pmc_\(.*\)_\(fetch\|write\|clk\|sq1\|sq2\)$/\1/p' | sort -u); do python3 tools/collect_profiles.py $TAG $n > /dev/null; done
ls profiles | grep "^${TAG}_" | tr '\n' ' '
