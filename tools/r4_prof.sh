# kernel stats of the cfg3 bench for one environment setting:  bash tools/r4_prof.sh <tag> [VAR=value ...]   (through gpurun)
tag=$1; shift
R=/root/repo
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_$tag.log 2>&1
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/prof_${tag}_kernel_stats.csv
