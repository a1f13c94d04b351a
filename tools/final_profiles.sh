# Round-end artifacts of one build, one box:  bash tools/final_profiles.sh   (run through gpurun; writes gpurun_out/final_*)
set -o pipefail
R=/root/repo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_prof -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/final_prof.log 2>&1
cd $R
python tools/s2_levels.py > gpurun_out/final_s2_levels.txt 2>&1
python tools/ab_levels.py conv_bt > gpurun_out/final_ab_levels_conv_bt.txt 2>&1
for cfg in cfg3 cfg5 cfg4 cfg2; do python bench.py --config $cfg --steps 10 --warmup 3 > gpurun_out/final_bench_$cfg.json 2> gpurun_out/final_bench_$cfg.err; done
