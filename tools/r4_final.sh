# Round-4 artifacts of one build on one box:  bash tools/r4_final.sh   (through gpurun; writes gpurun_out/r4f_*)
R=/root/repo
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r4f_gpu_tests.log 2>&1; tail -3 gpurun_out/r4f_gpu_tests.log
for cfg in cfg3 cfg2 cfg4 cfg5; do python bench.py --config $cfg --steps 10 --warmup 3 > gpurun_out/r4f_bench_$cfg.json 2> gpurun_out/r4f_bench_$cfg.err; tail -c 300 gpurun_out/r4f_bench_$cfg.json; echo; done
MIA_F32_SPLIT=1 python bench.py --config cfg2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4f_bench_cfg2_f32_split.json 2>/dev/null
MIA_F32_SPLIT=1 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4f_bench_cfg4_f32_split.json 2>/dev/null
python bench.py --config cfg1 --steps 200 --warmup 5 --no-cpu-baseline --graph > gpurun_out/r4f_bench_cfg1_graph.json 2>/dev/null
python bench.py --config cfg1 --steps 200 --warmup 5 --no-cpu-baseline > gpurun_out/r4f_bench_cfg1.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4f_prof -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r4f_prof.log 2>&1
cp $(find $R/gpurun_out/r4f_prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r4f_kernel_stats.csv
