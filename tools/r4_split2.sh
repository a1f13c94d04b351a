cd /root/repo
python tools/probe/split_accuracy.py > gpurun_out/split_accuracy.txt 2> gpurun_out/split_accuracy.err; cat gpurun_out/split_accuracy.txt
cd /tmp && export TMPDIR=/tmp
MIA_F32_SPLIT=1 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r4s_prof -o p --output-format csv -- python3 /root/repo/bench.py --config cfg2 --steps 10 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/r4s_prof.log 2>&1
cp $(find /root/repo/gpurun_out/r4s_prof -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r4s_cfg2_split_kernel_stats.csv
head -25 /root/repo/gpurun_out/r4s_cfg2_split_kernel_stats.csv | cut -c1-200
