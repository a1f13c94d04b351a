cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r4p_aug -o p --output-format csv -- python3 /root/repo/tools/bench_augment.py --h0 496 --w0 608 --size 256 --iters 50 > /root/repo/gpurun_out/r4p_aug.log 2>&1
cp $(find /root/repo/gpurun_out/r4p_aug -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r4p_aug_kernel_stats.csv
tail -2 /root/repo/gpurun_out/r4p_aug.log
