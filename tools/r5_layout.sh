cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r5_layout -o t --output-format csv -- python3 /root/repo/tools/probe/layout_ab.py > /root/repo/gpurun_out/r5_layout.log 2>&1
tail -2 /root/repo/gpurun_out/r5_layout.log
python3 /root/repo/tools/probe/layout_ab_diff.py $(find /root/repo/gpurun_out/r5_layout -name "*kernel_trace.csv" | head -1) | tee /root/repo/gpurun_out/r5_layout_diff.txt
rm -rf /root/repo/gpurun_out/r5_layout
