# SQ stall breakdown of the conv fast kernel (one --pmc pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
for c in 64 256; do
  s=$((512*64/c)); [ $c = 256 ] && s=128
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d /root/repo/gpurun_out/pmc_c$c -o p --output-format csv -- python3 /root/repo/tools/microbench.py conv --c $c --size $s --batch 32 --iters 3 > /root/repo/gpurun_out/pmc_c$c.log 2>&1
done
ls /root/repo/gpurun_out/pmc_c64 /root/repo/gpurun_out/pmc_c256
