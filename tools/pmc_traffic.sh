# HBM traffic of the canonical 64->64 3x3 launch: FETCH_SIZE and WRITE_SIZE in two separate --pmc passes (guide: TCC slots)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /root/repo/gpurun_out/pmc_fetch -o p --output-format csv -- python3 /root/repo/tools/microbench.py conv --c 64 --size 512 --batch 32 --iters 3 > /root/repo/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /root/repo/gpurun_out/pmc_write -o p --output-format csv -- python3 /root/repo/tools/microbench.py conv --c 64 --size 512 --batch 32 --iters 3 > /root/repo/gpurun_out/pmc_write.log 2>&1
ls /root/repo/gpurun_out/pmc_fetch /root/repo/gpurun_out/pmc_write
