cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_unet.py -m gpu -x -q -k "graph" > gpurun_out/r4g_tests.log 2>&1; tail -25 gpurun_out/r4g_tests.log
