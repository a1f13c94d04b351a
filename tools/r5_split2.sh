cd /root/repo
python tools/probe/amax_calls.py > gpurun_out/r5_amax_calls.txt 2>&1; cat gpurun_out/r5_amax_calls.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5s2_tests.log 2>&1; tail -5 gpurun_out/r5s2_tests.log
for cfg in cfg2 cfg4; do for v in 0 1; do
  MIA_F32_SPLIT=$v python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/r5s2_$cfg.$v.err | tail -1 > gpurun_out/r5s2_$cfg.$v.json
  python -c "import json; d=json.load(open('gpurun_out/r5s2_$cfg.$v.json')); print('$cfg f32_split=$v', d['ms_per_step'], d['value'])"
done; done
