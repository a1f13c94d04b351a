# round 5: split-f16 default -- tests, then the fp32 configs (same box: exact vs split)
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -x -q -k "f32_split or batched_weight_pack or kernel_selection" > gpurun_out/r5s_tests.log 2>&1; tail -15 gpurun_out/r5s_tests.log
python __graft_entry__.py smoke 2>&1 | tail -2
for cfg in cfg2 cfg4; do for v in 0 1; do
  MIA_F32_SPLIT=$v python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/r5s_$cfg.$v.err | tail -1 > gpurun_out/r5s_$cfg.$v.json
  python -c "import json; d=json.load(open('gpurun_out/r5s_$cfg.$v.json')); print('$cfg f32_split=$v', d['ms_per_step'], d['value'], {k: d[k] for k in d if k.startswith('parity')})"
done; done
