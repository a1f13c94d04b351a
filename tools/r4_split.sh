# fp32 split-bf16 option: tests, then same-box A/B of the fp32 configs.   bash tools/r4_split.sh  (through gpurun)
cd /root/repo
python -m pytest tests -m gpu -x -q -k "f32_split" > gpurun_out/r4s_tests.log 2>&1; tail -5 gpurun_out/r4s_tests.log
for cfg in cfg2 cfg4; do for v in 0 1 0 1; do
  MIA_F32_SPLIT=$v python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/r4s_$cfg.$v.err | tail -1 > gpurun_out/r4s_$cfg.$v.json
  python -c "import json; d=json.load(open('gpurun_out/r4s_$cfg.$v.json')); print('$cfg f32_split=$v', d['ms_per_step'], d['value'], {k: d[k] for k in d if k.startswith('parity')})"
done; done
