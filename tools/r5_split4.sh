cd /root/repo
for cfg in cfg2 cfg4; do for v in 0 1; do
  MIA_F32_SPLIT=$v python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/r5s4_$cfg.$v.err | tail -1 > gpurun_out/r5s4_$cfg.$v.json
  python -c "import json; d=json.load(open('gpurun_out/r5s4_$cfg.$v.json')); print('$cfg f32_split=$v', d['ms_per_step'], d['value'])"
done; done
bash tools/r5_prof_cfg.sh cfg2
bash tools/r5_prof_cfg.sh cfg4
