# round 5, first GPU call: fp16-denormal behaviour of the f16 MFMA + baseline of the fp32 configs at f32_split = 0 / 1
cd /root/repo
./tools/probe/mfma_f16_denorm > gpurun_out/r5_f16_denorm.txt 2>&1; cat gpurun_out/r5_f16_denorm.txt
for cfg in cfg2 cfg4; do for v in 0 1; do
  MIA_F32_SPLIT=$v python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/r5p_$cfg.$v.err | tail -1 > gpurun_out/r5p_$cfg.$v.json
  python -c "import json; d=json.load(open('gpurun_out/r5p_$cfg.$v.json')); print('$cfg f32_split=$v', d['ms_per_step'], d['value'])"
done; done
