cd /root/repo
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r5_full5.log 2>&1; tail -4 gpurun_out/r5_full5.log
out=gpurun_out/r05_bench_all.txt
echo "# bench.py --steps 20 --warmup 5 --no-cpu-baseline, every config, final round-5 build (finite-loss check on): ms_per_step img/s | host-fed ms_per_step | final loss" > $out
for cfg in cfg1 cfg2 cfg3 cfg4 cfg5; do
  python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_$cfg.json
  python -c "import json; d=json.load(open('gpurun_out/r05_bench_$cfg.json')); print('$cfg', d['ms_per_step'], d['value'], '|', d.get('ms_per_step_host_fed'), '|', d['final_loss'], d['config'].get('graph','')[:40])" >> $out
done
cat $out
