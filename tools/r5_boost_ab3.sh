cd /root/repo
out=gpurun_out/r05_ab_boost_state3.txt
echo "# bench.py --steps 20 --warmup 5 (final form: every warm-up step host-fed; TrainEngine uses plain .to() for its first 3 host batches, HostFeed after): timed resident loop / host-fed loop behind it" > $out
for cfg in cfg1 cfg2 cfg3 cfg4 cfg5; do for w in resident host resident host; do
  MIA_BENCH_WARM=$w python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg warm-up=$w', d['ms_per_step'], 'ms/step', d['value'], 'img/s | host-fed loop', d.get('ms_per_step_host_fed'))" >> $out
done; done
cat $out
