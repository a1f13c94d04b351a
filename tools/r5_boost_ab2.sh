cd /root/repo
out=gpurun_out/r05_ab_boost_state2.txt
echo "# bench.py --steps 20 --warmup 5, warm-up = 3 resident + 2 host-fed steps (default) vs 5 resident: ms_per_step of the timed resident loop / of the host-fed loop behind it" > $out
for cfg in cfg2 cfg3 cfg4 cfg5; do for w in resident host resident host; do
  MIA_BENCH_WARM=$w python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg warm-up=$w', d['ms_per_step'], 'ms/step', d['value'], 'img/s | host-fed loop', d.get('ms_per_step_host_fed'))" >> $out
done; done
cat $out
