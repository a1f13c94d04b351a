# A/B of the warm-up feed (bench.py, MIA_BENCH_WARM): same box, interleaved
cd /root/repo
out=gpurun_out/r05_ab_boost_state.txt
echo "# bench.py --steps 20 --warmup 5: ms_per_step (resident timed loop) / ms_per_step_host_fed, by warm-up feed; same box, interleaved" > $out
for cfg in cfg3 cfg2 cfg4 cfg5 cfg1; do for w in resident host resident host; do
  MIA_BENCH_WARM=$w python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg warm-up=$w', d['ms_per_step'], 'ms/step', d['value'], 'img/s | host-fed loop', d.get('ms_per_step_host_fed'))" >> $out
done; done
cat $out
