for b in 4096 32768 65536 131072 4096 32768 65536 131072; do
  MIA_STREAM_BLOCKS=$b python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); h=d['roofline']['hbm_streams']; print($b, d['ms_per_step'], h['norm_act_fwd']['achieved'], h['norm_act_bwd']['achieved'], d['roofline']['parts_ms']['norm_act_fwd'])"
done
