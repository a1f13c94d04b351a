for b in 64 128 256 512 64 128 256 512; do
  MIA_COL_SLABS=$b python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print($b, d['ms_per_step'], d['final_loss'])"
done
