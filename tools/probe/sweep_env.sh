# usage: bash tools/probe/sweep_env.sh VAR v1 v2 ...   (each value run twice: whole-step ms, forward-apply GB/s, its level-0 launch ms)
var=$1; shift
for v in "$@" "$@"; do
  env $var=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$var=$v', d['ms_per_step'], r['hbm_streams']['norm_act_fwd']['achieved'], r['parts_ms']['norm_act_fwd'])"
done
