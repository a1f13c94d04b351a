cd /root/repo
run() { # cfg steps extra-env graphflag
  out=$(env $3 python bench.py --config $1 --steps $2 --warmup 5 --no-cpu-baseline $4 2>gpurun_out/r4g_err.txt | tail -1)
  python - "$1" "$3" "$4" <<PY
import json,sys
d=json.loads('''$out''')
print(sys.argv[1], sys.argv[2] or '-', sys.argv[3] or 'eager', d['ms_per_step'], d['value'], d.get('ms_per_step_without_augmentation'), d['final_loss'])
PY
}
for rep in 1 2; do
run cfg1 200 "" ""; run cfg1 200 "" --graph
run cfg4 20 "" ""; run cfg4 20 "" --graph
run cfg4 20 MIA_F32_SPLIT=1 ""; run cfg4 20 MIA_F32_SPLIT=1 --graph
run cfg2 10 MIA_F32_SPLIT=1 ""; run cfg2 10 MIA_F32_SPLIT=1 --graph
run cfg3 10 "" ""; run cfg3 10 "" --graph
done
