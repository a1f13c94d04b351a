#!/bin/bash
# same-box A/B of library options on the whole training step:  tools/ab_bench.sh MIA_OPTIONS=conv_xcd=0 MIA_OPTIONS=conv_xcd=1  (any VAR=value) [...]
for o in "$@" "$@"; do
  env $o python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o', d['ms_per_step'], d['value'])"
done
