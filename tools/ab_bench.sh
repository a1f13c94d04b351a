#!/bin/bash
# same-box A/B of library options on the whole training step:  tools/ab_bench.sh conv_xcd=0 conv_xcd=1 [...]
for o in "$@" "$@"; do
  MIA_OPTIONS=$o python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o', d['ms_per_step'], d['value'])"
done
