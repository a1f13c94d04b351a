cd /root/repo
MIA_HIP_LIB=/root/repo/tools/ab/libmia_nlscalar.so python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "normalise_on_load or fused_level" 2>&1 | tail -2
for rep in 1 2 3; do for lib in "" /root/repo/tools/ab/libmia_nlscalar.so; do
  echo -n "lib=${lib:-production}  "; MIA_HIP_LIB=$lib python tools/microbench.py block --c 64 --size 512 --batch 32 --iters 20 --nl 1 2>&1 | tail -1
done; done
for rep in 1 2 3; do for lib in "" /root/repo/tools/ab/libmia_nlscalar.so; do
  echo -n "step lib=${lib:-production}  "; MIA_HIP_LIB=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['parts_ms'], d['roofline']['frac'])"
done; done
