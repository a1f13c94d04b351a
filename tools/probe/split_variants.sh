cd /root/repo
for shape in "64 256" "128 128" "256 64" "512 32"; do set -- $shape
  for lib in "" tools/ab/libmia_sv1.so tools/ab/libmia_sv2.so tools/ab/libmia_sv3.so tools/ab/libmia_sv4.so; do
    for sp in 0 1; do
      if [ -n "$lib" ] && [ $sp = 0 ]; then continue; fi
      echo -n "lib=${lib:-production} f32_split=$sp  "
      MIA_HIP_LIB=${lib:+/root/repo/$lib} MIA_F32_SPLIT=$sp python tools/microbench.py conv --c $1 --size $2 --batch 32 --dtype f32 --iters 20 2>&1 | tail -1
    done
  done
done
