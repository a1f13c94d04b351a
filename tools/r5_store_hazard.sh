# Store-data hazard of the column-reduce epilogue (csrc/conv64.hip, round 4), re-measured for the record:
# the same library with 0 / 2 / 4 / 8 / 16 wait states behind the epilogue's 16-byte store; tools/probe/diag_cr4.py counts the tiles of
# 10 x 32768 whose output differs from the plain input-gradient launch, diag_cr3.py lists which lanes / dwords of the first bad tiles.
#   HERE (no GPU):  bash tools/r5_store_hazard.sh build      then      gpurun -- 'bash tools/r5_store_hazard.sh run'
cd /root/repo
PKG=medical-image-analysis_amd
if [ "$1" = build ]; then
  mkdir -p tools/ab
  # the column-reduce epilogue lives in probe builds only (-DMIA_EXPERIMENTS: conv64.hip, its entry points in conv_mma.hip / norm.hip)
  for f in conv_mma norm; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -DMIA_EXPERIMENTS -c $PKG/csrc/$f.hip -o tools/ab/${f}_exp.o || exit 1
  done
  for pad in 0 2 4 8 16; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -DMIA_EXPERIMENTS -DCR_PAD=$pad -c $PKG/csrc/conv64.hip -o tools/ab/conv64_pad$pad.o || exit 1
    objs=$(ls $PKG/mia_hip/_obj/*.o | grep -v "/conv64.o\|/conv_mma.o\|/norm.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libmia_crpad$pad.so $objs tools/ab/conv64_pad$pad.o tools/ab/conv_mma_exp.o tools/ab/norm_exp.o || exit 1
  done
  ls -la tools/ab/*.so
  exit 0
fi
out=gpurun_out/r05_store_hazard.txt
echo "# diag_cr4: bad tiles of 10 launches x 32768 tiles (64 -> 64 input gradient with the column-reduce epilogue, 32 x 512 x 512 bf16) by wait states behind the store" > $out
for pad in 0 2 4 8; do
  MIA_FUSE_CR=1 MIA_HIP_LIB=/root/repo/tools/ab/libmia_crpad$pad.so python tools/probe/diag_cr4.py 2>&1 | tail -1 | sed "s/^/pad $pad: /" >> $out
done
MIA_FUSE_CR=1 MIA_HIP_LIB=/root/repo/tools/ab/libmia_crpad16.so python tools/probe/diag_cr4.py 2>&1 | tail -1 | sed "s/^/pad 16 (the padding the epilogue carries): /" >> $out
echo "# diag_cr3 on the unpadded build: which lanes / dwords of the first bad tiles" >> $out
MIA_FUSE_CR=1 MIA_HIP_LIB=/root/repo/tools/ab/libmia_crpad0.so python tools/probe/diag_cr3.py 2>&1 | tail -25 >> $out
cat $out
