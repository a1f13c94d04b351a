# HBM traffic (PMC) of the canonical full-resolution C0 -> C0 PlainBlock of a config: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# over tools/microbench.py block (forward + backward of one block), summarised by tools/r5_pmc_block.py into
# profiles/r05_pmc_canonical_block_<cfg>.{csv,json} (bench.py reads the JSON for roofline.traffic).
#   bash tools/r5_pmc_block.sh cfg5      (through gpurun)
cfg=$1
case $cfg in
  cfg2) C=64; S=256; B=32; DT=f32; NL=0;;
  cfg3) C=64; S=512; B=32; DT=bf16; NL=1;;
  cfg5) C=96; S=768; B=16; DT=bf16; NL=0;;
  *) echo "unknown config"; exit 1;;
esac
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for PMC in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $PMC -d $R/gpurun_out/pmc5_${cfg}_$PMC -o p --output-format csv -- python3 $R/tools/microbench.py block --c $C --size $S --batch $B --dtype $DT --nl $NL --iters 3 > $R/gpurun_out/pmc5_${cfg}_$PMC.log 2>&1
done
python3 $R/tools/r5_pmc_block.py $cfg $C $S $B $DT
rm -rf $R/gpurun_out/pmc5_${cfg}_FETCH_SIZE $R/gpurun_out/pmc5_${cfg}_WRITE_SIZE
cp $R/profiles/r05_pmc_canonical_block_${cfg}.json $R/profiles/r05_pmc_canonical_block_${cfg}.csv $R/gpurun_out/   # (profiles/ on the GPU box is not merged back; gpurun_out/ is)
