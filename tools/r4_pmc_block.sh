# PMC passes over the canonical fused PlainBlock (64 -> 64 @ 512 x 512 x 32, bf16; forward + backward in one process):
#   bash tools/r4_pmc_block.sh <tag> [--nl 0]      -> gpurun_out/pmc_<tag>_{sq,sq2,fetch,write}/ ; summarise with tools/r4_pmc_block.py
# (separate --pmc passes, kernel trace only: MI355X_MICROARCH.md "rocprofv3 PMC slots")
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=/root/repo
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc $PMC -d $R/gpurun_out/pmc_${tag}_$name -o p --output-format csv -- python3 $R/tools/microbench.py block --c 64 --size 512 --batch 32 --iters 3 "$@" > $R/gpurun_out/pmc_${tag}_$name.log 2>&1; }
PMC="FETCH_SIZE"; run fetch "$@"
PMC="WRITE_SIZE"; run write "$@"
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; run sq "$@"
PMC="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; run sq2 "$@"
