# PMC passes on the canonical 64->64 3x3 launch (persistent conv64 kernel): SQ stall breakdown, FETCH_SIZE, WRITE_SIZE in
# separate --pmc passes (TCC slot limits), kernel-trace only.  Usage: bash tools/pmc_conv64.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_${tag}_$name -o p --output-format csv -- python3 $R/tools/microbench.py conv --c 64 --size 512 --batch 32 --iters 3 > $R/gpurun_out/pmc_${tag}_$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
ls $R/gpurun_out/pmc_${tag}_*
