# PMC passes on the 64->64 3x3 weight-gradient launch: bash tools/pmc_wgrad.sh <tag> <w8 0|1>
tag=${1:-r02w}; w8=${2:-1}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
export MIA_WGRAD_W8=$w8
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_${tag}_$name -o p --output-format csv -- python3 $R/tools/microbench.py wgrad --c 64 --size 512 --batch 32 --iters 3 > $R/gpurun_out/pmc_${tag}_$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC
run fetch FETCH_SIZE
run grbm GRBM_GUI_ACTIVE
