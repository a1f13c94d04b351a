# PMC passes over an arbitrary single-kernel command:  bash tools/pmc_cmd.sh <tag> python3 tools/microbench.py conv --c 64 --cout 128 --stride 2
# (separate --pmc passes, no other trace domains; summarise with tools/pmc_summarise.py <tag> <kernel-substring> <label>)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=/root/repo
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc $PMC -d $R/gpurun_out/pmc_${tag}_$name -o p --output-format csv -- "$@" > $R/gpurun_out/pmc_${tag}_$name.log 2>&1; }
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; run sq "$@"
PMC="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; run sq2 "$@"
PMC="FETCH_SIZE"; run fetch "$@"
PMC="WRITE_SIZE"; run write "$@"
PMC="GRBM_GUI_ACTIVE"; run grbm "$@"
