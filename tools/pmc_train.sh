# PMC passes over the training step's backward kernels: bash tools/pmc_train.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmct
run() { timeout 300 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmct/$1 -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > /tmp/pmct_$1.log 2>&1 || { echo "pass $1 failed"; tail -3 /tmp/pmct_$1.log; }; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run b "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA"
run c "FETCH_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
run d "WRITE_SIZE"
python $R/tools/pmc_summary.py /tmp/pmct k_train_b1 k_train_b0 k_as_b1 k_train_b2
