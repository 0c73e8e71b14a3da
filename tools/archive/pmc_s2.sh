# PMC passes over the stage kernels (each counter group in its own rocprofv3 run): bash tools/pmc_s2.sh <outfile>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { timeout 240 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmcs/$1 -- python $R/tools/stage_profile.py cfg2_200x10k 3 > /tmp/pmcs_$1.log 2>&1 || { echo "pass $1 failed"; tail -3 /tmp/pmcs_$1.log; }; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run b "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM"
run c "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run d "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
python $R/tools/pmc_summary.py /tmp/pmcs stage1 stage2 split
