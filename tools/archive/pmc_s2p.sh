# PMC passes over the stage kernels of an irregular product graph (tools/s2p_time.py): bash tools/pmc_s2p.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmcp
run() { timeout 300 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmcp/$1 -- python $R/tools/s2p_time.py > /tmp/pmcp_$1.log 2>&1 || { echo "pass $1 failed"; tail -3 /tmp/pmcp_$1.log; }; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run b "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_ANY"
run c "FETCH_SIZE"
run d "TCC_HIT_sum TCC_MISS_sum"
run e "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
python $R/tools/pmc_summary.py /tmp/pmcp k_stage2_pcsr k_stage1_h2 k_bip_out_seg
