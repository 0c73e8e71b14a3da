cd /tmp && export TMPDIR=/tmp
run() { timeout 240 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc/$1 -- python /root/repo/tools/stage_profile.py cfg2_200x10k 3 > /tmp/pmc_$1.log 2>&1 || echo "pass $1 failed"; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run b "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"
run c "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR"
python /root/repo/tools/pmc_summary.py /tmp/pmc stage1 stage2 split
