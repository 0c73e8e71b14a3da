#!/bin/bash
# PMC passes over the stage kernels only (tools/stage_profile.py), one counter group per pass: bash tools/pmc_quick.sh <tag>
TAG=${1:-pmc}
OUT=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { rm -rf /tmp/pp; timeout 240 rocprofv3 --pmc $2 --output-format csv -d /tmp/pp -- python /root/repo/tools/stage_profile.py cfg2_200x10k 5 > /tmp/pp.log 2>&1 || echo "pass $1 failed" >> $OUT/${TAG}_pmc.txt; echo "## pass $1: $2" >> $OUT/${TAG}_pmc.txt; python /root/repo/tools/pmc_summary.py /tmp/pp stage1 stage2 split >> $OUT/${TAG}_pmc.txt; }
rm -f $OUT/${TAG}_pmc.txt
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run sq2 "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"
run sq3 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run tcc "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
