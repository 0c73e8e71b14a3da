cd /tmp && export TMPDIR=/tmp
run() { timeout 240 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmcm/$1 -- python /root/repo/tools/stage_profile.py cfg2_200x10k 3 > /tmp/pmcm_$1.log 2>&1 || { echo "pass $1 failed"; tail -3 /tmp/pmcm_$1.log; }; }
run a "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
run b "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
run c "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum"
run d "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_BUSY_sum TA_TA_BUSY_sum"
python /root/repo/tools/pmc_summary.py /tmp/pmcm stage1 stage2
