#!/bin/bash
# Round profile set: bench line, rocprofv3 kernel-trace summary of the same command, PMC passes (one counter group per pass).
# Usage (on the GPU box): bash tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_*
TAG=${1:-r01_e}
OUT=$GRAFT_REPO_ROOT/gpurun_out
[ -z "$GRAFT_REPO_ROOT" ] && OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 python /root/repo/bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python /root/repo/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-cfg4-one-gpu --no-live-traffic --no-stream --no-day-loops > /tmp/kt.log 2>&1
grep "^{" /tmp/kt.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
DB=$(find /tmp/kt -name "*.db" | head -1)
python /root/repo/tools/prof_summary.py $DB 24 > $OUT/${TAG}_kernel_stats.txt
run() { rm -rf /tmp/pp; timeout 240 rocprofv3 --pmc $2 --output-format csv -d /tmp/pp -- python /root/repo/tools/stage_profile.py cfg2_200x10k 5 > /tmp/pp.log 2>&1 || echo "pass $1 failed" >> $OUT/${TAG}_pmc.txt; echo "## pass $1: $2" >> $OUT/${TAG}_pmc.txt; python /root/repo/tools/pmc_summary.py /tmp/pp stage1 stage2 split >> $OUT/${TAG}_pmc.txt; }
rm -f $OUT/${TAG}_pmc.txt
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run sq2 "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_ANY"
run tcc "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
