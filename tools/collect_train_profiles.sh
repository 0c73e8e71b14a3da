#!/bin/bash
# Training-step profile set: bench line (--mode train), rocprofv3 kernel-trace summary of the same command.
# Usage (on the GPU box): bash tools/collect_train_profiles.sh <tag>     -> gpurun_out/<tag>_*
TAG=${1:-r04_h}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py --mode train --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_train.json
rm -rf /tmp/ktt
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/ktt -o kt -- python $R/bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline > /tmp/ktt.log 2>&1
grep -a "^{" /tmp/ktt.log | tail -1 > $OUT/${TAG}_bench_train_under_rocprof.json
DB=$(find /tmp/ktt -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB 30 > $OUT/${TAG}_train_kernel_stats.txt
