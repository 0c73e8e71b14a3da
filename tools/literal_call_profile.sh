cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ktl
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktl -o kt -- python $R/bench.py --steps 300 --warmup 30 --settle 300 --no-pipeline --no-cpu-baseline --no-cfg4-one-gpu --no-live-traffic --no-train-step --no-stream --no-day-loops > /tmp/ktl.log 2>&1
DB=$(find /tmp/ktl -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB 16
