# A/B of the association forward's P-sized kernels under library variants (genie_amd/lib/libgenie_*.so): per-kernel times of the
# bench's day-loops leg. Usage (GPU box): LIBS="libgenie_hip.so libgenie_x.so" bash tools/assoc_ab.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:-libgenie_hip.so}; do
  [ -f $R/genie_amd/lib/$lib ] || continue
  echo "== $lib"
  rm -rf /tmp/ra_$lib
  GENIE_LIB_PATH=$R/genie_amd/lib/$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ra_$lib -o ra -- python $R/bench.py --steps 5 --warmup 2 --settle 5 --no-cpu-baseline --no-cfg4-one-gpu --no-train-step --no-stream --no-live-traffic > /tmp/ra.log 2>&1
  grep -a "^{" /tmp/ra.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['day_loops_config2']['association_ms_per_source'], d['day_loops_config2']['refine_ms_per_source'])"
  python $R/tools/prof_summary.py $(find /tmp/ra_$lib -name "*.db" | head -1) 40 | grep "k_assoc_\|k_stage2_ord" | cut -c1-160
done
