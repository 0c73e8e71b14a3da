# A/B of the G-sized tail: per-kernel times of tools/tail_time.py under rocprofv3 for the built library and (if present) a
# second one at genie_amd/lib/libgenie_old.so. Usage (GPU box): bash tools/tail_ab.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:-libgenie_hip.so libgenie_old.so}; do
  [ -f $R/genie_amd/lib/$lib ] || continue
  echo "== $lib"
  rm -rf /tmp/rp_$lib
  GENIE_LIB_PATH=$R/genie_amd/lib/$lib python $R/tools/tail_time.py 2>&1 | grep "per-window\|batched"
  GENIE_LIB_PATH=$R/genie_amd/lib/$lib timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/rp_$lib -o rp -- python $R/tools/tail_time.py > /dev/null 2>&1
  python $R/tools/prof_summary.py $(find /tmp/rp_$lib -name "*.db" | head -1) 14 | grep "k_readout\|k_sa_\|k_ro_\|k_bip" | cut -c1-150
done
