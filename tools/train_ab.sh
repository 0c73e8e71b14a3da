# A/B of the training step under library variants: LIBS="libgenie_hip.so libgenie_x.so" bash tools/train_ab.sh  (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:-libgenie_hip.so}; do
  [ -f $R/genie_amd/lib/$lib ] || continue
  echo "== $lib"
  rm -rf /tmp/rt_$lib
  GENIE_LIB_PATH=$R/genie_amd/lib/$lib timeout 300 python $R/bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['phase_ms'], d['four_output_step']['ms_per_step'])"
  GENIE_LIB_PATH=$R/genie_amd/lib/$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rt_$lib -o rt -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > /tmp/rt.log 2>&1
  python $R/tools/prof_summary.py $(find /tmp/rt_$lib -name "*.db" | head -1) 40 | grep "${KERN:-k_train_b2}" | cut -c1-160
done
