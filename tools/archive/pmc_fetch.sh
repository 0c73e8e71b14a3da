# FETCH_SIZE / WRITE_SIZE of the stage kernels (separate passes): bash tools/pmc_fetch.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmcf
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcf/$c -- python $R/tools/stage_profile.py cfg2_200x10k 3 > /tmp/pmcf_$c.log 2>&1 || { echo "pass $c failed"; tail -3 /tmp/pmcf_$c.log; }
done
python $R/tools/pmc_summary.py /tmp/pmcf stage1 stage2 split
