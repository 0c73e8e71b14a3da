# HBM / fabric read traffic of the training backward kernels (one counter per pass): bash tools/pmc_train_fetch.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmctf
run() { timeout 300 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmctf/$1 -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > /tmp/pmctf_$1.log 2>&1 || { echo "pass $1 failed"; tail -3 /tmp/pmctf_$1.log; }; }
run a "FETCH_SIZE"
run b "TCC_HIT_sum TCC_MISS_sum"
run c "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
python $R/tools/pmc_summary.py /tmp/pmctf k_train_b1 k_train_b0 k_as_b1 k_train_b2
