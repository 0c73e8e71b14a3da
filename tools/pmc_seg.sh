cd /tmp && export TMPDIR=/tmp
for cfg in "0 1" "1 1" "1 1250" "1 256"; do set -- $cfg
  rm -rf /tmp/ps
  if [ "$1" = "1" ]; then export TUNE_STAPERM=1; else unset TUNE_STAPERM; fi
  export GENIE_SEG=$2
  timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/ps -- python /root/repo/tools/tune.py cfg2_200x10k SEG=$2 > /tmp/ps.log 2>&1 || echo fail
  echo "staperm=$1 SEG=$2: $(grep SEG /tmp/ps.log | tail -1 | cut -c40-100)"; python /root/repo/tools/pmc_summary.py /tmp/ps stage2_fast stage1_b3 | grep -A1 "^k_"
done
