cd /tmp && export TMPDIR=/tmp
export GENIE_LIB_PATH=/root/repo/genie_amd/lib/libgenie_tune.so
for cfg in "1 0" "1 33" "1250 33" "256 33" "1250 35" "1 35"; do set -- $cfg
  rm -rf /tmp/ps
  export GENIE_SEG=$1 GENIE_ABLATE=$2 TUNE_STAPERM=1
  timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/ps -- python /root/repo/tools/stage_profile.py cfg2_200x10k 5 > /tmp/ps.log 2>&1 || echo fail
  echo "SEG=$1 ABLATE=$2"; python /root/repo/tools/pmc_summary.py /tmp/ps stage2_fast | grep FETCH
done
