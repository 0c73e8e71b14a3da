cd /tmp && export TMPDIR=/tmp
export GENIE_LIB_PATH=/root/repo/genie_amd/lib/libgenie_tune.so
export GENIE_SEG=${SEG:-1}
for abl in 0 128 2 1 32 384; do
  export GENIE_ABLATE=$abl
  rm -rf /tmp/pa
  timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pa -- python /root/repo/tools/stage_profile.py cfg2_200x10k 3 > /tmp/pa.log 2>&1 || echo fail
  echo "ABLATE=$abl SEG=$GENIE_SEG"; python /root/repo/tools/pmc_summary.py /tmp/pa stage2 | grep -v "^k_"
done
