# FETCH_SIZE of the stage-2 kernel under different sweep orders (round 3 experiment): bash tools/s2_traffic.sh
cd /tmp && export TMPDIR=/tmp
for cfg in "GENIE_SEG2=1" "GENIE_SEG2=8" "GENIE_SEG2=32" "GENIE_BPC2=2" "GENIE_S2_WGMAP=1"; do
  rm -rf /tmp/pf; env $cfg timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python /root/repo/tools/stage_profile.py cfg2_200x10k 3 > /tmp/pf.log 2>&1
  echo "## $cfg"; python /root/repo/tools/pmc_summary.py /tmp/pf stage2
done
