# FETCH_SIZE of the stage-2 kernel under the scheduling switches (round-3 experiment, profiles/r03_k_s2traffic.txt):
# bash tools/s2_traffic.sh   (the GENIE_SEG2 rows of that profile came from an experiment-only switch of commit f1fa2e4)
cd /tmp && export TMPDIR=/tmp
for cfg in "GENIE_BPC2=3" "GENIE_BPC2=2" "GENIE_S2_WGMAP=1"; do
  rm -rf /tmp/pf; env $cfg timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python /root/repo/tools/stage_profile.py cfg2_200x10k 3 > /tmp/pf.log 2>&1
  echo "## $cfg"; python /root/repo/tools/pmc_summary.py /tmp/pf stage2
done
