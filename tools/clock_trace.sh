# Clock / power samples (rocm-smi, every ~0.2 s) while a stage-1 loop runs at config 2 and at config 4, next to the kernel's time per
# product node: evidence for (or against) "config 4's stage 1 is slower per node because a 17-ms kernel runs at a lower clock".
# Usage (GPU box): bash tools/clock_trace.sh > gpurun_out/clock_trace.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
sample() {  # $1 = label; samples until the file /tmp/clk_stop exists
  while [ ! -f /tmp/clk_stop ]; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr -s ' ' | tr '\n' ';'; echo " [$1]"
    sleep 0.2
  done
}
for cfg in cfg2_200x10k cfg4_2000x50k; do
  rm -f /tmp/clk_stop
  it=400; [ $cfg = cfg4_2000x50k ] && it=40
  sample $cfg > /tmp/clk_$cfg.txt &
  SP=$!
  timeout 600 python $R/tools/s1_time.py $cfg $it 2>&1 | tail -1
  touch /tmp/clk_stop; wait $SP
  echo "--- $cfg: samples during the loop (last 12 of $(wc -l < /tmp/clk_$cfg.txt))"
  tail -12 /tmp/clk_$cfg.txt
done
rm -f /tmp/clk_stop
