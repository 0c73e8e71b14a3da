# k_train_b1 ablations (GENIE_TUNING build at genie_amd/lib/libgenie_tune.so): bash tools/train_abl.sh
# bits (GENIE_TRABL): 8 no dt stores, 16 per-phase clocks with drains at the phase boundaries (printf). The round-4 ablations that
# removed the transposed means (1, 2) and the weight-gradient section (4) ran on the pre-pipeline kernel (commit 85211d7 + hooks).
for abl in ${ABLS:-0 16}; do
  GENIE_TRABL=$abl GENIE_LIB_PATH=$PWD/genie_amd/lib/libgenie_tune.so timeout 200 python bench.py --mode train --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline 2>&1 | grep -a "b1 blk\|metric" | sed 's/^{"metric.*"ms_per_step": \([0-9.]*\).*bwd_front": \([0-9.]*\).*/ms_per_step \1 bwd_front \2/' | sort | uniq -c | sort -rn | head -8
done
