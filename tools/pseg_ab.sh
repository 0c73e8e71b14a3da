#!/bin/bash
# needs: python -c "from genie_amd import _lib; _lib.build(extra_flags=['-DGENIE_TUNING=1','-DGENIE_ABL_MFMA=0'], out_path='genie_amd/lib/variants/libgenie_tune.so')"
# Round 6, VERDICT item 7: stage 2 of the irregular product graph (use_subgraph) as k_stage2_pcsr + k_seg_sum32 (GENIE_S2_PSEG=0) against
# k_stage2_pseg (a wave per source node, station sum folded in). Same box, tuning build. Output: gpurun_out/r06_pseg_ab.txt
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r06_pseg_ab.txt
export GENIE_LIB_PATH=$R/genie_amd/lib/variants/libgenie_tune.so
cd /tmp && export TMPDIR=/tmp
{
for rep in 1 2; do for v in 0 1; do
  echo "== GENIE_S2_PSEG=$v run $rep"
  GENIE_S2_PSEG=$v timeout 600 python $R/tools/bench_variants.py subgraph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('ms_per_window','ms_per_window_batched_tails','ns_per_product_node','max_abs_y_vs_cpu','max_abs_x_vs_cpu')})"
done; done
for v in 0 1; do
  rm -rf /tmp/kv_$v
  GENIE_S2_PSEG=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kv_$v -o kv -- python $R/tools/bench_variants.py subgraph > /tmp/kv.log 2>&1
  echo "== GENIE_S2_PSEG=$v kernel trace"
  python $R/tools/prof_summary.py $(find /tmp/kv_$v -name "*.db" | head -1) 12 | cut -c1-170
done
} > $OUT 2>&1
cat $OUT
