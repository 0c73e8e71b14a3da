#!/bin/bash
# needs: python -c "from genie_amd import _lib; _lib.build(extra_flags=['-DGENIE_H2_THREADS=1024'], out_path='genie_amd/lib/variants/libgenie_h2t1024.so')"
# Round 6, VERDICT item 6: k_stage1_h2 with one 8-wave workgroup per CU (production: 256 unified registers per wave, two waves per SIMD)
# against one 16-wave workgroup per CU sharing the weight image (GENIE_H2_THREADS=1024: the compiler is held to 128 registers).
# Same box: HIP-event time of stage 1 (tools/s1_time.py) + SQ counters of both. Output: gpurun_out/r06_s1_occupancy_ab.txt
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r06_s1_occupancy_ab.txt
V=$R/genie_amd/lib/variants/libgenie_h2t1024.so
{
echo "== kernel resources (tools/kernel_resources.py: unified VGPRs, LDS bytes, spilled VGPRs, workgroup size)"
python $R/tools/kernel_resources.py ILi8ELi15ELb0ELb0ELb0ELb0
python $R/tools/kernel_resources.py $V ILi8ELi15ELb0ELb0ELb0ELb0
echo "== HIP-event time of stage 1 (split pass + k_stage1_h2), config 2, clocks settled, A B A B"
for i in 1 2; do
  python $R/tools/s1_time.py cfg2_200x10k 60
  GENIE_LIB_PATH=$V python $R/tools/s1_time.py cfg2_200x10k 60
done
cd /tmp && export TMPDIR=/tmp
for tag in wg512 wg1024; do
  if [ $tag = wg1024 ]; then export GENIE_LIB_PATH=$V; fi
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$tag -- python $R/tools/stage_profile.py cfg2_200x10k 3 > /tmp/pmc_$tag.log 2>&1 || echo "pmc pass $tag failed"
  echo "== counters, $tag (per launch, summed over the chip)"
  python $R/tools/pmc_summary.py /tmp/pmc_$tag k_stage1_h2
done
} > $OUT 2>&1
cat $OUT
