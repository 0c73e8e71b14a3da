#!/bin/bash
# needs: python -c "from genie_amd import _lib; _lib.build(extra_flags=['-DGENIE_TUNING=1','-DGENIE_ABL_MFMA=0'], out_path='genie_amd/lib/variants/libgenie_tune.so')"
# Round 6, VERDICT item 4: pass 1' of the training backward as ONE wave per tile (k_train_b1: 434 registers, one wave per SIMD) against
# TWO waves per tile (k_train_b1s: 256 registers, two waves per SIMD). Same box, tuning build (GENIE_B1_SPLIT=0 / 1 selects the kernel):
# step time + phases (bench --mode train), kernel-trace averages, SQ counters of both. Output: gpurun_out/r06_b1_split_ab.txt
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r06_b1_split_ab.txt
export GENIE_LIB_PATH=$R/genie_amd/lib/variants/libgenie_tune.so
cd /tmp && export TMPDIR=/tmp
{
python $R/tools/kernel_resources.py $GENIE_LIB_PATH k_train_b1s k_train_b1ILb0ELb1 k_train_b1ILb1ELb1
for rep in 1 2; do
for v in 0 1; do
  echo "== GENIE_B1_SPLIT=$v (run $rep): ms_per_step, phases, four-output step"
  GENIE_B1_SPLIT=$v timeout 300 python $R/bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['phase_ms'], d['four_output_step']['ms_per_step'])"
done
done
for v in 0 1; do
  rm -rf /tmp/rt_$v /tmp/pmc_$v
  GENIE_B1_SPLIT=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rt_$v -o rt -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > /tmp/rt.log 2>&1
  echo "== GENIE_B1_SPLIT=$v kernel trace"
  python $R/tools/prof_summary.py $(find /tmp/rt_$v -name "*.db" | head -1) 40 | grep "k_train_b1" | cut -c1-170
  GENIE_B1_SPLIT=$v timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc_$v -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > /tmp/pmc.log 2>&1 || echo "pmc pass failed"
  echo "== GENIE_B1_SPLIT=$v counters (quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES, per launch, summed over the chip)"
  python $R/tools/pmc_summary.py /tmp/pmc_$v k_train_b1
done
} > $OUT 2>&1
cat $OUT
