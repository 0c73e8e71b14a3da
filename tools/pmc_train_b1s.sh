#!/bin/bash
# SQ counters of the production k_train_b1s / k_train_b0 / k_train_b2 (bench --mode train, 4 steps). Output: gpurun_out/r06_m_pmc_train_front.txt
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_b1s /tmp/pmc_b1s2
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc_b1s -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > /tmp/pmc.log 2>&1 || echo "pmc pass 1 failed"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_b1s2 -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > /tmp/pmc.log 2>&1 || echo "pmc pass 2 failed"
{
echo "# production library, bench.py --mode train; SQ_* in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles) and instruction counts; per launch, summed over the chip"
python $R/tools/pmc_summary.py /tmp/pmc_b1s k_train_b1s k_train_b0 k_train_b2
python $R/tools/pmc_summary.py /tmp/pmc_b1s2 k_train_b1s k_train_b0 k_train_b2
} > $R/gpurun_out/r06_m_pmc_train_front.txt 2>&1
cat $R/gpurun_out/r06_m_pmc_train_front.txt
