#!/bin/bash
# Same-box A/B of bench.py under environment-variable variants: bash tools/bench_ab.sh "VAR=1" "VAR=2 OTHER=3" ...
# prints ms_per_step and the HIP-event stage times of every variant (box-to-box spread is +-3 %: only same-box pairs compare)
for v in "$@"; do
  r=$(env $v python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-cfg4-one-gpu --no-live-traffic --no-train-step --no-stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print(d['ms_per_step'], k['k_stage1']['ms'], k['k_stage2']['ms'])")
  echo "$v: $r"
done
