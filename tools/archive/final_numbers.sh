#!/bin/bash
# end-of-round numbers of the tree (profiles/r04_p_*): model / graph variants single stream and with batched tails, training bench
timeout 900 python tools/bench_variants.py default edges abspos subgraph > gpurun_out/r04_p_variants.json 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r04_p_variants.json"):
    d = json.loads(l); print(d["variant"], d["ms_per_window"], d["ms_per_window_batched_tails"], d["batched_equals_single_stream_bitwise"], d["max_abs_y_vs_cpu"], d["max_abs_x_vs_cpu"])
PY
timeout 600 python bench.py --mode train > gpurun_out/r04_p_bench_train.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_p_bench_train.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["phase_ms"], d["four_output_step"]["ms_per_step"])
PY
