import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:40])
print({k: d["roofline"]["kernels"][k]["ms"] for k in d["roofline"]["kernels"]}, d["roofline"]["kernels"]["k_stage1"].get("traffic"), d["roofline"]["kernels"]["k_stage2"].get("traffic"))
print(d.get("sharded_workload_on_one_gpu"))
print(d.get("training_step_config3", {}).get("ms_per_step"), d.get("streaming_config5"))
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:100], d["cpu_baseline"].get("max_abs_y_vs_cpu"))
