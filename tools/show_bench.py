import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("literal call ms", d["ms_per_step"], "picks/s", d["value"], "frac", r["frac"], "| pipelined ms", d.get("pipelined_windows_ms"), "frac", r.get("pipelined_frac"))
print("fused_bytes_frac", r.get("fused_bytes_frac"), "hbm_real_frac", r.get("hbm_real_frac"), "traffic", r["traffic"], r["traffic_source"][:40])
print({k: r["kernels"][k]["ms"] for k in r["kernels"]}, r["kernels"]["k_stage1"].get("traffic"), r["kernels"]["k_stage2"].get("traffic"))
print("cfg4 one gpu:", {k: v for k, v in (d.get("sharded_workload_on_one_gpu") or {}).items() if k in ("ms_per_step", "rank0_phase_ms_sequential", "error")})
t3 = d.get("training_step_config3", {})
print("train:", t3.get("ms_per_step"), t3.get("phase_ms"), "| four:", (t3.get("four_output_step") or {}).get("ms_per_step"), "| rebuild:", t3.get("four_output_step_new_graph_per_sample"))
print("stream:", d.get("streaming_config5"))
print("day loops:", d.get("day_loops_config2"))
c = d.get("cpu_baseline", {})
print("cpu:", c.get("value"), c.get("max_abs_y_vs_cpu"), c.get("max_abs_x_vs_cpu"), c.get("mask_mean_of_that_window"), c.get("sparse_window"))
