#!/usr/bin/env python3
"""The figures of a bench line a reader looks for first.  usage: show_bench.py <bench.json>"""
import json
import sys
d = json.load(open(sys.argv[1]))
print("value %.4g %s  ms_per_step %.4f  roofline.frac %.3f (avg launch %.4f ms)  value_strict %.4g" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("avg_launch_ms", 0), d.get("value_strict", 0)))
print("kernels_ms_per_step", d.get("kernels_ms_per_step"), " cold", d.get("cold", {}).get("value"))
if "pcie_inclusive" in d: print("pcie_inclusive %.4g" % d["pcie_inclusive"]["value"], d["pcie_inclusive"].get("sub_batches"))
if "api_inclusive" in d: print("api_inclusive %.4g  frac_of_value %s" % (d["api_inclusive"]["value"], d["api_inclusive"].get("frac_of_value")))
if "cpu_baseline" in d: print("cpu_baseline %s %s on %s cores" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], d["cpu_baseline"]["cores"]))
for k, v in d.get("next_rows", {}).get("categorical_bootstrap", {}).items():
    print("categorical", k, v.get("replicates_per_s"), v.get("ms_per_step"), v.get("kernel_ms_per_step"), v.get("error", ""))

for k, v in d.get("next_rows", {}).get("metric_models_next_to_the_headline", {}).items():
    if isinstance(v, dict): print("metric", k, v.get("replicates_per_s"), v.get("ms_per_step"), v.get("kernels_ms"), v.get("solver_kernel"))
