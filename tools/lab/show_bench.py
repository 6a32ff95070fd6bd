#!/usr/bin/env python3
"""Print the key numbers of a bench.py JSON line: train / forward clips/s, roofline, secondary lines."""
import json
import sys

d = json.load(open(sys.argv[1]))
print("train", d["value"], "clips/s", d["ms_per_step"], "ms | fwd", d.get("forward_only", {}).get("value"), "| frac", d["roofline"]["frac"],
      d["roofline"]["avg_launch_ms"], "ms | secondary", [x["value"] for x in d.get("secondary", [])], "| host", d["host_enqueue_ms_per_step"], d["host_cpu_ms_per_step"], "median", d.get("host_enqueue_ms_median"), "stall", d.get("host_queue_stall_ms_total"))
