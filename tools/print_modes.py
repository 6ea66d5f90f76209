"""Condense a bench.py JSON line: python tools/print_modes.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"], 1), d["unit"], "| ms/step", round(d["ms_per_step"], 4))
print({k: (round(v["it_per_s"]), round(v["ms_per_step"], 4)) for k, v in d["modes"].items()})
r = d.get("roofline")
if r:
    print("roofline frac", round(r["frac"], 3), "launch", round(r["avg_launch_ms"] * 1e3, 1), "us | stats", round(r["stats_launches_ms"] * 1e3, 1), "| walk",
          round(r["resample_launch_ms"] * 1e3, 1), "| single", round(r["single_image"]["frac"], 3), "| operator", round(r["operator_face"]["frac"], 3),
          "| env_prt", round(r["env_prt"]["frac"], 3) if r.get("env_prt") else None)
