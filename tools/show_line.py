#!/usr/bin/env python3
"""Print the headline of a bench.py JSON line: python tools/show_line.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"),
      r.get("traffic_frac"))
for s in d.get("secondary", []):
    print(" ", s["workload"], round(s.get("value", 0)), s.get("frac"),
          s.get("error"))
