#!/usr/bin/env python
"""One-line digest of a bench.py JSON line (stdin): headline + the extra records."""
import json
import sys
d = json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1])
x3 = d.get("opt_in_bf16x3_inference", {})
print("rays/s", round(d["value"]), "| strong512", round(d.get("strong512", {}).get("value", 0)), "| render", round(d.get("render_fwd", {}).get("value", 0)),
      "| grid512 ms", round(d.get("grid512", {}).get("grid_ms", 0), 1), "| tt", round(d.get("tt_sh25", {}).get("value", 0)),
      "| x3 render", round(x3.get("render_fwd_rays_per_s", 0)), "x3 grid ms", round(x3.get("grid512_ms", 0), 1))
