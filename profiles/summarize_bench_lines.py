#!/usr/bin/env python3
"""gpurun_out/bench_<tag>/*.json (profiles/run_bench_lines.sh on the GPU box) -> profiles/<tag>_bench_lines.json."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
lines = {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"bench_{tag}", "*.json"))):
    txt = open(f).read().strip()
    if txt:
        try:
            lines[os.path.basename(f)[:-5]] = json.loads(txt.splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            lines[os.path.basename(f)[:-5]] = {"error": str(e)}
json.dump(lines, open(os.path.join(ROOT, "profiles", f"{tag}_bench_lines.json"), "w"), indent=1)
for k, v in lines.items():
    if "value" in v:
        r = v["roofline"]
        print(f"{k:30s} {v['value']:12.1f} {v['unit']}  launch {r['avg_launch_us']:9.1f} us  {r['bound']} frac {r['frac']:.3f}")
