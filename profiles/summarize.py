#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by profiles/collect.sh on the GPU box) into the committed
summaries: kernel stats CSV, PMC means per launch, pmc_traffic.json (read by bench.py) and the
bench lines."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "grid3d_strip_kernel"  # the kernel bench.py's default workload launches

for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    shutil.copyfile(f, os.path.join(dst, f"{tag}_kernel_stats_bench_wavelet3d.csv"))
counters = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        counters[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
json.dump({"kernel": KERNEL, "command": "rocprofv3 --pmc <one group per pass> -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline",
           "counters": counters}, open(os.path.join(dst, f"{tag}_pmc_{KERNEL}.json"), "w"), indent=1)
if "WRITE_SIZE" in counters and "FETCH_SIZE" in counters:
    # MI355X_MICROARCH.md (HBM): WRITE_SIZE and FETCH_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the
    # bytes of a wide coalesced read stream -> doubled before it is compared with a byte count.
    w = counters["WRITE_SIZE"]["mean_per_launch"] * 1024
    r = counters["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
    json.dump({"kernel": KERNEL, "write_bytes": w, "fetch_bytes_corrected": r, "bytes_per_launch": w + r,
               "source": f"profiles/{tag}_pmc_{KERNEL}.json (separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md)"},
              open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
lines = {}
for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    txt = open(f).read().strip()
    if txt:
        try:
            lines[os.path.basename(f)[6:-5]] = json.loads(txt.splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            lines[os.path.basename(f)[6:-5]] = {"error": str(e)}
json.dump(lines, open(os.path.join(dst, f"{tag}_bench_lines.json"), "w"), indent=1)
for k, v in lines.items():
    if "value" in v:
        print(f"{k:18s} {v['value']:12.1f} {v['unit']}  launch {v['roofline']['avg_launch_us']:9.1f} us  frac {v['roofline']['frac']:.3f}")
print(json.dumps(counters.get("WRITE_SIZE")), json.dumps(counters.get("FETCH_SIZE")))
