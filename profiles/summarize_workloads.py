#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/<workload>/ (written on the GPU box by profiles/collect_workloads.sh)
into the committed summary profiles/<tag>_workloads.json: per workload the rocprofv3 --kernel-trace
--stats line of its dominant kernel and the mean per launch of every --pmc counter collected for
it (separate passes), plus the ratios DESIGN.md quotes.  Also refreshes
profiles/pmc_traffic_workloads.json (HBM bytes per launch per kernel: WRITE_SIZE + 2 x FETCH_SIZE,
the gfx950 correction of MI355X_MICROARCH.md), which bench.py reads for `roofline.traffic`.

  python profiles/summarize_workloads.py r02a [workload ...]
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_OF = {  # substring of the dominant kernel's name per workload
    "perlin": "perlin_grid", "turb7": "perlin_grid", "multiband5": "grid3d_mbp_kernel<5",
    "texture_points": "row_slab_points_kernel", "texture_points_perlin": "noise_texture_kernel",
    "wavelet3d": "grid3d_mbp_kernel<1", "wavelet3d_exact": "grid3d_exact_lds_kernel",
    "wavelet3d_1024": "grid3d_mbp_kernel<1", "wavelet3d_2048x2048x256": "grid3d_mbp_kernel<1",
    "wavelet3d_512x512x64": "grid3d_strip_kernel",
}


# a second kernel of the same call whose traffic belongs to the call (texture_points: the plane-ordered pass that keeps or marks
# every chunk before the row-slab kernel runs)
COMPANION_OF = {"texture_points": "plane_sorted_points_kernel"}


def summarize(tag, workloads):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    doc = {"tag": tag,
           "commands": {"trace": "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload W --steps 10 --warmup 2 --no-cpu-baseline",
                        "pmc": "rocprofv3 --pmc <one group per pass> -- python3 bench.py --workload W --steps 4 --warmup 1 --no-cpu-baseline"},
           "units": "SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over the chip; WRITE_SIZE / FETCH_SIZE are KiB "
                    "(FETCH_SIZE reads half the bytes of a wide read stream on gfx950: doubled in hbm_bytes)",
           "workloads": {}}
    traffic = []
    for wl in workloads or sorted(os.listdir(src)):
        d = os.path.join(src, wl)
        if not os.path.isdir(d) or wl not in KERNEL_OF:
            continue
        key = KERNEL_OF[wl]
        entry = {"kernel_match": key}
        for f in glob.glob(os.path.join(d, "trace", "*", "*kernel_stats.csv")):
            for r in csv.DictReader(open(f)):
                if key in r["Name"]:
                    entry["kernel"] = r["Name"]
                    entry["trace"] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                      "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
        counters = {}
        companion = collections.defaultdict(list)
        for f in sorted(glob.glob(os.path.join(d, "pmc_*", "*", "*counter_collection.csv"))):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if wl in COMPANION_OF and COMPANION_OF[wl] in r["Kernel_Name"] and r["Counter_Name"] in ("WRITE_SIZE", "FETCH_SIZE"):
                    companion[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if key in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    for col in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
                        if col in r:
                            entry.setdefault("dispatch", {})[col] = int(float(r[col]))
            for k, v in agg.items():
                counters[k] = round(sum(v) / len(v), 1)
        entry["counters_mean_per_launch"] = counters
        c = counters
        derived = {}
        f64 = c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_FMA_F64", 0)
        if c.get("SQ_INSTS_VALU"):
            derived["fp64_share_of_issued_valu"] = round(f64 / c["SQ_INSTS_VALU"], 3)
        if c.get("SQ_WAVE_CYCLES"):
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS"):
                if k in c:
                    derived[k.lower() + "_share_of_wave_cycles"] = round(c[k] / c["SQ_WAVE_CYCLES"], 3)
        if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            derived["l1_miss_share_(tcc_read_req/cache_accesses)"] = round(c.get("TCP_TCC_READ_REQ_sum", 0) / c["TCP_TOTAL_CACHE_ACCESSES_sum"], 3)
        if "WRITE_SIZE" in c and "FETCH_SIZE" in c:
            derived["hbm_bytes"] = c["WRITE_SIZE"] * 1024 + 2 * c["FETCH_SIZE"] * 1024
            if companion.get("WRITE_SIZE") and companion.get("FETCH_SIZE"):
                cw, cf = (sum(companion[k]) / len(companion[k]) for k in ("WRITE_SIZE", "FETCH_SIZE"))
                entry["companion"] = {"kernel_match": COMPANION_OF[wl], "WRITE_SIZE": round(cw, 1), "FETCH_SIZE": round(cf, 1),
                                      "hbm_bytes": cw * 1024 + 2 * cf * 1024}
                derived["hbm_bytes_with_companion"] = derived["hbm_bytes"] + entry["companion"]["hbm_bytes"]
            traffic.append({"kernel": key, "workload": wl, "write_bytes": c["WRITE_SIZE"] * 1024,
                            "fetch_bytes_corrected": 2 * c["FETCH_SIZE"] * 1024,
                            "bytes_per_launch": derived.get("hbm_bytes_with_companion", derived["hbm_bytes"]),
                            "source": f"profiles/{tag}_workloads.json (separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md)"})
        if c.get("GRBM_GUI_ACTIVE") and entry.get("trace"):
            derived["effective_clock_GHz"] = round(c["GRBM_GUI_ACTIVE"] / 8 / (entry["trace"]["avg_us"] * 1e3), 3)
        entry["derived"] = derived
        doc["workloads"][wl] = entry
    out = os.path.join(ROOT, "profiles", f"{tag}_workloads.json")
    json.dump(doc, open(out, "w"), indent=1)
    if traffic:
        p = os.path.join(ROOT, "profiles", "pmc_traffic_workloads.json")
        old = []
        if os.path.exists(p):
            try:
                old = [e for e in json.load(open(p)) if e.get("workload") not in {t["workload"] for t in traffic}]
            except Exception:  # noqa: BLE001
                old = []
        json.dump(old + traffic, open(p, "w"), indent=1)
    for wl, e in doc["workloads"].items():
        print(wl, e.get("trace"), e["derived"])
    return out


if __name__ == "__main__":
    summarize(sys.argv[1] if len(sys.argv) > 1 else "r02a", sys.argv[2:])
