#!/usr/bin/env python3
"""Summarise rocprofv3 outputs under a directory tree into one text + traffic.json.

usage: pmc_summary.py <gpurun_out/prof_dir> <workload> <out_prefix>
Reads every */*_counter_collection.csv and */*_kernel_stats.csv below the directory,
averages counters over the dispatches of kernels whose name contains 'k_render'.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  Per MI355X_MICROARCH.md
(HBM section) FETCH_SIZE under-counts wide coalesced streams by 2x on gfx950 and is
uncalibrated for other access widths; this kernel's reads are 8-byte gathers, so the
raw figure is kept and the x2 bound is listed next to it."""
import collections, csv, glob, json, os, sys

root, workload, out_prefix = sys.argv[1], sys.argv[2], sys.argv[3]
counters = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"] and "Lb1E" not in r["Kernel_Name"].split("k_render")[1][:14]:
            per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, name), v in per_dispatch.items():
        counters[name].append(v)
lines = []
avg = {k: sum(v) / len(v) for k, v in counters.items()}
for k in sorted(avg):
    lines.append(f"{k:24s} {avg[k]:18.1f}  (mean of {len(counters[k])} dispatches)")
stats = []
for f in glob.glob(os.path.join(root, "**", "*_kernel_stats.csv"), recursive=True):
    stats.append(open(f).read())
text = "\n".join(lines) + "\n\n" + "\n".join(stats)
open(out_prefix + ".txt", "w").write(text)
print(text)
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    fetch, write = avg["FETCH_SIZE"] * 1024.0, avg["WRITE_SIZE"] * 1024.0
    tj = os.path.join(os.path.dirname(out_prefix), "traffic.json")
    data = json.load(open(tj)) if os.path.exists(tj) else {}
    data[workload] = {"hbm_bytes_per_launch": fetch + write, "fetch_bytes_raw": fetch, "write_bytes": write,
                      "fetch_bytes_if_undercounted_2x": 2 * fetch,
                      "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean per k_render launch; "
                              "FETCH_SIZE kept raw: reads are 8-byte gathers (uncalibrated width on gfx950)"}
    json.dump(data, open(tj, "w"), indent=1, sort_keys=True)
