#!/usr/bin/env python3
"""Fold the csv files of scripts/pmc_collect.sh into one json: per kernel (template instances of the bind
kernel folded together) the counter sums per search, dispatch counts and average durations.

  python scripts/pmc_summarize.py gpurun_out/pmc profiles/r01d_pmc_config2.json [searches]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\[clone.*", "", name).strip()
    m = re.search(r"wbind_kernel<(\d+), *(true|false)", name)  # the wavefront-per-window form: rows per lane
    if m:
        return "wbind_kernel<%s>" % ("unit" if m.group(2) == "true" else "R")
    m = re.search(r"(bind_kernel)<(\d+), *(true|false)", name)
    if m:
        return "bind_kernel<%s>" % ("unit" if m.group(3) == "true" else "L")
    if "ufilter_kernel<" in name:  # the pre-filter on queues regrouped by oligo (round 5)
        return "ufilter_kernel<L>"
    if "filter_kernel<" in name:
        return "filter_kernel<L>"
    m = re.search(r"(\w+_kernel)\b", name)
    return m.group(1) if m else name.split("(")[0]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    searches = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0  # time_variants.py runs 3 searches
    counters = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            counters[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add((os.path.basename(f), r["Dispatch_Id"]))
    dur = defaultdict(list)
    per_inst = defaultdict(list)
    per_filt = defaultdict(list)
    for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            dur[short(r["Kernel_Name"])].append(d)
            m = re.search(r"bind_kernel<(\d+), *false", r["Kernel_Name"])
            if m:
                per_inst[int(m.group(1))].append(d)
            m = re.search(r"filter_kernel<(\d+)", r["Kernel_Name"])
            if m:
                per_filt[int(m.group(1))].append(d)
    out = {"searches": searches, "kernels": {}}
    head = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".git_head")  # written before the snapshot leaves for the GPU box
    if os.path.exists(head):
        out["git_head"] = open(head).read().strip()
    # the hash of the sources the measured library was built from: bench.py only uses counters that carry its own build's hash
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["csrc_sha16"] = bench.csrc_hash()
    for k in sorted(set(counters) | set(dur)):
        e = {c: v / searches for c, v in sorted(counters[k].items())}
        if dur[k]:
            e["dispatches_per_search"] = len(dur[k]) / searches
            e["ms_per_search"] = sum(dur[k]) / searches
            e["avg_ms_per_dispatch"] = sum(dur[k]) / len(dur[k])
        out["kernels"][k] = e
    if per_inst:
        out["bind_instances_ms_per_search"] = {str(l): sum(v) / searches for l, v in sorted(per_inst.items())}
    if per_filt:
        out["filter_instances_ms_per_search"] = {str(l): sum(v) / searches for l, v in sorted(per_filt.items())}
    json.dump(out, open(dst, "w"), indent=1)
    for k, e in out["kernels"].items():
        print(k, {c: (round(v, 3) if v < 1e4 else "%.4g" % v) for c, v in e.items()})


if __name__ == "__main__":
    main()
