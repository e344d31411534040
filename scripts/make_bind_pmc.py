#!/usr/bin/env python3
"""profiles/bind_pmc.json (what bench.py reads for roofline.traffic / valu_issue) from a PMC summary of
scripts/pmc_summarize.py:   python scripts/make_bind_pmc.py profiles/r02n_pmc_config2_1M_serial.json"""
import json
import os
import sys

src = sys.argv[1]
summary = json.load(open(src))
d = summary["kernels"]
stage = [k for k in ("bucket_kernel", "regroup_count_kernel", "regroup_scan_kernel", "regroup_scatter_kernel", "ufilter_kernel<L>", "filter_kernel<L>", "sbound_kernel", "bind_kernel<L>") if k in d]
seed = "seed2_kernel" if "seed2_kernel" in d else "seed_kernel"  # (round 3: tnt_seed2.hip.h)
fetch = sum(float(d[k]["FETCH_SIZE"]) for k in stage)   # KB per search (= per step of the full-size config)
write = sum(float(d[k]["WRITE_SIZE"]) for k in stage)
old = {}
dst = os.path.join(os.path.dirname(os.path.abspath(src)), "bind_pmc.json")
if os.path.exists(dst):
    old = json.load(open(dst))
out = {
    "git_head": summary.get("git_head"),
    "csrc_sha16": summary.get("csrc_sha16"),  # bench.py uses the counters only for a library built from these sources  # the build the counters were collected on (scripts/pmc_summarize.py reads .git_head)
    "source": "%s (rocprofv3 --pmc, separate passes, scripts/pmc_collect.sh 2 1000000, bind launches serialised on one stream)" % src,
    "kernel": "bind stage of one step (config 2, 1 000 000 x 1 kb targets): regroup kernels + ufilter_kernel<L> + sbound_kernel<L> (phase A) / bucket_kernel + filter_kernel<L> (phase B) + bind_kernel<L> on the survivors",
    "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
    "per_kernel_KB": {k: {"FETCH_SIZE": float(v["FETCH_SIZE"]), "WRITE_SIZE": float(v["WRITE_SIZE"])}
                      for k, v in d.items() if k in stage + [seed, "pair_kernel"]},
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled; WRITE_SIZE taken as is",
    "hbm_bytes_per_step": int((2.0 * fetch + write) * 1024),
    "SQ_INSTS_VALU_per_step": sum(float(d[k]["SQ_INSTS_VALU"]) for k in stage),
    "lds": {k: {c: float(d[k][c]) for c in ("SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE") if c in d[k]} for k in stage},
    # the seed stage's instruction counts (wave-instructions per step): what bounds it is issue, not the 250 MB it streams
    "seed_kernel_insts_per_step": {k: float(d[seed][k]) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")},
    "history": (old.get("history") or ([{"hbm_bytes_per_step": old["round1"]["hbm_bytes_per_step"], "SQ_INSTS_VALU_per_step": old["round1"]["SQ_INSTS_VALU_per_step"], "state": "round 1 (r01k)"}] if "round1" in old else []))
               + ([{"hbm_bytes_per_step": old["hbm_bytes_per_step"], "SQ_INSTS_VALU_per_step": old["SQ_INSTS_VALU_per_step"], "state": old.get("source", "")[:40]}] if "hbm_bytes_per_step" in old and old.get("source") != None and src not in old.get("source", "") else []),
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_step", "SQ_INSTS_VALU_per_step")}))
