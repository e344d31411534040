#!/usr/bin/env python3
"""GPU box: rows of tnt_first_cull from the round-3 seed kernel, the round-2 one and the oracle, side by side for the
(target, assay) groups that differ."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tntcheck as tc  # noqa: E402
from thermonucleotideblast_amd import synth  # noqa: E402
from thermonucleotideblast_amd.engine import Engine  # noqa: E402

cfg, nt, na, seed = (int(x) for x in (sys.argv[1:5] + ["2", "200", "48", "31"][len(sys.argv) - 1:]))
wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=seed)
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    setattr(wl.options, k, int(v))
res = {}
for name in ("v3", "v2"):
    if name == "v2":
        os.environ["TNT_MI355X_SEED_V2"] = "1"
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    res[name] = sorted(tuple(int(x) for x in r) for r in e.first_cull(db))
    db.close()
    e.close()
res["oracle"] = sorted(tuple(int(x) for x in r) for r in tc.OracleImpl().first_cull(wl.options, wl.expanded, wl.codes, wl.offsets))
print({k: len(v) for k, v in res.items()}, "v2 == oracle:", res["v2"] == res["oracle"], "v3 == oracle:", res["v3"] == res["oracle"])
bad = sorted({(r[0], r[1]) for r in set(res["v3"]) ^ set(res["oracle"])})
print(len(bad), "groups differ; first:", bad[:10])
for g in bad[:4]:
    for name in ("v3", "oracle"):
        print(name, g, [r[2:] for r in res[name] if (r[0], r[1]) == g])
