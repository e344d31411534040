import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import tntcheck as tc
from thermonucleotideblast_amd import synth
from thermonucleotideblast_amd.engine import Engine
wl = synth.make_workload(2, n_targets=120, n_assays=12, seed=5)
k = 60
offsets = np.concatenate([[0], wl.offsets[k:]]).astype(np.uint64)
e = Engine(wl.options, wl.expanded)
db = e.pack(wl.codes, offsets)
try:
    got = tc.sorted_hit_keys(e.search(db))
    want = tc.sorted_hit_keys(tc.OracleImpl().search(wl.options, wl.expanded, wl.codes, offsets))
    print("hits", len(got), len(want), got == want)
except Exception as ex:
    print("ERR", ex)
