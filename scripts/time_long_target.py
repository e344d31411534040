"""Time a search of a few long targets (seeded in windows, culled and paired on the host): MB megabases in N targets."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thermonucleotideblast_amd import synth
from thermonucleotideblast_amd.engine import Engine
mb = float(os.environ.get("MB", "5")); n = int(os.environ.get("N", "1")); cfg = int(os.environ.get("CFG", "2")); na = int(os.environ.get("NA", "96"))
wl = synth.make_workload(cfg, n_targets=int(mb * 1000), n_assays=na)
per = (len(wl.offsets) - 1) // n
offsets = np.array([wl.offsets[i * per] for i in range(n)] + [wl.offsets[-1]], dtype=np.uint64)
eng = Engine(wl.options, wl.expanded)
db = eng.pack(wl.codes, offsets)
for rep in range(2):
    t = time.perf_counter(); hits = eng.search(db); dt = time.perf_counter() - t
tm = eng.timing()
print("cfg %d, %d assays, %d target(s) of %.1f Mb: %.1f ms per search (seed %.1f bind %.1f pair %.1f), %d hits, %.1f Mb/s" % (
    cfg, na, n, mb / n, dt * 1e3, tm.seed_ms, tm.bind_ms, tm.pair_ms, len(hits), mb / dt))
