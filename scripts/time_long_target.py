"""Time a search of a few long targets (seeded in windows; lists, first cull and binding on the device, pairing on host
threads): MB megabases in N targets."""
import json, os, sys, time
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
times = []
for rep in range(int(os.environ.get("REPS", "6"))):
    t = time.perf_counter(); hits = eng.search(db); times.append(time.perf_counter() - t)
dt = sorted(times[1:])[len(times[1:]) // 2]  # median of the searches behind the first (which builds buffers and threads)
print("searches, ms: " + " ".join("%.1f" % (x * 1e3) for x in times))
tm = eng.timing()
print("cfg %d, %d assays, %d target(s) of %.1f Mb: %.1f ms per search (seed %.1f bind %.1f pair %.1f), %d hits, %.1f Mb/s" % (
    cfg, na, n, mb / n, dt * 1e3, tm.seed_ms, tm.bind_ms, tm.pair_ms, len(hits), mb / dt))
print(json.dumps({"what": "long-target path (targets seeded in windows)", "config": cfg, "assays": na, "targets": n, "megabases": mb,
                  "ms_per_search_median": round(dt * 1e3, 2), "ms_all": [round(x * 1e3, 2) for x in times], "hits": len(hits),
                  "megabases_per_s": round(mb / dt, 1)}))
