"""Time the stages of one search for the library named by TNT_MI355X_LIB (kernel experiments)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thermonucleotideblast_amd import synth
from thermonucleotideblast_amd.engine import Engine
cfg = int(os.environ.get("CFG", "2")); nt = int(os.environ.get("NT", "100000"))
na = os.environ.get("NA")
wl = synth.make_workload(cfg, n_targets=nt, n_assays=int(na) if na else None)
for kv in filter(None, os.environ.get("OPTS", "").split(",")):  # option overrides, e.g. OPTS=min_primer_tm=0,max_primer_dg=-8
    k, v = kv.split("=")
    setattr(wl.options, k, type(getattr(wl.options, k))(float(v)))
eng = Engine(wl.options, wl.expanded)
db = eng.pack(wl.codes, wl.offsets)
seed = []
for rep in range(int(os.environ.get("REPS", "3"))):
    t = time.perf_counter(); hits = eng.search(db); dt = time.perf_counter() - t
    tm = eng.timing()
    seed.append(tm.seed_ms)
print("%s %s cfg %d nt %d: wall %.1f ms seed %.2f (min %.2f) bind %.2f pair %.2f | binds %d filtered %d groups %d bound %d hits %d" % (
    os.path.basename(os.environ.get("TNT_MI355X_LIB", "default")), os.environ.get("OPTS", ""), cfg, nt, dt * 1e3, tm.seed_ms, min(seed[1:] or seed), tm.bind_ms, tm.pair_ms, tm.n_binds, tm.n_filtered, tm.n_seeds, tm.n_bound, len(hits)))
