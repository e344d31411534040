"""Where a step goes outside the stage kernels: wall time of enqueue / collect per search for the three forms of collect
(records = 1464-byte hit records to the host, compact = the compact blob, device = nothing copied)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thermonucleotideblast_amd import synth
from thermonucleotideblast_amd.engine import Engine
nt = int(os.environ.get("NT", "1000000"))
wl = synth.make_workload(2, n_targets=nt)
eng = Engine(wl.options, wl.expanded)
db = eng.pack(wl.codes, wl.offsets)
for mode in ("records", "compact", "device", "compact"):
    te = tc = 0.0
    n = 12
    eng.enqueue(db)
    t_all = None
    for i in range(n):
        t0 = time.perf_counter()
        if mode == "records":
            eng.collect(copy=False)
        elif mode == "compact":
            eng.collect_compact()
        else:
            eng.collect_device()
        t1 = time.perf_counter()
        tm = eng.timing()
        if i + 1 < n:
            eng.enqueue(db)
        t2 = time.perf_counter()
        if i == 1:
            t_all = t0
        if i >= 2:
            tc += t1 - t0; te += t2 - t1
    wall = (t2 - t_all) / (n - 1) if False else None
    print("%-8s collect %.2f ms (incl. waiting for the GPU), enqueue %.2f ms, sum %.2f ms per search; stages seed %.2f bind %.2f pair %.2f = %.2f ms" % (
        mode, 1e3 * tc / (n - 2), 1e3 * te / (n - 2), 1e3 * (tc + te) / (n - 2), tm.seed_ms, tm.bind_ms, tm.pair_ms, tm.seed_ms + tm.bind_ms + tm.pair_ms))
