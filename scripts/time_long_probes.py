"""Time a search with long hybridisation probes (65-128 bases): NP probes of LEN bases against NT random 1-kb targets,
a few of them with a planted site; minimum probe Tm from MIN_TM.  TNT_MI355X_NO_FILTER=1 for the comparison."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thermonucleotideblast_amd import abi, synth
from thermonucleotideblast_amd.engine import Engine
nt = int(os.environ.get("NT", "100000")); npb = int(os.environ.get("NP", "96")); ln = int(os.environ.get("LEN", "100"))
rng = np.random.default_rng(12)
probes = [synth._rand_oligo(rng, ln, ln) for _ in range(npb)]
codes = rng.integers(0, 4, nt * 1000).astype(np.uint8)
offsets = (np.arange(nt + 1) * 1000).astype(np.uint64)
for t in range(0, nt, max(1, nt // 200)):  # planted sites: a probe's reverse complement with two mismatches
    site = list(synth.revcomp(probes[t % npb])); site[10] = "A"; site[50] = "C"
    codes[t * 1000 + 300:t * 1000 + 300 + ln] = synth.encode("".join(site))
opt = abi.options_default(min_primer_tm=40.0, min_probe_tm=float(os.environ.get("MIN_TM", "60")))
opt.assay_format = abi.ASSAY_PROBE
eng = Engine(opt, [synth.ExpandedAssay(i, i, None, None, p) for i, p in enumerate(probes)])
db = eng.pack(codes, offsets)
times = []
for rep in range(4):
    t0 = time.perf_counter(); hits = eng.search(db); times.append(time.perf_counter() - t0)
tm = eng.timing()
print("%d probes of %d bases x %d targets, -E %s%s: %s ms per search; seed %.1f bind stage %.1f pair %.1f ms; windows %d, rejected by the pre-filter %d, bound %d, hits %d" % (
    npb, ln, nt, os.environ.get("MIN_TM", "60"), " (no pre-filter)" if os.environ.get("TNT_MI355X_NO_FILTER") else "", " ".join("%.1f" % (x * 1e3) for x in times),
    tm.seed_ms, tm.bind_ms, tm.pair_ms, tm.n_binds, tm.n_filtered, tm.n_bound, len(hits)))
