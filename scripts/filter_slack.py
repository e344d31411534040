"""How tight is the pre-filter's bound on the windows it lets through?  Config 2: the seeds the first cull keeps, their bound
and threshold (tnt_filter_bound), and the status the bind kernel gives the survivors: a histogram of (bound - threshold) for the
survivors that bind and for those that do not."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from thermonucleotideblast_amd import synth
from thermonucleotideblast_amd.engine import Engine
import tntcheck as tc
nt = int(os.environ.get("NT", "20000"))
wl = synth.make_workload(2, n_targets=nt)
e = Engine(wl.options, wl.expanded)
db = e.pack(wl.codes, wl.offsets)
rows = e.first_cull(db)
om = {o: i for i, o in enumerate(e.oligos)}
ex = {(x.id if hasattr(x, "id") else i): x for i, x in enumerate(wl.expanded)}
reqs = []
for r in rows:
    a = wl.expanded[int(r[1])]
    o = a.forward if r[2] == 0 else (a.reverse if r[2] == 1 else a.probe)
    reqs.append((int(r[0]), om[o], int(r[3]), int(r[4]), int(r[5]), 0 if r[2] < 2 else 1, wl.options.primer_strand if r[2] < 2 else wl.options.probe_strand))
arr = tc.make_reqs(reqs)
fb = e.filter_bound(db, arr)
surv = np.nonzero(fb[:, 0] >= fb[:, 1])[0]
print("windows %d, survivors %d (%.2f %%)" % (len(reqs), len(surv), 100.0 * len(surv) / len(reqs)))
sub = tc.make_reqs([reqs[i] for i in surv])
out = e.bind_batch(db, sub, route=0)
status = np.array([o.status for o in out])
slack = (fb[surv, 0] - fb[surv, 1]) / 1000.0
edges = [0, 0.25, 0.5, 1, 1.5, 2, 3, 4, 6, 9, 99]
print("slack (kcal/mol)   bound   not bound (status 1 = Tm)   other status")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (slack >= lo) & (slack < hi)
    print("%5.2f - %5.2f   %7d   %9d   %9d" % (lo, hi, int(((status == 0) & m).sum()), int(((status == 1) & m).sum()), int(((status > 1) & m).sum())))
tm = np.array([o.tm for o in out])
nb = status == 1
print("not bound for Tm: Tm quantiles (threshold %.1f):" % wl.options.min_primer_tm, np.round(np.quantile(tm[nb], [0.05, 0.25, 0.5, 0.75, 0.95]), 1))
