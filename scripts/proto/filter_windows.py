# windows + truth for config-2-like seeds
import sys, pickle
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import tntcheck as tc
from thermonucleotideblast_amd import synth, abi
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
wl = synth.make_workload(cfg, n_targets=nt)
ora = tc.OracleImpl()
oligos = sorted({o for e in wl.expanded for o in (e.forward, e.reverse, e.probe) if o})
sd = ora.seeds(wl.options, oligos, wl.codes, wl.offsets)
print("seeds", len(sd))
reqs = tc.make_reqs([(s[0], s[1], s[2], s[3], s[4], 0, 9e-7) for s in sd])
out = ora.bind_batch(wl.options, oligos, wl.codes, wl.offsets, reqs)
rec = []
comp = {0:3,1:2,2:1,3:0}
Q = []; T = []
for s, b in zip(sd, out):
    q = oligos[s[1]]
    L = len(q)
    t = s[0]
    seq = wl.codes[int(wl.offsets[t]):int(wl.offsets[t+1])]
    w = seq[b.target_start:b.target_stop]
    if (w > 3).any():
        continue
    minus = s[2] == abi.MINUS
    tw = np.array([comp[int(x)] for x in w[::-1]], dtype=np.int8) if minus else w.astype(np.int8)
    qq = np.array(["ACGT".index(c) for c in q], dtype=np.int8)
    Q.append(qq); T.append(tw)
    rec.append((b.status, b.tm, b.dH, b.dS, b.valid, b.dp_max, b.num_gap, b.num_mm))
pickle.dump((Q, T, rec, wl.options.target_t, wl.options.salt), open("/tmp/proto/win%d.pkl" % cfg, "wb"))
st = np.array([r[0] for r in rec]); print("n", len(rec), "bound", (st == 0).sum(), "frac", (st == 0).mean())
