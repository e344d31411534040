"""Replay a pickled fuzz case (exp/fuzz_case.pkl) stage by stage: seeds, binds, hits of the offending (target, assay)."""
import os, pickle, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tntcheck as tc
from thermonucleotideblast_amd import abi, synth
from thermonucleotideblast_amd.engine import Engine
d = pickle.load(open(os.path.join(ROOT, "exp", "fuzz_case.pkl"), "rb"))
opt = abi.options_default(**d["options"])
ex = [synth.ExpandedAssay(*e) for e in d["expanded"]]
h = d["only_hip"][0] if d["only_hip"] else d["only_oracle"][0]
aid, t = h[1], h[2]
e1 = [e for e in ex if e.degen_id == aid]
codes = d["codes"][int(d["offsets"][t]):int(d["offsets"][t + 1])]
offs = np.array([0, len(codes)], dtype=np.uint64)
ora = tc.OracleImpl()
eng = Engine(opt, e1)
db = eng.pack(codes, offs)
oligos = eng.oligos
so = sorted((s[0], oligos[s[1]] if isinstance(s[1], int) else s[1], s[2], s[3], s[4]) for s in ora.seeds(opt, oligos, codes, offs))
sg = sorted(eng.seeds(db))
print("seeds equal:", so == sg, len(so), len(sg))
rows = [(0, oligos.index(s[1]), s[2], s[3], s[4], 0, opt.primer_strand) for s in sg]
reqs = tc.make_reqs(rows)
bo = ora.bind_batch(opt, oligos, codes, offs, reqs)
bg = eng.bind_batch(db, tc.make_reqs(rows))
nbad = 0
for i in range(len(rows)):
    df = tc.bind_compare_fields(bo[i], bg[i])
    if df:
        nbad += 1
        print("bind differs", rows[i], df[:4])
    if bo[i].status == 0:
        print("bound", rows[i][1:5], "loc", bo[i].loc5, bo[i].loc3, "tm %.3f a5 %d a3 %d mm %d" % (bo[i].tm, bo[i].anchor5, bo[i].anchor3, bo[i].num_mm))
print("binds differing:", nbad)
print("hip hits:", [(x.amplicon_start, x.amplicon_stop, x.forward_oligo, x.reverse_oligo, x.primer_strand, x.forward_primer_clamp, x.reverse_primer_clamp) for x in eng.search(db)])
print("oracle hits:", [(x.amplicon_start, x.amplicon_stop, x.forward_oligo, x.reverse_oligo, x.primer_strand) for x in ora.search(opt, e1, codes, offs)])
