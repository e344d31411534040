"""GPU check of the oligo self-structure entry (a17) against the oracle and, when present, the reference."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tntcheck as tc
from thermonucleotideblast_amd import abi, synth
from thermonucleotideblast_amd.engine import Engine
ora = C.CDLL(tc.ORACLE_SO)
rng = np.random.default_rng(3)
oligos = [synth._rand_oligo(rng, 15, 40) for _ in range(150)] + ["GGAGGATCCAACAGCAAGG","TTCCCCTTTGGAGGCATC","CGATGGGCTTTCAGGACAGGTGT","GCGCGCGCGCGCGCGC","ATATATATATATATAT","AAAAAAAATTTTTTTT","GGGGGAAAACCCCC","ACGTIACGTTGCAIGGATCCA","GCGAAGC","GCGAAAGC","ACGT","ACGTA","CGCGAATTCGCG","GGGCGAAGCCC","GGGCGAAAGCCC","CCCTTTTGGG"]
for _ in range(100):
    stem = synth._rand_oligo(rng, 4, 9); loop = "".join("ACGT"[i] for i in rng.integers(0,4,int(rng.integers(3,8))))
    oligos.append("".join("ACGT"[i] for i in rng.integers(0,4,int(rng.integers(0,6)))) + stem + loop + synth.revcomp(stem) + "".join("ACGT"[i] for i in rng.integers(0,4,int(rng.integers(0,5)))))
bad = 0; nz = 0
for dangle in (0, 1):
    opt = abi.options_default(allow_dangle_5=dangle, allow_dangle_3=dangle)
    eng = Engine(opt, [synth.ExpandedAssay(0, 0, None, None, "ACGTACGTACGTACGTAC")])
    for i, o in enumerate(oligos):
        p = oligos[(i * 7 + 3) % len(oligos)]
        g = eng.oligo_structure(o, p, 9e-7, 4e-7)
        h, d, x = C.c_float(), C.c_float(), C.c_float()
        assert ora.tntor_oligo_structure(C.byref(opt), o.encode(), C.c_float(9e-7), C.byref(h), C.byref(d)) == 0
        assert ora.tntor_heterodimer(C.byref(opt), o.encode(), p.encode(), C.c_float(9e-7), C.c_float(4e-7), C.byref(x)) == 0
        w = (h.value, d.value, x.value)
        nz += sum(1 for v in w if v > 0)
        if [tc.f32_bits(v) for v in g] != [tc.f32_bits(v) for v in w]:
            bad += 1
            if bad < 8: print("DIFF", dangle, o, p, g, w)
    eng.close()
print("oligos", len(oligos), "nonzero Tm values", nz, "bad", bad)
sys.exit(1 if bad else 0)
