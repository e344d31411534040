"""First GPU bring-up: HIP path vs oracle vs reference harness on small workloads."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tntcheck as tc
from thermonucleotideblast_amd import abi, synth
from thermonucleotideblast_amd.engine import Engine

ora = tc.OracleImpl()
ref = tc.RefImpl() if tc.have_ref() else None
print("have ref:", ref is not None)
ok = True
for cfg, nt, na in ((2, 300, 24), (3, 300, 24), (5, 300, 30), (4, 200, 40), (1, 1000, None)):
    wl = synth.make_workload(cfg, n_targets=nt, n_assays=na)
    eng = Engine(wl.options, wl.expanded)
    db = eng.pack(wl.codes, wl.offsets)
    # tables
    dg = eng.delta_g()
    assert np.array_equal(dg, ora.tables(wl.options)["delta_g"]), "delta_g"
    # seeds
    t = time.time(); sg = eng.seeds(db); tg = time.time() - t
    so = ora.seeds(wl.options, eng.oligos, wl.codes, wl.offsets)
    so = sorted((s[0], eng.oligos[s[1]], s[2], s[3], s[4]) for s in so)
    sg = sorted(sg)
    print("cfg", cfg, "seeds gpu", len(sg), "oracle", len(so), "equal", sg == so, "%.3fs" % tg)
    ok &= (sg == so)
    # binds on all seeds
    om = {o: i for i, o in enumerate(eng.oligos)}
    sub = so[:20000]
    reqs = tc.make_reqs((s[0], om[s[1]], s[2], s[3], s[4], 0, 9e-7) for s in sub)
    t = time.time(); bg = eng.bind_batch(db, reqs); tg = time.time() - t
    bo = ora.bind_batch(wl.options, eng.oligos, wl.codes, wl.offsets, reqs)
    bad = 0
    for i in range(len(reqs)):
        d = tc.bind_compare_fields(bo[i], bg[i])
        if d:
            bad += 1
            if bad <= 3:
                print("  BIND DIFF", sub[i], d); print(bo[i].align.decode()); print(bg[i].align.decode())
    print("cfg", cfg, "binds", len(reqs), "bad", bad, "%.3fs" % tg)
    ok &= (bad == 0)
    # search
    t = time.time(); hg = eng.search(db); tg = time.time() - t
    ho = ora.search(wl.options, wl.expanded, wl.codes, wl.offsets)
    kg, ko = tc.sorted_hit_keys(hg), tc.sorted_hit_keys(ho)
    print("cfg", cfg, "hits gpu", len(kg), "oracle", len(ko), "equal", kg == ko, "%.3fs" % tg)
    if kg != ko:
        sa, sb = set(kg), set(ko)
        print("  only gpu", len(sa - sb), "only oracle", len(sb - sa))
        for x in sorted(sa - sb)[:3]: print("   G", x[:30])
        for x in sorted(sb - sa)[:3]: print("   O", x[:30])
    ok &= (kg == ko)
    if ref is not None:
        hr = ref.search(wl.options, wl.expanded, wl.codes, wl.offsets)
        kr = tc.sorted_hit_keys(hr)
        print("cfg", cfg, "hits ref", len(kr), "gpu==ref", kg == kr)
        ok &= (kg == kr)
    tm = eng.timing()
    print("   timing seed %.2f bind %.2f pair %.2f total %.2f ms; binds %d bound %d" % (tm.seed_ms, tm.bind_ms, tm.pair_ms, tm.total_ms, tm.n_binds, tm.n_bound))
    db.close(); eng.close()
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
