#!/usr/bin/env python3
"""Cycle split of the seed kernel's phases (GPU box): runs scripts/time_variants.py with the TNT_EXP_STATS build
(exp/libstats.so: `python -c "import __graft_entry__ as g; g.build_hip(extra_flags=['-DTNT_EXP_STATS'], out='exp/libstats.so', host_only=True)"`)
and writes the shares as JSON.  bench.py reads profiles/seed_phases.json for roofline_seed.scan_only.
   python scripts/seed_phases.py out.json [CFG] [NT]"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "2"
nt = sys.argv[3] if len(sys.argv) > 3 else "1000000"
env = dict(os.environ, CFG=cfg, NT=nt, TNT_MI355X_LIB=os.path.join(ROOT, "exp", "libstats.so"))
r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "time_variants.py")], env=env, capture_output=True, text=True)
lines = [l for l in r.stderr.splitlines() if l.startswith("seed2 waves")]
m = re.search(r"seed2 waves (\d+); kilo-cycles per wave: setup (\d+) scan (\d+) dense (\d+) sort (\d+) cull (\d+) emit (\d+)", lines[-1])
waves = int(m.group(1))
names = ["setup", "scan", "dense", "sort", "cull", "emit"]
kc = dict(zip(names, (int(x) for x in m.groups()[1:])))
total = sum(kc.values())
rec = {"source": "scripts/seed_phases.py: s_memtime stamps of the TNT_EXP_STATS build (the stamps themselves cost about a tenth of the wave cycles)",
       "config": int(cfg), "targets": int(nt), "waves": waves, "kilo_cycles_per_wave": kc, "share": {k: v / total for k, v in kc.items()},
       "stats_line": lines[-1], "timing_line": r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec["share"]))
