"""The scan of the seed stage on its own (tnt_seed_scan_only) over config 2 at full size: run under rocprofv3 for the kernel
statistics / FETCH_SIZE of scan_only_kernel (scripts/final_profiles.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thermonucleotideblast_amd import synth
from thermonucleotideblast_amd.engine import Engine
nt = int(os.environ.get("NT", "1000000"))
wl = synth.make_workload(int(os.environ.get("CFG", "2")), n_targets=nt)
eng = Engine(wl.options, wl.expanded)
db = eng.pack(wl.codes, wl.offsets)
ms, n = eng.scan_only(db, repeats=10)
print("scan only: %.3f ms per pass over %d targets (%.0f GB/s of packed targets), %d candidates per pass" % (ms, nt, 0.25 * wl.n_bases / ms / 1e6, n))
