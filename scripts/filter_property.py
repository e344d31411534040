#!/usr/bin/env python3
"""The soundness property of the bind pre-filter (tests/test_filter_bound.py) on N windows per option set
(default 1 000 000): prints one JSON record per option set (second argument: only the sets whose name holds it).
TIER=3 in the environment: both tiers (the smaller of the first tier's bound and the second tier's credit of align_dimer's
maximal-score paths, tnt_sbound.hip.h).  CPU only (oracle)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_filter_bound as T  # noqa: E402
import tntcheck as tc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
chunk = 100000
tier = int(os.environ.get("TIER", "1"))
ora = tc.OracleImpl()
SETS = [(name, kw, {}) for name, kw in T.OPTION_SETS]
# inosine-bearing windows (every target gets one to three letters, 70 % of them inosines), against a Tm, a dG and a dangle set
SETS += [(T.OPTION_SETS[k][0] + ", inosine in the targets", T.OPTION_SETS[k][1], dict(degen=1.0, inosine=0.7)) for k in (0, 5, 9)]
only = sys.argv[2] if len(sys.argv) > 2 else ""  # substring of the names of the sets to run
SETS += [(name, kw, dict(lo=65, hi=128)) for name, kw in T.LONG_OLIGO_SETS]
for k, (name, kw, wkw) in enumerate(SETS):
    if only not in name:
        continue
    tot = None
    for part in range((n + chunk - 1) // chunk):
        r = T.check_option_set(ora, name, kw, min(chunk, n - part * chunk), 1000 * (k + 1) + part, tier=tier, **wkw)
        if tot is None:
            tot = r
        else:
            for key in ("windows", "with_alignment", "accepted", "rejected_by_filter"):
                tot[key] += r[key]
            tot["min_slack_kcal"] = min(tot["min_slack_kcal"], r["min_slack_kcal"])
    tot.pop("reject_rate_of_non_accepted", None)
    tot["violations"] = 0
    tot["tier"] = tier
    print(json.dumps(tot), flush=True)
