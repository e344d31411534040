#!/bin/bash
# Randomised differential runs of a round on the GPU box (through gpurun): HIP path == oracle over random option sets and
# target shapes, the same with the pre-filter's audit mode, the front end against the reference binary.
#   bash scripts/gpu_fuzz_round.sh <tag> <cases> <seed> [audit cases]      (T1 / T2 / T3: time limits of the three steps in seconds;
#   progress lines are flushed every 50 cases, so a step cut short still tells how far it got)
set -u
TAG=${1:-fuzz}; N=${2:-400}; SEED=${3:-301}; NA=${4:-0}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
( time FUZZ_SLOW=8 FUZZ_LONG=0.05 timeout ${T1:-1500} python3 scripts/fuzz_parity.py $N $SEED ) > "$OUT/fuzz_seed${SEED}_${N}cases.log" 2>&1
tail -4 "$OUT/fuzz_seed${SEED}_${N}cases.log"
if [ "$NA" != "0" ]; then
  ( time TNT_MI355X_FILTER_AUDIT=1 FUZZ_SLOW=8 FUZZ_LONG=0.05 timeout ${T2:-700} python3 scripts/fuzz_parity.py $NA $((SEED + 1)) ) > "$OUT/fuzz_seed$((SEED + 1))_${NA}cases_filter_audit.log" 2>&1
  tail -4 "$OUT/fuzz_seed$((SEED + 1))_${NA}cases_filter_audit.log"
  timeout ${T3:-400} python3 scripts/cli_compare.py > "$OUT/cli_compare.txt" 2>&1; echo "cli_compare: exit $?"; tail -3 "$OUT/cli_compare.txt"
fi
