#!/bin/bash
# One GPU-box session of the development loop (through gpurun): targeted parity tests first, then the whole GPU suite,
# then a short bench.  Logs land under gpurun_out/<tag>/.
#   bash scripts/gpu_step.sh <tag> ["pytest -k expression" | -] [bench args]
set -u
TAG=${1:-step}; KEXPR=${2:--}; shift; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
if [ "$KEXPR" != "-" ]; then
  timeout 1200 python3 -m pytest tests -m gpu -x -q -k "$KEXPR" > "$OUT/pytest_k.log" 2>&1
  echo "targeted tests: exit $?"; tail -15 "$OUT/pytest_k.log"
fi
timeout 2400 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest_all.log" 2>&1
echo "GPU suite: exit $?"; tail -8 "$OUT/pytest_all.log"
timeout 900 python3 bench.py --steps 5 --warmup 1 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench: exit $?"; tail -c 1500 "$OUT/bench.json"; echo; tail -5 "$OUT/bench.err"
