#!/bin/bash
# Round-end measurement on the GPU box (through gpurun): bench line, rocprofv3 kernel stats of the same command,
# PMC passes.  Outputs land under gpurun_out/<tag>/; scripts/pmc_summarize.py folds them into profiles/.
#   bash scripts/profile_round.sh r01d
set -u
TAG=${1:-round}
ROOT=$(pwd)
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python3 bench.py --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT" -o bench -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$ROOT/$OUT/bench_prof.json" 2> "$ROOT/$OUT/bench_prof.err"
cd "$ROOT"
head -12 "$OUT/bench_kernel_stats.csv"
bash scripts/pmc_collect.sh "$OUT/pmc" 2 1000000 | tail -6
