#!/bin/bash
# Round-end measurement on the GPU box (through gpurun): bench line, rocprofv3 kernel stats of the same command (as it
# runs, and with the bind launches serialised on one stream so that per-kernel durations add up), the other configs.
# Outputs land under gpurun_out/<tag>/; copy what is to be judged into profiles/.
#   bash scripts/profile_round.sh r02
set -u
TAG=${1:-round}
ROOT=$(pwd)
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python3 bench.py --steps 5 --warmup 1 > "$OUT/bench_config2.json" 2> "$OUT/bench_config2.err"
tail -c 400 "$OUT/bench_config2.json"; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT" -o bench -- python3 "$ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline > "$ROOT/$OUT/bench_prof.json" 2> "$ROOT/$OUT/bench_prof.err"
TNT_MI355X_SERIAL_BIND=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT" -o bench_serial -- python3 "$ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline > "$ROOT/$OUT/bench_serial_prof.json" 2> "$ROOT/$OUT/bench_serial_prof.err"
cd "$ROOT"
head -14 "$OUT/bench_kernel_stats.csv" | cut -c1-160
for c in 3 4 5; do
  python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_config$c.json" 2> "$OUT/bench_config$c.err"
  tail -c 300 "$OUT/bench_config$c.json"; echo
done
