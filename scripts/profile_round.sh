#!/bin/bash
# Round-end measurement on the GPU box (through gpurun): bench line, rocprofv3 kernel stats of the same command (as it
# runs, and with the bind launches serialised on one stream so that per-kernel durations add up), the other configs with a
# bounded parity sample, the pre-filter's audit mode on full-size configs, PMC passes.
# Outputs land under gpurun_out/<tag>/; copy what is to be judged into profiles/.
#   bash scripts/profile_round.sh r03
set -u
TAG=${1:-round}
ROOT=$(pwd)
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python3 bench.py --steps 5 --warmup 1 > "$OUT/bench_config2.json" 2> "$OUT/bench_config2.err"
tail -c 600 "$OUT/bench_config2.json"; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT" -o bench -- python3 "$ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline > "$ROOT/$OUT/bench_prof.json" 2> "$ROOT/$OUT/bench_prof.err"
TNT_MI355X_SERIAL_BIND=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT" -o bench_serial -- python3 "$ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline > "$ROOT/$OUT/bench_serial_prof.json" 2> "$ROOT/$OUT/bench_serial_prof.err"
cd "$ROOT"
head -14 "$OUT/bench_kernel_stats.csv" | cut -c1-160
for c in 3 4 5; do
  python3 bench.py --config $c --steps 2 --warmup 1 --cpu-seconds 6 --parity-targets 16384 > "$OUT/bench_config$c.json" 2> "$OUT/bench_config$c.err"
  tail -c 300 "$OUT/bench_config$c.json"; echo
done
# pre-filter audit (everything is bound; the bind kernel raises ERR_FILTER_UNSOUND if it accepts a window the filter marked)
for c in 2 5; do
  TNT_MI355X_FILTER_AUDIT=1 python3 bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/audit_config$c.json" 2> "$OUT/audit_config$c.err"
  echo "audit config $c: exit $?"; tail -c 200 "$OUT/audit_config$c.json"; echo
done
