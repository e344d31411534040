#!/bin/bash
# End-of-round measurement on the GPU box (through gpurun), on the final build: the bench line, rocprofv3 kernel statistics of
# the same command (as it runs, and with the bind launches serialised), the PMC passes of config 2 at full size (FETCH / WRITE
# sizes, instruction counts: profiles/bind_pmc.json is made from them), the standalone scan launch under rocprofv3, the other
# configurations.  Outputs land under gpurun_out/<tag>/; copy what is to be judged into profiles/.
#   bash scripts/final_profiles.sh r04z
set -u
TAG=${1:-final}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
python3 bench.py --steps 10 --warmup 2 > "$OUT/bench_config2.json" 2> "$OUT/bench_config2.err"; tail -c 400 "$OUT/bench_config2.json"; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- python3 "$ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/bench_prof.json" 2> "$OUT/bench_prof.err"
TNT_MI355X_SERIAL_BIND=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench_serial -- python3 "$ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/bench_serial_prof.json" 2> "$OUT/bench_serial_prof.err"
cd "$ROOT"
TNT_MI355X_SERIAL_BIND=1 bash scripts/pmc_collect.sh "gpurun_out/$TAG/pmc" 2 1000000 > "$OUT/pmc_collect.log" 2>&1
python3 scripts/pmc_summarize.py "gpurun_out/$TAG/pmc" "gpurun_out/$TAG/pmc_config2_1M_serial.json" > "$OUT/pmc_summary.txt" 2>&1; tail -3 "$OUT/pmc_summary.txt" | cut -c1-300
# the standalone scan launch: kernel statistics + FETCH_SIZE
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/scan" -o scan -- python3 "$ROOT/scripts/scan_only.py" > "$OUT/scan_only.log" 2>&1; tail -1 "$OUT/scan_only.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/scan" -o scan_fetch -- python3 "$ROOT/scripts/scan_only.py" > "$OUT/scan_fetch.log" 2>&1
cd "$ROOT"
for c in 3 4 5; do
  python3 bench.py --config $c --steps 2 --warmup 1 --cpu-seconds 6 --parity-targets 16384 > "$OUT/bench_config$c.json" 2> "$OUT/bench_config$c.err"
  tail -c 300 "$OUT/bench_config$c.json"; echo
done
for c in 2 5; do
  TNT_MI355X_FILTER_AUDIT=1 python3 bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/audit_config$c.json" 2> "$OUT/audit_config$c.err"
  echo "audit config $c: exit $?"; tail -c 200 "$OUT/audit_config$c.json"; echo
done
