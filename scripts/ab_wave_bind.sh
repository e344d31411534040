#!/bin/bash
# A/B of the two forms of the bind kernel on the GPU box: lane-per-window (bind_kernel, default) against
# wavefront-per-window (wbind_kernel, TNT_MI355X_WAVE_BIND=<rows from which buckets use it>).
#   bash scripts/ab_wave_bind.sh <tag>      -> gpurun_out/<tag>/ab_*.txt
set -u
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
run() { # label, env..., -- command
  label=$1; shift
  echo "== $label" >> "$OUT/ab_wave_bind.txt"
  env "$@" 2>&1 | tail -1 >> "$OUT/ab_wave_bind.txt"
}
: > "$OUT/ab_wave_bind.txt"
# config 2 (primers of 18-25 bases) behind the pre-filter, and with every window bound (no pre-filter: the kernels alone)
run "config 2, 200k targets, lane form"            CFG=2 NT=200000 python3 scripts/time_variants.py
run "config 2, 200k targets, wave form"            CFG=2 NT=200000 TNT_MI355X_WAVE_BIND=1 python3 scripts/time_variants.py
run "config 2, 50k targets, no filter, lane form"  CFG=2 NT=50000 TNT_MI355X_NO_FILTER=1 python3 scripts/time_variants.py
run "config 2, 50k targets, no filter, wave form"  CFG=2 NT=50000 TNT_MI355X_NO_FILTER=1 TNT_MI355X_WAVE_BIND=1 python3 scripts/time_variants.py
# config 5 (padlock arms 20-25 + probes 25-40, IUPAC targets)
run "config 5, 100k targets, lane form"            CFG=5 NT=100000 python3 scripts/time_variants.py
run "config 5, 100k targets, wave form from 33"    CFG=5 NT=100000 TNT_MI355X_WAVE_BIND=33 python3 scripts/time_variants.py
run "config 5, 100k targets, wave form"            CFG=5 NT=100000 TNT_MI355X_WAVE_BIND=1 python3 scripts/time_variants.py
# long hybridisation probes, every window bound
for len in 40 64 100 128; do
  run "96 probes of $len bases x 20k targets, no filter, lane form" LEN=$len NT=20000 TNT_MI355X_NO_FILTER=1 python3 scripts/time_long_probes.py
  run "96 probes of $len bases x 20k targets, no filter, wave form" LEN=$len NT=20000 TNT_MI355X_NO_FILTER=1 TNT_MI355X_WAVE_BIND=1 python3 scripts/time_long_probes.py
done
cat "$OUT/ab_wave_bind.txt"
