#!/bin/bash
# rocprofv3 evidence for the A/B of the two bind-kernel forms (GPU box, through gpurun): kernel-trace statistics and PMC
# passes (counters in runs of their own) of the same workload under the lane-per-window kernel (default) and the
# wavefront-per-window kernel (TNT_MI355X_WAVE_BIND).  Launches serialised on one stream so that durations add up.
#   bash scripts/ab_wave_profile.sh <tag>     -> gpurun_out/<tag>/{lane,wave}_<workload>/...
set -u
TAG=${1:-abprof}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export TNT_MI355X_SERIAL_BIND=1
passes() { # dir, program
  d=$1; prog=$2; mkdir -p "$d"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o trace -- python3 "$ROOT/scripts/$prog" > "$d/trace.log" 2>&1; tail -1 "$d/trace.log"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$d" -o sq_insts -- python3 "$ROOT/scripts/$prog" > "$d/sq_insts.log" 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d "$d" -o sq_cycles -- python3 "$ROOT/scripts/$prog" > "$d/sq_cycles.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$d" -o tcc_fetch -- python3 "$ROOT/scripts/$prog" > "$d/tcc_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$d" -o tcc_write -- python3 "$ROOT/scripts/$prog" > "$d/tcc_write.log" 2>&1
}
# (a) config 2, every window bound (no pre-filter): the kernels alone on 18-25-mers
export CFG=2 NT=50000 TNT_MI355X_NO_FILTER=1
passes "$OUT/lane_config2_nofilter" time_variants.py
TNT_MI355X_WAVE_BIND=1 passes "$OUT/wave_config2_nofilter" time_variants.py
# (b) config 5 (probes of 25-40 bases, IUPAC targets) behind the pre-filter: the buckets of 33 rows and more through the wave form
unset TNT_MI355X_NO_FILTER; export CFG=5 NT=100000
passes "$OUT/lane_config5" time_variants.py
TNT_MI355X_WAVE_BIND=33 passes "$OUT/wave33_config5" time_variants.py
# (c) 96 hybridisation probes of 100 bases, every window bound
export LEN=100 NT=20000 TNT_MI355X_NO_FILTER=1
passes "$OUT/lane_probes100" time_long_probes.py
TNT_MI355X_WAVE_BIND=1 passes "$OUT/wave_probes100" time_long_probes.py
cd "$ROOT"
for d in lane_config2_nofilter wave_config2_nofilter lane_config5 wave33_config5 lane_probes100 wave_probes100; do
  python3 scripts/pmc_summarize.py "gpurun_out/$TAG/$d" "gpurun_out/$TAG/$d.json" $([ "${d#*probes}" != "$d" ] && echo 4 || echo 3) > /dev/null 2>&1
  echo "$d: $(ls gpurun_out/$TAG/$d.json 2>/dev/null)"
done
