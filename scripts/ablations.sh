#!/bin/bash
# One bench line + one kernel-stats CSV per "tried / not kept" variant of the search (switches of DESIGN.md 8), so that
# each can be re-examined from counters rather than prose.  Run on the GPU box: bash scripts/ablations.sh <tag>
#   outputs: gpurun_out/<tag>/ablation_<name>.json, ablation_<name>_kernel_stats.csv
set -u
TAG=${1:-ablations}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, then VAR=value ...
  local name=$1; shift
  for kv in "$@"; do export "$kv"; done
  timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-targets 16384 > $OUT/ablation_$name.json 2> $OUT/ablation_$name.err
  TNT_MI355X_SERIAL_BIND=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-targets 0 > /dev/null 2>> $OUT/ablation_$name.err
  local f; f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/ablation_${name}_kernel_stats.csv
  rm -rf $OUT/prof_$name
  for kv in "$@"; do unset "${kv%%=*}"; done
  python3 - "$OUT/ablation_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s step %.2f ms  seed %.2f  bind %.2f  pair %.2f  hits %d  equal %s" % (sys.argv[2], d["ms_per_step"], d["stage_ms_per_step"]["seed"],
          d["stage_ms_per_step"]["bind"], d["stage_ms_per_step"]["pair"], d["hits_per_step"], d.get("hit_multiset_equal")))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
}
run default
run seed_kernel_round2 TNT_MI355X_SEED_V2=1
run seed_kernel_round2_lds_tables TNT_MI355X_SEED_V2=1 TNT_MI355X_SEED_LDS_TABLES=1
run phase_b_bucket_launches TNT_MI355X_PHASE_B_BUCKETS=1
run no_prefilter TNT_MI355X_NO_FILTER=1
run prefilter_plain_only TNT_MI355X_NO_FILTER_GENERAL=1
run pipeline_2_batches TNT_MI355X_PIPELINE=2
run bind_lanes_8 TNT_MI355X_BIND_LANES=8
