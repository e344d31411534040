#!/bin/bash
# PMC passes of rocprofv3 over one search workload (run on the GPU box through gpurun).
#   scripts/pmc_collect.sh <out-dir under gpurun_out> [CFG] [NT]
# Counters are collected in their own runs (no trace domains besides the kernel dispatch records), one
# pass per hardware-slot budget: SQ 8, TCC 4 (FETCH_SIZE costs 3, WRITE_SIZE 2), GRBM 2.
set -u
OUT=${1:-gpurun_out/pmc}; export CFG=${2:-2}; export NT=${3:-200000}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/$OUT" -o "$name" -- python3 "$ROOT/scripts/time_variants.py" > "$ROOT/$OUT/$name.log" 2>&1
  tail -1 "$ROOT/$OUT/$name.log"
}
pass sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass sq_cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
pass tcc_fetch FETCH_SIZE
pass tcc_write WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT" -o trace -- python3 "$ROOT/scripts/time_variants.py" > "$ROOT/$OUT/trace.log" 2>&1
tail -1 "$ROOT/$OUT/trace.log"
find "$ROOT/$OUT" -name "*.csv" | head -20
