#!/bin/bash
# Collects rocprofv3 PMC counters for the bench workload, one counter group per pass
# (never combined with tracing domains -- only --kernel-trace is allowed next to --pmc on this pool).
#   tools/pmc_profile.sh <out_dir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUT" in /*) ;; *) OUT="$REPO/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-outputs --no-other-configs --parity-scans 0 $*"
pass() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
pass insts   SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass stalls  SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
pass fetch   FETCH_SIZE
pass write   WRITE_SIZE
pass l2      TCC_HIT_sum TCC_MISS_sum
find "$OUT" -name "*.csv" | head -20
