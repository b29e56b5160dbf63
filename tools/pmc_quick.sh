#!/bin/bash
# two PMC passes (instruction mix, stalls / LDS) for one workload: tools/pmc_quick.sh <out_dir> <workload>
set -u
OUT=$1; W=${2:-cfg3}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUT" in /*) ;; *) OUT="$REPO/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs --parity-scans 0"
pass() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
pass insts   SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass stalls  SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
