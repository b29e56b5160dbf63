#!/bin/bash
P=${1:-16}; NF=${2:-1500}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6_fuzz
for i in $(seq 0 $((P-1))); do
  ( timeout 3000 python tools/front_check.py --basic 0 --fuzz $NF --callback ${CALLBACK:-0} --seed0 $((${SEED0:-9100000} + i*NF)) > gpurun_out/r6_fuzz/front_$i.log 2>&1 ) &
done
wait
grep -h "fuzz:" gpurun_out/r6_fuzz/front_*.log | awk '{c+=$2; f+=$4; r+=$10; m+=$15} END {print "fused front end, organised sweeps with holes:", c, "cases,", f, "took it in firing order,", r, "of their row-major twins,", m, "mismatches"}'
grep -h "MISMATCH" gpurun_out/r6_fuzz/front_*.log | head
