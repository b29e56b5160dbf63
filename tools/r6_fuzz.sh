#!/bin/bash
# round 6 fuzz on the GPU box, P processes side by side (the CPU oracle is what takes the time):
#   the organised sweeps with holes through the FUSED front end (tools/front_check.py), the random unorganised clouds through the
#   general kernels (tools/fuzz_more.py: k_index's run flags, k_star_sort_runs), the organised sweeps through the general kernels
P=${1:-16}; NF=${2:-1500}; NU=${3:-800}; NO=${4:-500}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6_fuzz
for i in $(seq 0 $((P-1))); do
  ( timeout 1500 python tools/front_check.py --basic 0 --fuzz $NF --seed0 $((9100000 + i*NF)) > gpurun_out/r6_fuzz/front_$i.log 2>&1
    timeout 1500 python tools/fuzz_more.py $((6000000 + i*NU)) $((6000000 + (i+1)*NU)) > gpurun_out/r6_fuzz/unorg_$i.log 2>&1
    timeout 1500 python tools/fuzz_organised_more.py $((7600000 + i*NO)) $((7600000 + (i+1)*NO)) > gpurun_out/r6_fuzz/org_$i.log 2>&1 ) &
done
wait
grep -h "fuzz:" gpurun_out/r6_fuzz/front_*.log | awk '{c+=$2; f+=$4; r+=$10; m+=$15} END {print "fused front end, organised sweeps with holes:", c, "cases,", f, "took it in firing order,", r, "of their row-major twins,", m, "mismatches"}'
grep -h "mismatches\|MISMATCH" gpurun_out/r6_fuzz/unorg_*.log | tail -$((P+3))
grep -h "organised fuzz\|MISMATCH" gpurun_out/r6_fuzz/org_*.log | tail -$((P+3))
