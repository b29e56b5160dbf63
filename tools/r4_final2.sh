#!/bin/bash
# round 4, evidence at the round's last code state: profile round (cfg3 bench + kernel trace + PMC), kernel trace of the callback
# path, the GPU suite, smoke(), 200 000 fuzz clouds
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4_final2"
mkdir -p "$OUT"
bash tools/profile_round.sh gpurun_out/r4_final2 2>&1 | grep -v amdgpu.ids | cut -c1-300
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d "$OUT/cbtrace" -o t -- python "$GRAFT_REPO_ROOT/tools/host_times.py" 200 > "$OUT/cbtrace.log" 2>&1; echo "callback trace rc=$?" )
python tools/trace_overlap.py $(ls $OUT/cbtrace/*results.db $OUT/cbtrace/*/*results.db 2>/dev/null | head -1) > "$OUT/callback_trace.txt" 2>&1; tail -14 "$OUT/callback_trace.txt"
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python tools/fuzz_more.py 1000000 1200000 > "$OUT/fuzz.log" 2>&1; echo "fuzz rc=$?"; tail -2 "$OUT/fuzz.log"
