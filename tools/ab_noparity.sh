#!/bin/bash
# timing experiments with builds whose results are WRONG on purpose (parts of a kernel compiled out): no parity gate.
# (The probes of rounds 2-5 -- URF_EXP_SKIP_EVAL / _CAND / _EPILOGUE in k_ring, the "off" switches of k_split's organised modes, the
# arc-cosine and staging variants -- left the source in round 6: their A/B records are profiles/r4_*_ab.txt, r5_*_ab.txt.)
# tools/ab_noparity.sh <workload> <lib.so>...
W=$1; shift
for lib in "$@"; do URF_LIB_PATH=$PWD/$lib timeout 300 python bench.py --workload $W --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-outputs --no-other-configs --parity-scans 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$W', d['ms_per_step'], ' '.join('%s=%.3f'%(k[2:],v) for k,v in d['kernel_ms'].items()))"; done
