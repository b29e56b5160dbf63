#!/bin/bash
# timing experiments with builds whose results are WRONG on purpose (parts of a kernel compiled out):
# no parity gate.  tools/ab_noparity.sh <workload> <lib.so>...
W=$1; shift
for lib in "$@"; do URF_LIB_PATH=$PWD/$lib timeout 300 python bench.py --workload $W --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-outputs --no-other-configs --parity-scans 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$W', d['ms_per_step'], ' '.join('%s=%.3f'%(k[2:],v) for k,v in d['kernel_ms'].items()))"; done
