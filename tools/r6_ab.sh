#!/bin/bash
# A/B of the fused front end inside one gpurun call: same box, same batch
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-e2e --no-outputs --no-other-configs --steps 20 --warmup 5"
for wl in cfg3 sensor; do
  for f in 0 1; do
    python bench.py $Q --workload $wl --front $f > gpurun_out/r6_ab_${wl}_front$f.json 2> gpurun_out/r6_ab_${wl}_front$f.err
    python - <<P
import json
try:
    d=json.load(open("gpurun_out/r6_ab_${wl}_front$f.json"))
    print("$wl front $f: %.1f scans/s, %.4f ms/step, fused %s" % (d["value"], d["ms_per_step"], d.get("front_scans_per_gpu")), d["kernel_ms"])
except Exception as e:
    print("$wl front $f failed", e); print(open("gpurun_out/r6_ab_${wl}_front$f.err").read()[-2000:])
P
  done
done
