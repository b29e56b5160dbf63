#!/bin/bash
# A/B inside one gpurun call: libraries x workloads x front mode.  tools/r6_ab2.sh "<lib or - ...>" "<workload ...>" "<front ...>"
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-e2e --no-outputs --no-other-configs --steps 20 --warmup 5"
for lib in $1; do for wl in $2; do for f in $3; do
  tag=$(basename $lib .so)_${wl}_f$f
  if [ "$lib" = "-" ]; then unset URF_LIB_PATH; else export URF_LIB_PATH=$PWD/$lib; fi
  timeout 300 python bench.py $Q --workload $wl --front $f > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/ab_$tag.json"))
    print("%-34s %9.1f scans/s %.4f ms fused %s  " % ("$tag", d["value"], d["ms_per_step"], d.get("front_scans_per_gpu")) + " ".join("%s=%.3f" % (k[2:], v) for k, v in d["kernel_ms"].items()))
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/ab_$tag.err").read()[-1500:])
P
done; done; done
