#!/bin/bash
# A/B of a variant build of the wave kernels against the product on the workloads they carry: tools/ab_wave_variants.sh TAG VARIANT
# (build_exp/VARIANT/libstringsext_amd.so, tools/build_variant.sh) -> gpurun_out/TAG_{base,VARIANT}_*.{json,txt}
tag=$1; var=$2
for lib in base $var; do
  if [ $lib = base ]; then unset SX_LIB; else export SX_LIB=$(pwd)/build_exp/$lib/libstringsext_amd.so; fi
  python bench.py --workload c1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_${lib}_c1.json
  python bench.py --workload c1 --steps 20 --warmup 5 --no-cpu-baseline --result-on-device 2>/dev/null | tail -1 > gpurun_out/${tag}_${lib}_c1_dev.json
  python tools/gpu_text.py > gpurun_out/${tag}_${lib}_text.txt 2>&1
  sleep 10
  python bench.py --workload c5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_${lib}_c5.json
  python - <<PY
import json
for w in ("c1", "c1_dev", "c5"):
    d = json.load(open("gpurun_out/${tag}_${lib}_%s.json" % w))
    print("$lib", w, d["value"], "GiB/s", d["ms_per_step"], "ms", d["roofline"]["wave_passes_ms"])
PY
  cat gpurun_out/${tag}_${lib}_text.txt | cut -c1-110
done
